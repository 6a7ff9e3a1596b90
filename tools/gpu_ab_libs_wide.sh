#!/bin/bash
# A/B of library builds on ONE box over the other workloads: every BASELINE configuration (tools/bench_configs.py) and the
# reference-shaped request (tools/refmode_latency.py), alternating, REPS rounds.
#   LIBS="base=build/varBase/libmi355_icp.so new=pose_estimation_amd/lib/libmi355_icp.so" TAG=abw bash tools/gpu_ab_libs_wide.sh
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-abw}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
: > "$O/summary.txt"
for rep in $(seq 1 ${REPS:-2}); do
  for kv in $LIBS; do
    name=${kv%%=*}; lib=${kv#*=}
    MI355_ICP_LIB=$R/$lib timeout -k 10 300 python tools/bench_configs.py > "$O/cfg_${name}_$rep.jsonl" 2> "$O/cfg_${name}_$rep.err"
    rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a "$O/summary.txt"; exit 99; fi
    MI355_ICP_LIB=$R/$lib timeout -k 10 200 python tools/refmode_latency.py --reps 40 > "$O/lat_${name}_$rep.json" 2> "$O/lat_${name}_$rep.err"
    rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a "$O/summary.txt"; exit 99; fi
    python3 - "$O/cfg_${name}_$rep.jsonl" "$O/lat_${name}_$rep.json" "$name" "$rep" <<'PY' | tee -a "$O/summary.txt"
import json, sys
out = []
for ln in open(sys.argv[1]):
    try:
        d = json.loads(ln)
    except Exception:
        continue
    if "ms_per_iteration" in d:
        out.append(f"{str(d.get('config', '?'))[:12]} {d['ms_per_iteration']:.4f} (scan {d.get('scan_ms', 0):.4f}, resolve {d.get('resolve_ms', 0):.4f})")
try:
    lat = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    out.append(f"request {lat['wall_ms_median']:.4f} ms (scan {lat['profiled_scan_ms_total']:.3f}, other {lat['profiled_other_kernels_ms_total']:.3f})")
except Exception as e:
    out.append(f"request FAILED {e}")
print(f"{sys.argv[3]:6s} rep {sys.argv[4]}: " + " | ".join(out))
PY
  done
done
