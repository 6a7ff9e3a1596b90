#!/bin/bash
# A/B of environment settings on ONE box, alternating, REPS rounds: the reference request, the four single-GPU
# configurations and the request harness at the 450k-point frame under each setting.
#   VARIANTS="on= off=MI355_ICP_SPLIT_MODEL=0" TAG=absplit bash tools/gpu_ab_env.sh
# (name=ENVVAR=value; an empty setting is the default build; several variables in one setting are separated by COMMAS:
#  off=MI355_ICP_SPLIT_MODEL=0,MI355_ICP_MAX_SUB_SPLIT=1 - a setting may not contain spaces)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-abenv}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
: > "$O/summary.txt"
for rep in $(seq 1 ${REPS:-2}); do
  for kv in $VARIANTS; do
    name=${kv%%=*}; setting=${kv#*=}
    IFS=',' read -r -a assignments <<< "$setting"
    run() { local secs=$1 out=$2; shift 2; if [ -n "$setting" ]; then env "${assignments[@]}" timeout -k 10 "$secs" "$@" > "$out" 2> "$out.err"; else timeout -k 10 "$secs" "$@" > "$out" 2> "$out.err"; fi; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a "$O/summary.txt"; exit 99; fi; return $rc; }
    run 120 "$O/lat_${name}_$rep.json" python tools/refmode_latency.py --reps 40
    run 300 "$O/cfg_${name}_$rep.jsonl" python tools/bench_configs.py
    run 300 "$O/ref_${name}_$rep.jsonl" python tools/bench_reference_mode.py
    run 120 "$O/har_${name}_$rep.txt" ./build/estimate_pose_harness ${HARNESS:-20000 300000} 6 pipeline
    python3 - "$O" "$name" "$rep" <<'PY' | tee -a "$O/summary.txt"
import json, sys
O, name, rep = sys.argv[1:4]
def lines(p):
    try: return [json.loads(l) for l in open(p).read().strip().splitlines() if l.startswith("{")]
    except Exception: return []
lat = lines(f"{O}/lat_{name}_{rep}.json")
cfg = lines(f"{O}/cfg_{name}_{rep}.jsonl")
ref = lines(f"{O}/ref_{name}_{rep}.jsonl")
har = [l for l in open(f"{O}/har_{name}_{rep}.txt").read().splitlines() if l.startswith("pipeline pass 2")]
print(f"{name:8s} rep {rep}: request {lat[0]['wall_ms_median']:.4f} ms |", " ".join(f"{c['config'].split()[0]} {c['ms_per_iteration']:.4f}" for c in cfg), "|",
      " ".join(f"{1e3 * r['gpu_s']:.3f}" for r in ref), "|", har[0].split("ICP of 6 candidates ")[1] if har else "harness ?")
PY
  done
done
