#!/bin/bash
# A/B of library builds on ONE box: bench.py's scan-launch time with each library, alternating, REPS rounds.
#   LIBS="base=pose_estimation_amd/lib/libmi355_icp.so varJ=build/varJ/libmi355_icp.so" TAG=abJ bash tools/gpu_ab_libs.sh
# Optional PARITY=<name>: runs the filter / parity tests with that library first.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-ab}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
: > "$O/summary.txt"
if [ -n "${PARITY:-}" ]; then
  for kv in $LIBS; do
    name=${kv%%=*}; lib=${kv#*=}
    if [ "$name" = "$PARITY" ]; then
      MI355_ICP_LIB=$R/$lib timeout -k 10 600 python -m pytest tests/test_gpu_fused_filter.py tests/test_gpu_filter_bounds.py tests/test_gpu_parity.py tests/test_gpu_filter_stress.py tests/test_gpu_reference_mode.py -m gpu -x -q > "$O/parity_$name.txt" 2>&1
      rc=$?; echo "parity $name rc=$rc: $(tail -1 "$O/parity_$name.txt")" | tee -a "$O/summary.txt"
      if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping" | tee -a "$O/summary.txt"; exit 99; fi
    fi
  done
fi
for rep in $(seq 1 ${REPS:-3}); do
  for kv in $LIBS; do
    name=${kv%%=*}; lib=${kv#*=}
    MI355_ICP_LIB=$R/$lib timeout -k 10 200 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > "$O/bench_${name}_$rep.json" 2> "$O/bench_${name}_$rep.err"
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a "$O/summary.txt"; exit 99; fi
    python3 - "$O/bench_${name}_$rep.json" "$name" "$rep" <<'PY' | tee -a "$O/summary.txt"
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:8s} rep {sys.argv[3]}: ms_per_step {d['ms_per_step']:.3f}  scan launch {d['roofline']['avg_launch_ms']:.3f} ms  frac {d['roofline']['frac']:.4f}  pose_vs_truth {d.get('final_pose_error_vs_truth')}")
except Exception as e:
    print(sys.argv[2], "rep", sys.argv[3], "FAILED", e)
PY
  done
done
