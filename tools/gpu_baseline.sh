#!/bin/bash
# Round-start baseline on one GPU box: GPU suite, bench line, reference-request latency, kernel trace of one request.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-base}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
step() { local secs=$1 log=$2; shift 2; echo "== $* (limit ${secs}s)" | tee -a "$O/steps.log"; timeout -k 10 "$secs" "$@" > "$O/$log" 2>&1; local rc=$?; echo "rc=$rc $log" | tee -a "$O/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $log: stopping" | tee -a "$O/steps.log"; exit 99; fi; return $rc; }
: > "$O/steps.log"
cd "$R"
step 900 pytest_gpu.txt python -m pytest tests -m gpu -x -q || echo "PYTEST FAILED" | tee -a "$O/steps.log"
step 400 bench.txt python bench.py --steps 20 --warmup 5 || exit 1
step 200 refmode.txt python tools/refmode_latency.py --reps 60 || exit 1
step 200 harness.txt ./build/estimate_pose_harness 20000 100000 6 pipeline || true
cd /tmp && export TMPDIR=/tmp
step 300 trace.log rocprofv3 --kernel-trace --output-format csv -d "$O/trace" -- python3 "$R/tools/refmode_latency.py" --reps 3 || true
find "$O/trace" -name "*kernel_trace.csv" | head -3
tail -2 "$O/pytest_gpu.txt"; tail -1 "$O/bench.txt" | cut -c1-400; cat "$O/refmode.txt"; tail -5 "$O/harness.txt"
