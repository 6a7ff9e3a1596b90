#!/bin/bash
# End-of-round evidence on ONE box: rocprofv3 trace + PMC passes of bench.py (tools/gpu_profile.sh), the full default bench
# line, every single-GPU configuration against the oracle, reference mode (latency, workloads, per-pass kernel trace), the
# ROS-free request harness at three frame sizes, scene normals and scene filters.  Everything lands in gpurun_out/<TAG>/;
# python tools/summarize_prof.py <TAG> + a copy of the .json/.jsonl/.txt files into profiles/ is the commit.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r03k}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
step() { local secs=$1 log=$2; shift 2; echo "== $* (limit ${secs}s)" | tee -a "$O/steps.log"; timeout -k 10 "$secs" "$@" > "$O/$log" 2> "$O/$log.err"; local rc=$?; echo "rc=$rc $log" | tee -a "$O/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $log: stopping" | tee -a "$O/steps.log"; exit 99; fi; return $rc; }
: > "$O/steps.log"
step 400 bench.json python bench.py --steps 20 --warmup 5
TAG=$TAG STEPS=3 BENCH_ARGS="--no-exact-leg --no-config-legs" bash tools/gpu_profile.sh > "$O/profile.log" 2>&1 || { echo "profile failed" | tee -a "$O/steps.log"; tail -5 "$O/profile.log"; }
cd "$R"
step 400 baseline_configs.jsonl python tools/bench_configs.py
step 300 reference_request_latency.json python tools/refmode_latency.py --reps 60
step 400 reference_mode.jsonl python tools/bench_reference_mode.py
{ ./build/estimate_pose_harness 20000 100000 6 pipeline; ./build/estimate_pose_harness 20000 300000 6 pipeline; ./build/estimate_pose_harness 100000 1000000 6 pipeline; ./build/estimate_pose_harness 100000 1000000 6 pipeline pruned; } > "$O/harness.txt" 2>&1; echo "rc=$? harness" | tee -a "$O/steps.log"
step 300 scene_normals.jsonl python tools/bench_normals.py
step 300 scene_filter_bandwidth.jsonl python tools/bench_scene_filters.py
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/reftrace" -- python3 "$R/tools/refmode_latency.py" --reps 3 > "$O/reftrace.log" 2>&1; echo "rc=$? reftrace" | tee -a "$O/steps.log"
cd "$R"
f=$(find "$O/reftrace" -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_passes.py "$f" > "$O/reference_passes_trace.txt"
tail -3 "$O/steps.log"; tail -1 "$O/bench.json" | cut -c1-300; cat "$O/reference_request_latency.json"
