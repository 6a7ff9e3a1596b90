#!/bin/bash
# Runs on the GPU box via gpurun: micro-benchmark (optional), GPU parity tests, smoke, short bench.
# A step that is killed by its timeout ends the script (no further GPU work after a hang).
set -u
mkdir -p gpurun_out
step() {  # step <seconds> <logfile> <cmd...>
  local secs=$1 log=$2; shift 2
  echo "== $* (limit ${secs}s)" | tee -a gpurun_out/steps.log
  timeout -k 10 "$secs" "$@" > "gpurun_out/$log" 2>&1
  local rc=$?
  echo "rc=$rc $log" | tee -a gpurun_out/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $log: stopping" | tee -a gpurun_out/steps.log; exit 99; fi
  return $rc
}
# PART=1 | 2 | all (default all): gpurun caps one call at 20 minutes; the legs are split so that either half fits
PART=${PART:-all}
: > gpurun_out/steps.log
if [ "$PART" != 2 ]; then
if [ "${UBENCH:-0}" = "1" ]; then step 120 ubench.txt ./build/ubench_valu; fi
step 900 pytest_gpu.txt python -m pytest tests -m gpu -x -q
# the same suite with each scan path forced for every launch size (the planner normally picks by size)
MI355_ICP_SCAN_MODE=fused step 900 pytest_gpu_fused.txt python -m pytest tests -m gpu -x -q
MI355_ICP_SCAN_MODE=xfilter step 900 pytest_gpu_xfilter.txt python -m pytest tests -m gpu -x -q
MI355_ICP_SCAN_MODE=xbf16 step 900 pytest_gpu_xbf16.txt python -m pytest tests -m gpu -x -q
MI355_ICP_SCAN_MODE=exact step 900 pytest_gpu_exact.txt python -m pytest tests -m gpu -x -q
MI355_ICP_SCAN_MODE=pruned step 900 pytest_gpu_pruned.txt python -m pytest tests -m gpu -x -q
fi
if [ "$PART" != 1 ]; then
# the matrix-core scan's other workgroup shape (8 waves x 2 source sets), which no planner rule picks
MI355_ICP_XBF16_SHAPE=1 MI355_ICP_SCAN_MODE=xbf16 step 600 pytest_gpu_xbf16_shape1.txt python -m pytest tests/test_gpu_fused_filter.py tests/test_gpu_parity.py tests/test_gpu_filter_bounds.py -m gpu -x -q
# the A/B switches of the request path: exhaustive k-NN (the committed scene then has no sorted copy to inherit), and the
# pre-round-2 structure of a reference pass (separate reduce + copy, a sort per pyramid level, scalar solves)
MI355_ICP_KNN_MODE=brute step 600 pytest_gpu_knn_brute.txt python -m pytest tests/test_gpu_normals.py tests/test_gpu_opencv_adapter.py tests/test_gpu_scene_filters.py tests/test_gpu_plane_fit.py -m gpu -x -q
MI355_ICP_LEVEL_SORT=1 MI355_ICP_PASS_TAIL=0 MI355_ICP_SOLVE_ISA=scalar step 600 pytest_gpu_plain_passes.txt python -m pytest tests/test_gpu_reference_mode.py tests/test_gpu_normals.py tests/test_gpu_parity.py -m gpu -x -q
# round-3 planner knobs: odd cuts of the target (a chip of 48 or 200 resident workgroups makes the split model pick 3, 5, 7, 11 ...
# runs of tiles; the split model off; every tile of a small launch in 32 runs, or whole), and each wave blocking of the filter's resolve
FORCED="tests/test_gpu_fused_filter.py tests/test_gpu_parity.py tests/test_gpu_filter_bounds.py tests/test_gpu_reference_mode.py tests/test_gpu_reference_random.py tests/test_gpu_upstream_variant.py tests/test_gpu_multi_device.py tests/test_gpu_state_fuzz.py"
MI355_ICP_SCAN_SLOTS=48 step 600 pytest_gpu_slots48.txt python -m pytest $FORCED -m gpu -x -q
MI355_ICP_SCAN_SLOTS=200 MI355_ICP_RESOLVE_WAVE_POINTS=32 step 600 pytest_gpu_slots200.txt python -m pytest $FORCED -m gpu -x -q
MI355_ICP_SPLIT_MODEL=0 MI355_ICP_RESOLVE_WAVE_POINTS=64 MI355_ICP_MAX_SUB_SPLIT=1 step 600 pytest_gpu_model0.txt python -m pytest $FORCED -m gpu -x -q
MI355_ICP_SPLIT_MODEL_PAIRS=1e30 MI355_ICP_RESOLVE_WAVE_POINTS=16 MI355_ICP_SUB_SPLIT_WGS=100000 step 600 pytest_gpu_model_all.txt python -m pytest $FORCED -m gpu -x -q
# round-4 knobs: the ticket-queue layout of the matrix-core scan (off by default), the table-wide picky exchange of slab-sharded
# reference mode, the service path with every pass pruned
MI355_ICP_SCAN_QUEUE=1 step 600 pytest_gpu_queue.txt python -m pytest $FORCED -m gpu -x -q
MI355_ICP_SCAN_QUEUE=1 MI355_ICP_QUEUE_MEMBERS=3 MI355_ICP_QUEUE_PAIRS=1e30 step 600 pytest_gpu_queue3.txt python -m pytest $FORCED -m gpu -x -q
MI355_ICP_PICKY_OFFERS=0 MI355_ICP_SERVICE_PRUNE=1 step 600 pytest_gpu_offers0_prune1.txt python -m pytest tests/test_gpu_reference_mode.py tests/test_gpu_upstream_variant.py tests/test_gpu_multi_device.py tests/test_gpu_distributed.py -m gpu -x -q
fi
if [ "$PART" != 2 ]; then
step 300 smoke.txt python -c "import __graft_entry__ as g; g.smoke()"
step 400 bench.txt python bench.py --steps "${STEPS:-3}" --warmup 1
fi
for f in gpurun_out/pytest_gpu*.txt gpurun_out/smoke.txt; do [ -f "$f" ] && echo "$(basename $f): $(tail -1 $f)"; done; tail -1 gpurun_out/bench.txt 2>/dev/null | cut -c1-200; exit 0
tail -3 gpurun_out/pytest_gpu.txt; tail -2 gpurun_out/pytest_gpu_fused.txt; tail -2 gpurun_out/pytest_gpu_xfilter.txt; tail -2 gpurun_out/pytest_gpu_xbf16.txt; tail -2 gpurun_out/pytest_gpu_exact.txt; tail -2 gpurun_out/pytest_gpu_pruned.txt; tail -2 gpurun_out/pytest_gpu_knn_brute.txt; tail -2 gpurun_out/pytest_gpu_plain_passes.txt; tail -1 gpurun_out/pytest_gpu_slots48.txt; tail -1 gpurun_out/pytest_gpu_slots200.txt; tail -1 gpurun_out/pytest_gpu_model0.txt; tail -1 gpurun_out/pytest_gpu_model_all.txt; tail -2 gpurun_out/smoke.txt; tail -1 gpurun_out/bench.txt
