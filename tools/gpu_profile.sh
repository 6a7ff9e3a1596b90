#!/bin/bash
# rocprofv3 passes over `bench.py` on the GPU box (one trace pass + separate PMC passes, never combined).
# Outputs land in gpurun_out/prof_<tag>/; copy the summaries worth keeping into profiles/.
set -u
TAG=${TAG:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
run() {  # run <name> <rocprof args...>
  local name=$1; shift
  echo "== $name" | tee -a "$OUT/steps.log"
  timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc $name" | tee -a "$OUT/steps.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping" | tee -a "$OUT/steps.log"; exit 99; fi
}
: > "$OUT/steps.log"
run trace --kernel-trace --stats
run pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run pmc_sq2 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run pmc_lds --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
find "$OUT" -name "*.csv" | head -30
