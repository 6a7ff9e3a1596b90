#!/bin/bash
# mid-size scan efficiency study: how scan time scales with Ns, Nt around 100k, plus a PMC pass at 100k
set -u
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
{
for shape in "100000 100000" "100000 200000" "100000 400000" "200000 100000" "400000 100000" "200000 200000" "400000 400000"; do
  set -- $shape
  timeout -k 10 120 python tools/scan_ab.py --n $1 --nt $2 --rounds 5 "PPL=2" "PPL=4" "PPL=8" || exit 99
done
} > gpurun_out/midsize.txt 2>&1
cat gpurun_out/midsize.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_mid/pmc -- python3 $R/tools/scan_ab.py --n 100000 --rounds 3 "PPL=2" > $R/gpurun_out/prof_mid.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_mid/trace -- python3 $R/tools/scan_ab.py --n 100000 --rounds 3 "PPL=2" >> $R/gpurun_out/prof_mid.log 2>&1
tail -3 $R/gpurun_out/prof_mid.log
