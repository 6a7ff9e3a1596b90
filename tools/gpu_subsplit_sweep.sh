#!/bin/bash
# Sweep of the small-pass planner knobs on ONE box: reference request latency (tools/refmode_latency.py) for each
# (MI355_ICP_MAX_SUB_SPLIT, MI355_ICP_SUB_SPLIT_WGS) pair, two rounds so box drift shows.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-subsplit}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
: > "$O/summary.txt"
for rep in 1 2; do
  for cap in ${CAPS:-8 16 32}; do
    for wgs in ${WGS:-512 1024 2048 4096}; do
      MI355_ICP_MAX_SUB_SPLIT=$cap MI355_ICP_SUB_SPLIT_WGS=$wgs timeout -k 10 120 python tools/refmode_latency.py --reps 40 > "$O/lat_${cap}_${wgs}_$rep.json" 2> "$O/lat_${cap}_${wgs}_$rep.err"
      rc=$?
      if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping" | tee -a "$O/summary.txt"; exit 99; fi
      python3 -c "
import json,sys
d=json.loads(open('$O/lat_${cap}_${wgs}_$rep.json').read().strip().splitlines()[-1])
print('cap %2d wgs %5d rep $rep: median %.4f min %.4f ms  scan %.3f other %.3f' % ($cap,$wgs,d['wall_ms_median'],d['wall_ms_min'],d['profiled_scan_ms_total'],d['profiled_other_kernels_ms_total']))" | tee -a "$O/summary.txt"
    done
  done
done
