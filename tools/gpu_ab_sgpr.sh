#!/bin/bash
# Round-1 A/B of the all-exact scan: scalar-load (s_load) target reads against LDS tiles, by register blocking and size (tools/scan_ab.py).
set -u
mkdir -p gpurun_out
{
timeout -k 10 200 python tools/scan_ab.py --rounds 3 "PPL=8" "MODE=sgpr,PPL=8" "MODE=sgpr,PPL=4" "MODE=sgpr,PPL=2" || exit 99
timeout -k 10 200 python tools/scan_ab.py --n 100000 --rounds 8 "PPL=2" "MODE=sgpr,PPL=2" "MODE=sgpr,PPL=4" "MODE=sgpr,PPL=8" "MODE=sgpr,PPL=2,MIN_TILES=1" || exit 99
timeout -k 10 200 python tools/scan_ab.py --n 50000 --jobs 64 --rounds 3 "PPL=2" "MODE=sgpr,PPL=2" "MODE=sgpr,PPL=4" "MODE=sgpr,PPL=8" || exit 99
} > gpurun_out/ab_sgpr.txt 2>&1
cat gpurun_out/ab_sgpr.txt
