#!/bin/bash
# Round-1 A/B of the all-exact scan: register blocking (2/4/8 points per lane) and workgroup-count goal by launch size (tools/scan_ab.py).
set -u
mkdir -p gpurun_out
{
timeout -k 10 200 python tools/scan_ab.py --rounds 3 "PPL=8" "PPL=4" "PPL=8,TARGET_BLOCKS=6144" "PPL=8,TARGET_BLOCKS=24576" "PPL=4,TARGET_BLOCKS=6144" || exit 99
timeout -k 10 200 python tools/scan_ab.py --n 100000 --rounds 5 "PPL=8" "PPL=4" "PPL=2" "PPL=4,MIN_TILES=1" "PPL=8,MIN_TILES=1" "PPL=4,MIN_TILES=4" "PPL=4,TARGET_BLOCKS=6144" "PPL=4,TARGET_BLOCKS=24576" "PPL=2,TARGET_BLOCKS=24576" || exit 99
timeout -k 10 200 python tools/scan_ab.py --n 50000 --jobs 64 --rounds 3 "PPL=8" "PPL=4" "PPL=2" "PPL=4,TARGET_BLOCKS=24576" "PPL=8,TARGET_BLOCKS=24576" || exit 99
timeout -k 10 200 python tools/scan_ab.py --n 1000 --rounds 5 "PPL=8" "PPL=4" "PPL=2" || exit 99
} > gpurun_out/ab_sizes.txt 2>&1
cat gpurun_out/ab_sizes.txt
