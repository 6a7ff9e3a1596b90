#!/bin/bash
# PMC passes (separate from any trace pass) over one bench.py run with the split-bf16 matrix-core scan forced.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_xbf16
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MI355_ICP_SCAN_MODE=${MODE:-xbf16}
BENCH="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
run() {
  local name=$1; shift
  timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc $name"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping"; exit 99; fi
}
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run sq2 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in ("sq1", "sq2"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, cs in acc.items():
            if "scan" not in k and "pruned" not in k: continue
            print(d, k)
            for c, v in sorted(cs.items()):
                print(f"   {c:28s} {v / n[(k, c)]:16.1f} per launch ({n[(k, c)]} launches)")
PY
