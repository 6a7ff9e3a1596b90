#!/bin/bash
# PMC pass over tools/bench_normals.py (k-NN + PCA normals), summarised for the k-NN kernel.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_normals
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 $R/tools/bench_normals.py > "$OUT/$name.log" 2>&1
  echo "rc=$? $name"
}
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
run sq2 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in ("sq1", "sq2"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "knn_wave_kernel" in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        # tools/bench_normals.py: 20k, 100k, 500k, 1M points, a warm-up launch and a timed one each
        for tag, which in (("100k points", 3), ("1M points", len(ids) - 1)):
            if which >= len(ids):
                continue
            print(d, "knn_wave_kernel,", tag)
            for r in rows:
                if int(r["Dispatch_Id"]) == ids[which]:
                    print(f"   {r['Counter_Name']:24s} {float(r['Counter_Value']):16.0f}")
PY
