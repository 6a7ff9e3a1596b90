#!/bin/bash
# Real cycles of the register-only loops of tools/ubench_bf16_min.hip: GRBM_GUI_ACTIVE per dispatch (the counter the scan kernel's
# "cycles per MFMA" is computed from), so that the loop and the kernel are compared in the same unit.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_ubench
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d "$OUT/p" -- "$R/build/ubench_bf16_min" > "$OUT/run.log" 2>&1
echo "rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -- "$R/build/ubench_bf16_min" > "$OUT/trace.log" 2>&1
echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.OrderedDict()
for f in glob.glob(f"{out}/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-12:], int(r["Grid_Size"]))
        rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
dur = {}
for f in glob.glob(f"{out}/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
iters = 100000
for (d, k, g), c in rows.items():
    wps = g // 256 // 256
    full = d % 2 == 0  # every run<> launches a quarter-length warm-up first, then the timed launch
    steps = 4.0 * (iters if full else iters // 4) * wps
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0  # summed over the 8 XCDs
    print(f"dispatch {d:3d} {k:12s} grid {g:7d} waves/SIMD {wps}  GUI_ACTIVE/8 {cyc:12.0f}  cycles/step/SIMD {cyc / steps:6.2f}  MFMA {c.get('SQ_INSTS_MFMA', 0):.3g}  us {dur.get(d, 0):9.1f}  -> clock {cyc / max(dur.get(d, 1), 1e-9) * 1e-3:5.2f} GHz")
PY
