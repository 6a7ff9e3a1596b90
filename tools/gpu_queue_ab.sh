set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04g_queue_ab; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/suite.log
for rep in 1 2; do
 for q in 1 0; do
  MI355_ICP_SCAN_QUEUE=$q timeout -k 10 200 python tools/bench_configs.py > $O/cfg_q${q}_$rep.jsonl 2>$O/cfg_q${q}_$rep.err || exit 9
  MI355_ICP_SCAN_QUEUE=$q timeout -k 10 100 python tools/refmode_latency.py --reps 40 > $O/lat_q${q}_$rep.json 2>>$O/err.txt || exit 9
 done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04g_queue_ab/cfg_*.jsonl")):
    rows=[json.loads(l) for l in open(f) if l.startswith("{")]
    print(f.split("/")[-1], " ".join(f"{r['config'].split()[0]} {r['ms_per_iteration']:.4f}" for r in rows))
for f in sorted(glob.glob("gpurun_out/r04g_queue_ab/lat_*.json")):
    r=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], r["wall_ms_median"], r["profiled_scan_ms_total"], r["profiled_other_kernels_ms_total"])
PY
