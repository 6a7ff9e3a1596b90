set -u
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in build/varT1024/libmi355_icp.so build/varT512w6/libmi355_icp.so build/varT768w6/libmi355_icp.so; do echo "== $lib"; MI355_ICP_LIB=$PWD/$lib timeout -k 10 300 python tools/bench_scene_filters.py 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    try:
        d=json.loads(l); print(f\"  {d['workload'][:60]:60s} {d['kernel_ms_best']*1e3:8.1f} us  {d['achieved_GBs']:7.0f} GB/s  {d['frac_of_8TBs']:.3f}\")
    except Exception: print(l.rstrip())
"; done; done

