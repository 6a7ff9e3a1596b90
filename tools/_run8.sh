set -u
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_scene_filters.py tests/test_gpu_plane_fit.py tests/test_gpu_normals.py tests/test_gpu_opencv_adapter.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for lib in pose_estimation_amd/lib/libmi355_icp.so build/varBase/libmi355_icp.so; do echo "== $lib"; MI355_ICP_LIB=$PWD/$lib timeout -k 10 300 python tools/bench_scene_filters.py 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    try:
        d=json.loads(l); print(f\"  {d['workload'][:60]:60s} {d['kernel_ms_best']*1e3:8.1f} us  {d['achieved_GBs']:7.0f} GB/s  {d['frac_of_8TBs']:.3f}\")
    except Exception: print(l.rstrip())
"; done; done
timeout -k 10 300 python tools/bench_scene_filters.py > gpurun_out/scene_filter_bandwidth_r03l.jsonl
timeout -k 10 300 python tools/scene_timeline.py > gpurun_out/scene_timeline_r03l.txt 2>&1; sed -n 1,11p gpurun_out/scene_timeline_r03l.txt
./build/estimate_pose_harness 20000 100000 6 pipeline | grep "pipeline pass"
./build/estimate_pose_harness 20000 300000 6 pipeline | grep "pipeline pass"
