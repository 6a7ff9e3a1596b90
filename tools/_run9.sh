set -u
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout -k 10 200 python tools/ref_kernel_phases.py 2>&1 | awk 'NR%2==1' | tail -17
for rep in 1 2; do for lib in pose_estimation_amd/lib/libmi355_icp.so build/varBase/libmi355_icp.so; do echo "== $lib"; MI355_ICP_LIB=$PWD/$lib timeout -k 10 200 python tools/refmode_latency.py --reps 60 | cut -c1-200; done; done
