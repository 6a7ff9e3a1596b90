#!/usr/bin/env python3
"""bench.py — correspondence-pairs/s and ICP iterations/s of the MI355X alignment engine.

Contract: `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the driver launches it under
torch.distributed.run, one rank per GPU (RCCL exchange).  Run plainly (no WORLD_SIZE in the environment) with --gpus N > 1 it
stays ONE process and drives the N devices through ONE multi-device handle (mi355_icp_create_multi: the shape the reference's
single node process can use - pose_estimation.hpp:160-162 - with the per-pass rows exchanged through pinned host memory in
device order, no RCCL, no torch); `--spawn-ranks` keeps the older behaviour of starting N rank processes itself - fresh
children, started before this process has touched torch or HIP - and relaying rank 0's JSON line.  `n_gpus` in that line is
what the communicator (ncclCommCount) or the handle (its device count) reports, and a run whose count differs from
--gpus exits non-zero instead of printing a mislabelled line.  A *step* is one ICP iteration: source transform + exact brute-force NN scan +
covariance accumulation + (N > 1: one RCCL exchange of 17 doubles per rank: all-gather + rank-ordered sum) + host Kabsch/convergence, all
through `mi355_icp_align`.  Clouds are uploaded before the timed region (inputs resident in HBM).

Workload (BASELINE.json): N = 1 is config[2] "1M<->1M synthetic scene, 1 MI355X"; N > 1 keeps one
2^20-point source slab per GPU against the replicated 2^20-point target (weak scaling), so N = 8 is
config[4] "8M<->1M, source slab-sharded across 8 GPUs, per-iteration RCCL all-reduce over xGMI".
A *pair* is one (source point, target point) distance evaluation: pairs = Ns * Nt * iterations
(SURVEY.md 8d); `correspondences_per_s` (= Ns * iterations / s) is reported next to it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VECTOR_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (vector)" (= f32 MFMA dense peak)
BF16_MFMA_PEAK_TFLOPS = 2500.0    # same guide, "BF16/F16 ~2.5 PF dense"
HBM_PEAK_GBS = 8000.0             # same guide, HBM3E spec
FLOP_PER_PAIR = 8                 # 3 sub + 3 mul + 2 add (SURVEY.md 8d FLOP model; the min is not counted)
PEAK_CLOCK_HZ = 2.4e9
N_SIMD = 256 * 4

# The dominant kernel by scan mode (csrc/icp_engine.hip make_plan_on; MI355_ICP_SCAN_MODE forces one).  Each entry:
# kernel, roofline bound, peak (TFLOP/s), flop EXECUTED per evaluated pair on that unit, vector-issue slots per pair.
SCAN_MODELS = {
    "xbf16": ("nn_scan_xbf16_kernel (split-bf16 expansion filter on the matrix cores) + exact resolve", "mfma",
              BF16_MFMA_PEAK_TFLOPS, 32.0, 9.0 / 1024.0,
              "Every pair is evaluated by one K=16 bf16 MFMA contraction (32 flop per pair: the expansion surrogate of "
              "points quantised to two bfloat16 pieces per coordinate; the three low x low slots hold zeros - less energy per "
              "MFMA, a higher clock: profiles/r04u_zero_slots_ab.txt) and one vector issue slot per 128 pairs per lane for "
              "the minimum (8 v_min3_u32 + the MFMA issue per 1024 pairs: the binding resource, see roofline_issue; a "
              "register-only loop of that mix takes 44.8 cycles per MFMA, the kernel 49.5 - and issues no MFMA whose result is dropped: "
              "profiles/r03s_pmc_summary.txt; round 2: 49.7 per issued, 51.3 per useful MFMA); "
              "the exact unfused float32 formula then decides inside the one or two candidate granules, so indices and "
              "distances stay bit-identical to the oracle, and the filter's lower bound is itself a tested invariant "
              "(DESIGN.md 3.1). `achieved` counts the 32 executed MFMA flop per pair; by SURVEY 8d's 8-flop model the same "
              "run is `model_tflops`, above the FP32 vector peak (157.3) that model was priced against; north_star's "
              "literal no-MFMA kernel is timed in this same run as `roofline_exact`."),
    "xfilter": ("nn_scan_xfilter_kernel (expansion filter, 4 VALU issue slots per pair) + exact resolve", "valu",
                FP32_VECTOR_PEAK_TFLOPS, 6.0, 4.0 / 64.0,
                "FP32 vector ALU bound: 3 fma + 1 min per pair on tile-centred coordinates; exact resolve as above."),
    "fused": ("nn_scan_fused_kernel (7 VALU issue slots per pair) + exact resolve", "valu", FP32_VECTOR_PEAK_TFLOPS, 8.0, 7.0 / 64.0,
              "FP32 vector ALU bound: 3 sub + 1 mul + 2 fma + 1 min per pair."),
    "exact": ("nn_scan_kernel (all-exact, 9 VALU issue slots per pair)", "valu", FP32_VECTOR_PEAK_TFLOPS, 8.0, 9.0 / 64.0,
              "FP32 vector ALU bound: the unfused contract formula, 3 sub + 3 mul + 2 add + 1 min per pair."),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-source", type=int, default=1 << 20, help="source points PER GPU")
    ap.add_argument("--n-target", type=int, default=1 << 20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-iters", type=int, default=20, help="ICP iterations of the CPU baseline sample (about 3 s on 16 threads)")
    ap.add_argument("--cpu-baseline-iters-1thread", type=int, default=1, help="... of its 1-thread leg (about 2 s)")
    ap.add_argument("--no-exact-leg", action="store_true", help="skip the short all-exact-scan leg (roofline_exact)")
    ap.add_argument("--exact-leg-steps", type=int, default=3)
    ap.add_argument("--no-config-legs", action="store_true", help="skip the short legs for the other BASELINE configurations (other_configs)")
    ap.add_argument("--launcher-selftest", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--spawn-ranks", action="store_true",
                    help="plain invocation with --gpus N > 1: start N rank processes (RCCL) instead of one multi-device handle in this process")
    ap.add_argument("--device-ids", type=int, nargs="+", default=None,
                    help="multi-device handle: the device of every slot (default 0 .. N-1; ids may repeat, e.g. 0 0 on a one-GPU box)")
    ap.add_argument("--collective", choices=["native", "torch"], default="native",
                    help="N > 1: the per-iteration exchange issued inside the engine over RCCL (native) or by torch.distributed")
    ap.add_argument("--force-dist", action="store_true", help="exercise the distributed code path even at N = 1")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): --n-source points PER GPU, so N = 8 is BASELINE config[4] 8M<->1M; strong: "
                         "--n-source points in total, slab-sharded over the ranks")
    return ap.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(src, tgt, iters, iters_1thread):
    """CPU restatement of point-to-point ICP (KD-tree), timed on this host: oracle/icp_oracle.c.
    Never labelled PCL/OpenCV: neither exists here.  Headline = ICP iterations/s on all the cores this
    job owns; the 1-thread figure and the CPU model are printed beside it.  The tree answers a
    query in O(log Nt): Ns*Nt "pairs" per iteration is only the brute-force-EQUIVALENT count, kept under that name."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ora
    # the GPU box gives a 1-GPU job a 16-core share of its host (more threads than that would be borrowed)
    threads = max(1, min(ora.max_threads(), len(os.sched_getaffinity(0)), 16))

    def timed(n_iter, nth):
        p = ora.params(max_iterations=n_iter)
        t0 = time.perf_counter()
        r = ora.icp(src, tgt, np.eye(4), p, use_kdtree=True, nthreads=nth)
        return r, time.perf_counter() - t0

    r, dt = timed(iters, threads)
    r1, dt1 = timed(iters_1thread, 1)
    return {
        "value": r["iterations"] / dt, "unit": f"ICP iterations/s ({src.shape[0]}x{tgt.shape[0]}, exact KD-tree NN)",
        "cores": threads, "kind": "port", "cpu_model": cpu_model(), "host_cores_visible": len(os.sched_getaffinity(0)),
        "iters_per_s": r["iterations"] / dt, "iters_per_s_1thread": r1["iterations"] / dt1,
        "brute_force_equivalent_pairs_per_s": float(src.shape[0]) * float(tgt.shape[0]) * r["iterations"] / dt,
        "sample": f"{r['iterations']} ICP iterations of the same {src.shape[0]}x{tgt.shape[0]} workload on {threads} threads in "
                  f"{dt:.2f} s + {r1['iterations']} iterations on 1 thread in {dt1:.2f} s, KD-tree build included in both; "
                  "O(Ns log Nt) exact tree search, NOT a brute-force scan - compare iterations/s, not pairs/s; "
                  "threads are capped at 16, this pool's CPU share per GPU (host_cores_visible says what the box has)",
        "pose": r["pose"],
    }


def other_config_legs(pa, st, np, main_engine):
    """Short legs for the other single-GPU configurations of BASELINE.json, in the same driver-timed run (the headline line
    stays config[2]; these are extra keys): config[1] 100k<->100k x 50 iterations, config[3] 64 x (50k<->50k) batched, the
    1k<->1k stand-in of config[0], and the reference's own request shape - <= 6 candidates of one model against one scene
    through cv::ppf_match_3d::ICP(250, 0.005f, 2.5f, 8) (opencv_surface_match.cpp:85-94), i.e. mi355_icp_align_reference.
    Un-profiled host wall clock on resident clouds, one warm-up call each.  tools/bench_configs.py runs the same
    configurations against the oracle (parity), tools/refmode_latency.py the request shape with its kernel split."""
    legs = {}

    def p2p(name, n_jobs, ns, nt, iters):
        pairs = [st.make_pair(pa.synth_cloud, ns, nt, object_index=(k if n_jobs > 1 else -1)) for k in range(n_jobs)]
        with pa.IcpEngine(main_engine.device_id) as e:
            e.set_targets([q[1] for q in pairs])
            e.set_sources([q[0] for q in pairs])
            ids = list(range(n_jobs))
            init = np.stack([np.eye(4)] * n_jobs)
            e.align(init, pa.default_params(max_iterations=2), ids, ids)
            t0 = time.perf_counter()
            r = e.align(init, pa.default_params(max_iterations=iters), ids, ids)
            dt = time.perf_counter() - t0
        assert all(int(x) == 0 for x in r.status) and all(int(x) == iters for x in r.iterations)
        legs[name] = {"jobs": n_jobs, "n_source": ns, "n_target": nt, "iterations": iters, "ms_per_iteration": 1e3 * dt / iters,
                      "pairs_per_s": n_jobs * float(ns) * nt * iters / dt,
                      "pose_error_vs_truth_max": float(max(np.linalg.norm(r.poses[k] - pairs[k][2]) for k in range(n_jobs)))}

    p2p("config0_stand_in_1k_x_1k", 1, 1000, 1000, 50)
    p2p("config1_100k_x_100k_50_iterations", 1, 100_000, 100_000, 50)
    p2p("config3_64_objects_50k_x_50k", 64, 50_000, 50_000, 20)
    # the reference's request: 6 candidate poses, one 20k-point model (N x 6, as the reference's models), one 100k-point scene
    # with normals.  The form the host mirrors run by default - MI355_ICP_FLAG_UPSTREAM, upstream's control flow - and, beside
    # it, the round 1-3 form on the same clouds (flags 0; what BENCH_r03 and earlier timed under this key)
    T = st.perturbation(-1)
    gt = np.linalg.inv(T)
    scene = pa.synth_cloud(st.TARGET_SEED, 100_000, normals=True)
    model = pa.synth_cloud(st.SOURCE_SEED, 20_000, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T, normals=True)
    inits = np.stack([np.eye(4)] + [st.rigid_transform(k, (0.2, 1, -0.4), (0.01, -0.02, 0.015)) @ gt for k in (1.0, 2.0, 3.0, 4.0, 5.0)])
    with pa.IcpEngine(main_engine.device_id) as e:
        e.set_targets([scene])
        e.set_sources([model])
        for form, flags, calls in (("upstream", pa._lib.FLAG_UPSTREAM, 20), ("round1to3", 0, 10)):
            for _ in range(3):
                r = e.align_reference(inits, 250, 0.005, 2.5, 8, flags=flags)
            times = []
            for _ in range(calls):
                t0 = time.perf_counter()
                r = e.align_reference(inits, 250, 0.005, 2.5, 8, flags=flags)
                times.append(time.perf_counter() - t0)
            times.sort()
            leg = {"call": f"mi355_icp_align_reference(250, 0.005, 2.5, 8, flags = {'MI355_ICP_FLAG_UPSTREAM' if flags else '0'})",
                   "ms_per_request_median": 1e3 * times[len(times) // 2], "ms_per_request_min": 1e3 * times[0], "calls": len(times),
                   "passes": [int(x) for x in r.passes], "pose_error_vs_truth_max": float(max(np.linalg.norm(q - gt) for q in r.poses))}
            if form == "upstream":
                legs["reference_request_6_candidates_20k_model_100k_scene"] = leg
            else:
                legs["reference_request_6_candidates_20k_model_100k_scene"]["round1to3_form"] = leg
    # the one HBM-bound kernel of the path: scene compaction (reference pose_estimation.cpp:248, NaN removal) over a frame far
    # beyond the 256 MB Infinity Cache; kernel time from the engine's own HIP events, algorithmic bytes = 12 B read per point +
    # 12 B written per kept point
    n = 64_000_000
    pts = np.empty((n, 3), dtype=np.float32)
    ramp = (np.arange(n, dtype=np.uint32) % np.uint32(4099)).astype(np.float32)
    pts[:, 0] = ramp * np.float32(1e-3)
    pts[:, 1] = ramp * np.float32(-7e-4)
    pts[:, 2] = np.float32(1.0)
    pts[::5, 0] = np.nan   # 80 % kept
    with pa.IcpEngine(main_engine.device_id) as e:
        e.set_profiling(True)
        best, kept = None, 0
        for rep in range(3):
            e.scene_load(pts)
            e.reset_profile()
            kept = e.scene_filter([(pa._lib.SCENE_FILTER_FINITE, [])])
            ms = e.profile().scene_filter_ms
            best = ms if best is None or (rep and ms < best) else best
    del pts
    alg = 12.0 * n + 12.0 * kept
    legs["scene_compaction_64M_points_nan_removal"] = {
        "kernel": "scene_compact_kernel", "points": n, "kept": int(kept), "kernel_ms_best_of_3": best, "algorithmic_bytes": alg,
        "roofline": {"bound": "hbm", "achieved": alg / (best * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg / (best * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    # a whole EstimatePose request through the C++ host mirror (build/estimate_pose_harness ... pipeline): camera-like frame
    # -> NaN removal -> RANSAC plane fit -> band filter -> k = 20 normals -> commit -> reference-mode ICP of 6 candidates ->
    # best pose; its third pass is the steady state of a service.  A child process (never an exec of this one).
    exe = os.path.join(ROOT, "build", "estimate_pose_harness")
    if os.path.exists(exe):
        import re
        import subprocess
        try:
            out = subprocess.run([exe, "20000", "100000", "6", "pipeline"], capture_output=True, text=True, timeout=120)
            passes = [ln for ln in out.stdout.splitlines() if ln.startswith("pipeline pass")]
            m = re.search(r"upload\+filters\+plane fit ([0-9.]+) ms, normals ([0-9.]+) ms, commit ([0-9.]+) ms, ICP of (\d+) candidates ([0-9.]+) ms, total ([0-9.]+) ms", passes[-1]) if passes else None
            if out.returncode == 0 and m:
                legs["whole_request_pipeline_150k_frame_20k_model"] = {
                    "command": "build/estimate_pose_harness 20000 100000 6 pipeline (third pass: steady state)",
                    "filters_and_plane_fit_ms": float(m.group(1)), "normals_ms": float(m.group(2)), "commit_ms": float(m.group(3)),
                    "icp_candidates": int(m.group(4)), "icp_ms": float(m.group(5)), "total_ms": float(m.group(6)),
                    "icp_form": "upstream-faithful (the mirror's default)"}
            else:
                legs["whole_request_pipeline_150k_frame_20k_model"] = {"error": (out.stderr or out.stdout)[-300:]}
        except Exception as ex:   # the headline line must not depend on this leg
            legs["whole_request_pipeline_150k_frame_20k_model"] = {"error": repr(ex)}
    return legs


def visible_gpu_count():
    """GPUs this process may use, WITHOUT touching HIP or torch (the launcher must stay GPU-free): the KFD topology's
    nodes that have SIMDs, cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set.  None if it cannot be told."""
    import glob
    n = 0
    try:
        for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            for line in open(f):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
    except OSError:
        return None
    if n == 0:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n, argv, extra_env=None, check_devices=True):
    """Plain `python bench.py --gpus N` (N > 1, no WORLD_SIZE): start N fresh rank processes - one per device,
    RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set - and relay rank 0's stdout.  Called before this process imports torch or
    touches HIP: children are ordinary child processes, never a re-exec of a GPU-initialised one.  A rank that dies takes
    the others with it (they would wait for it in the rendezvous for minutes).  Returns the exit code."""
    import socket
    import subprocess
    import tempfile
    if check_devices:
        have = visible_gpu_count()
        if have is not None and have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible to this process: refusing to start the ranks", file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    codes = [None] * n
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if any(c not in (None, 0) for c in codes):   # one rank failed: the rest would hang in the rendezvous / a collective
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
            break
        time.sleep(0.05)
    out0.seek(0)
    text = out0.read()
    if text:
        sys.stdout.write(text)
        sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py launcher: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def launcher_selftest():
    """Rank body of `--launcher-selftest`: rendezvous over gloo (no GPU), prove every rank is there, rank 0 prints one
    JSON line.  tests/test_bench_launcher_cpu.py drives it through launch_ranks with 2 ranks."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("MI355_BENCH_SELFTEST_DIE_RANK") == str(rank):
        raise SystemExit(3)   # tests/test_bench_launcher_cpu.py: a rank that dies before the rendezvous
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = torch.zeros(world, dtype=torch.int64)
    seen[rank] = 1 + int(os.environ["LOCAL_RANK"])
    dist.all_reduce(seen)
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_ranks": dist.get_world_size(), "local_ranks_plus_1": seen.tolist()}))
    dist.destroy_process_group()


def main():
    a = parse()
    multi_ids = None   # plain invocation with N > 1: ONE process, one multi-device handle
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if a.launcher_selftest or a.spawn_ranks:
            # nothing GPU-related has been imported yet: spawn the ranks and relay rank 0's line
            raise SystemExit(launch_ranks(a.gpus, sys.argv[1:], check_devices=not (a.launcher_selftest or os.environ.get("MI355_BENCH_SKIP_DEVICE_CHECK"))))
        multi_ids = a.device_ids if a.device_ids else list(range(a.gpus))
        if len(multi_ids) != a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but --device-ids names {len(multi_ids)} slots: refusing to print a mislabelled line")
        have = visible_gpu_count()
        if have is not None and max(multi_ids) >= have:
            print(f"bench.py: --gpus {a.gpus} (devices {multi_ids}) but only {have} GPU(s) are visible to this process", file=sys.stderr)
            raise SystemExit(2)
    if a.launcher_selftest:
        return launcher_selftest()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist_mode = world > 1 or a.force_dist
    if multi_ids is None and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: refusing to print a mislabelled line")
    n_slots = len(multi_ids) if multi_ids else world   # devices the workload is spread over
    torch = dist = None
    json_fd = 1
    if dist_mode:
        # librccl prints a version banner on STDOUT when its first communicator comes up; the contract is ONE JSON line
        # there.  Everything this process prints goes to stderr from here on, the line itself to the saved stdout.
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch  # loads the HIP runtime first so the engine binds to the same one
        import torch.distributed as dist
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"rank {rank}: --gpus {a.gpus} needs device {local_rank}, but only {torch.cuda.device_count()} are visible")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import numpy as np
    import pose_estimation_amd as pa
    from pose_estimation_amd import distributed as pd
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth_twin as st  # seeds / named configurations only

    nt = a.n_target
    ns_total = a.n_source * n_slots if a.scaling == "weak" else a.n_source
    ns_local = a.n_source if a.scaling == "weak" else (ns_total + n_slots - 1) // n_slots
    T = st.perturbation(-1)
    tgt = pa.synth_cloud(st.TARGET_SEED, nt)
    if multi_ids:   # the handle takes the WHOLE model and cuts the slabs itself
        begin, count = 0, ns_total
    else:
        begin, count = pa.slab_range(ns_total, rank, world)
    src = pa.synth_cloud(st.SOURCE_SEED, count, first_index=begin, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T)

    eng = pa.IcpEngine(device_ids=multi_ids) if multi_ids else pa.IcpEngine(local_rank)
    t_up = time.perf_counter()
    eng.set_targets([tgt])
    eng.set_sources([src])
    upload_s = time.perf_counter() - t_up

    collective = "in-process: per-device rows through pinned host memory, added in device order (mi355_icp_create_multi)" if multi_ids else "none"
    if dist_mode:
        if a.collective == "native":
            try:
                pd.attach_native_comm(eng)
                collective = "rccl-native"
            except Exception as e:  # keep the run alive: same path with torch issuing the all-reduce
                print(f"[rank {rank}] native RCCL attach failed ({e}); falling back to torch.distributed", file=sys.stderr)
                collective = "torch.distributed"
        else:
            collective = "torch.distributed"
        flag = torch.tensor([1 if collective == "rccl-native" else 0], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # all ranks must agree
        if flag.item() == 0 and collective == "rccl-native":
            eng.comm_detach()
            collective = "torch.distributed"

    def run(iters, profile):
        eng.set_profiling(profile)
        p = pa.default_params(max_iterations=iters)
        if collective == "torch.distributed":
            return pd.align_external_collective(pd.engine_pass(eng, [0], [0]), pd.torch_allreduce(device="cuda"),
                                                np.eye(4)[None], p)
        return eng.align(np.eye(4)[None], p)

    def fence():
        if dist_mode:
            dist.barrier()
            torch.cuda.synchronize()

    if a.warmup > 0:
        run(a.warmup, False)
    eng.reset_profile()
    fence()
    t0 = time.perf_counter()
    res = run(a.steps, True)   # align() returns after its stream has drained
    fence()
    elapsed = time.perf_counter() - t0
    if dist_mode:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert int(res.iterations[0]) == a.steps and int(res.status[0]) == 0, (res.iterations, res.status)

    prof = eng.profile()
    # the rank count the line reports is the communicator's own (ncclCommCount), not an environment variable
    if collective == "rccl-native" or multi_ids:
        n_ranks = eng.comm_info()[1]
    elif dist_mode:
        n_ranks = dist.get_world_size()
    else:
        n_ranks = 1
    if n_ranks != a.gpus:
        raise SystemExit(f"the communicator spans {n_ranks} ranks but --gpus is {a.gpus}: refusing to print a mislabelled line")
    traffic, traffic_note = None, None
    try:  # PMC counters cannot be read from inside this process; use the committed rocprofv3 pass of this same command
        tj = json.load(open(os.path.join(ROOT, "profiles", "scan_traffic.json")))
        if ns_local == (1 << 20) and nt == (1 << 20) and not os.environ.get("MI355_ICP_SCAN_MODE"):
            traffic, traffic_note = tj["traffic_bytes_per_launch"], f"profiles/scan_traffic.json ({tj['tag']}): {tj['method']}"
    except Exception:
        pass
    if rank == 0:
        pairs_step = float(ns_total) * float(nt)
        scan_ms = prof.nn_scan_ms / max(prof.nn_scan_launches, 1)
        count = ns_local if multi_ids else count   # points of ONE device's launch (the profile is device slot 0's)
        pairs_launch = float(count) * float(nt)
        alg_bytes = 12.0 * count + 12.0 * nt + 8.0 * count   # read both clouds once, write index + d2
        mode = os.environ.get("MI355_ICP_SCAN_MODE", "") or ("xbf16" if pairs_launch >= 2.5e8 else "exact")
        kernel, bound, peak, exec_flop, slots_per_pair, note = SCAN_MODELS.get(mode, SCAN_MODELS["xfilter"])
        out = {
            "metric": "correspondence_pairs_per_sec", "value": pairs_step * a.steps / elapsed, "unit": "pairs/s",
            "n_gpus": n_ranks, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "icp_iters_per_sec": a.steps / elapsed,
            "correspondences_per_sec": float(ns_total) * a.steps / elapsed,
            "config": {"workload": f"{ns_total}x{nt} point-to-point ICP, exact brute-force NN "
                                   f"({'BASELINE config[2] 1M<->1M' if n_slots == 1 else 'source slab-sharded, ' + str(ns_local) + ' pts/GPU, target replicated'})",
                       "process_model": "one process, one multi-device handle" if multi_ids else ("one process per GPU" if world > 1 else "one process, one GPU"),
                       "device_ids": multi_ids,
                       "n_source_total": ns_total, "n_target": nt, "source_points_per_gpu": ns_local,
                       "collective": collective, "pair_definition": "one (source,target) distance evaluation: Ns*Nt*iterations"},
            "roofline": {
                "bound": bound, "kernel": kernel,
                "achieved": exec_flop * pairs_launch / (scan_ms * 1e-3) / 1e12, "peak": peak,
                "unit": "TFLOP/s", "frac": exec_flop * pairs_launch / (scan_ms * 1e-3) / 1e12 / peak,
                "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_note,
                "traffic_from_this_run": False,   # PMC counters cannot be read in-process: the committed rocprofv3 pass of this command, or null
                "algorithmic_flop_per_launch": exec_flop * pairs_launch, "flop_per_pair": exec_flop,
                "model_flop_per_pair": FLOP_PER_PAIR, "model_tflops": FLOP_PER_PAIR * pairs_launch / (scan_ms * 1e-3) / 1e12,
                "avg_launch_ms": scan_ms, "launches": int(prof.nn_scan_launches), "note": note,
            },
            "roofline_issue": {
                "bound": "vector issue port", "unit": "pairs/s", "achieved": pairs_launch / (scan_ms * 1e-3),
                "peak": N_SIMD * PEAK_CLOCK_HZ / (4.0 * slots_per_pair) if mode == "xbf16" else N_SIMD * PEAK_CLOCK_HZ / (2.0 * slots_per_pair),
                "frac": pairs_launch / (scan_ms * 1e-3) / (N_SIMD * PEAK_CLOCK_HZ / ((4.0 if mode == "xbf16" else 2.0) * slots_per_pair)),
                "note": "arithmetic floor of the instruction mix: vector instructions per evaluated pair x 4 cycles per integer or "
                        "MFMA wave-instruction (float32: 2 with >= 2 waves per SIMD) at 2.4 GHz on 1024 SIMDs.  Measured on the "
                        "hardware the same mix costs more: 44.8 cycles per MFMA + 8 minima instead of 36 (tools/ubench_bf16_min.hip), "
                        "and the chip holds 1.95-2.1 GHz under this load - against THAT loop the kernel runs at 91 % in cycles (DESIGN.md 3.1); "
                        "the loop itself, on constant operands, holds 2.39 GHz (profiles/r04u_ubench_cycles.txt): what is left is clock, "
                        "i.e. energy per MFMA; "
                        "`traffic` counts Infinity-Cache hits too (the 33.5 MB operand image re-streamed per workgroup round and XCD)",
            },
            "roofline_hbm": {
                "bound": "hbm", "kernel": kernel, "achieved": alg_bytes / (scan_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": alg_bytes / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "note": "north_star asks for the HBM figure; the scan is compute-bound by ~4 orders of magnitude (DESIGN.md)",
            },
            "kernels_ms_per_step": {"nn_scan": scan_ms, "resolve_accumulate": prof.resolve_ms / max(prof.resolve_launches, 1),
                                    "final_reduce": prof.reduce_ms / max(prof.reduce_launches, 1)},
            "upload_s": upload_s, "device": eng.device_info()["name"],
            "final_pose_error_vs_truth": float(np.linalg.norm(res.poses[0] - np.linalg.inv(T))),
        }
        if n_slots == 1 and not a.no_exact_leg and mode != "exact":
            # north_star's literal kernel (LDS-tiled all-exact scan, no matrix cores) in the same driver-timed run: a
            # second engine with the planner forced to it, a few steps, priced against the FP32 vector peak
            os.environ["MI355_ICP_SCAN_MODE"] = "exact"
            try:
                ex = pa.IcpEngine(local_rank)
            finally:
                os.environ.pop("MI355_ICP_SCAN_MODE")
            ex.set_targets([tgt])
            ex.set_sources([src])
            ex.align(np.eye(4)[None], pa.default_params(max_iterations=1))
            ex.reset_profile()
            ex.set_profiling(True)
            t1 = time.perf_counter()
            rex = ex.align(np.eye(4)[None], pa.default_params(max_iterations=a.exact_leg_steps))
            ex_elapsed = time.perf_counter() - t1
            pe = ex.profile()
            ex.close()
            ex_ms = pe.nn_scan_ms / max(pe.nn_scan_launches, 1)
            k2, b2, peak2, flop2, _slots2, note2 = SCAN_MODELS["exact"]
            out["roofline_exact"] = {
                "bound": b2, "kernel": k2, "achieved": flop2 * pairs_launch / (ex_ms * 1e-3) / 1e12, "peak": peak2, "unit": "TFLOP/s",
                "frac": flop2 * pairs_launch / (ex_ms * 1e-3) / 1e12 / peak2, "avg_launch_ms": ex_ms, "launches": int(pe.nn_scan_launches),
                "steps": a.exact_leg_steps, "ms_per_step": 1e3 * ex_elapsed / a.exact_leg_steps, "pairs_per_s": pairs_step * a.exact_leg_steps / ex_elapsed,
                "flop_per_pair": flop2, "pose_vs_default_path_frobenius": float(np.linalg.norm(rex.poses[0] - eng.align(np.eye(4)[None], pa.default_params(max_iterations=a.exact_leg_steps)).poses[0])),
                "note": note2 + " This is north_star's literal form (target tiles through LDS, no MFMA); the default path above "
                                "returns the same bits faster.",
            }
        if n_slots == 1 and not a.no_config_legs:
            out["other_configs"] = other_config_legs(pa, st, np, eng)
        if n_slots == 1 and not a.no_cpu_baseline:
            cb = cpu_baseline(src, tgt, a.cpu_baseline_iters, a.cpu_baseline_iters_1thread)
            cpu_pose = cb.pop("pose")
            if a.cpu_baseline_iters == a.steps:
                out["pose_vs_cpu_frobenius"] = float(np.linalg.norm(res.poses[0] - cpu_pose))
            out["cpu_baseline"] = cb
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist_mode:
        if collective == "rccl-native":
            eng.comm_detach()
        eng.close()
        dist.destroy_process_group()
    else:
        eng.close()


if __name__ == "__main__":
    main()
