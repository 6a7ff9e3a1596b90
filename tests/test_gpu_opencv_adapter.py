"""The cv::Mat adapter end to end on the GPU (against the test doubles of tests/stubs/opencv2/, see test_opencv_adapter_cpu.py):
N x 6 model, M x 6 scene, three Pose3DPtr refined in place by mi355::cv_adapter::ICP(250, 0.005f, 2.5f, 8) must come out
exactly as IcpEngine.align_reference leaves them on the same inputs, each Pose3D updated once, votes untouched."""
import subprocess

import numpy as np
import pytest

import oracle_lib as ora
import synth_twin as st
from test_opencv_adapter_cpu import build

pytestmark = pytest.mark.gpu


def test_adapter_refines_candidates_like_the_engine_and_the_oracle(engine):
    import pose_estimation_amd as pa
    n_model, n_scene = 6000, 9000
    out = subprocess.run([build(), str(n_model), str(n_scene)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    poses, residual, updates, votes = {}, {}, {}, {}
    for line in out.stdout.splitlines():
        f = line.split()
        if f[0] == "pose":
            poses[int(f[1])] = np.array([float(v) for v in f[2:]]).reshape(4, 4)
        elif f[0] == "residual":
            residual[int(f[1])], updates[int(f[1])], votes[int(f[1])] = float(f[2]), int(f[4]), int(f[6])
    assert sorted(poses) == [0, 1, 2] and all(updates[k] == 1 for k in poses) and [votes[k] for k in range(3)] == [100, 101, 102]
    T = np.array([[0.9961946980917455, -0.0716791773074155, 0.0491085115000797, 0.02],
                  [0.0738010430565223, 0.9963425807891292, -0.0428275146684137, -0.01],
                  [-0.0458590115510097, 0.0462887393339173, 0.9978748139193905, 0.03], [0, 0, 0, 1.0]])
    scene = pa.synth_cloud(st.TARGET_SEED, n_scene, normals=True)
    model = pa.synth_cloud(st.SOURCE_SEED, n_model, noise_seed=st.NOISE_SEED, sigma=st.SIGMA, transform=T, normals=True)
    inits = []
    for k in range(3):
        P = np.eye(4)
        P[0, 3] = 0.002 * k
        inits.append(P)
    engine.set_targets([scene])
    engine.set_sources([model])
    # the adapter's default is the upstream-faithful form of reference mode (MI355_ICP_FLAG_UPSTREAM)
    res = engine.align_reference(np.stack(inits), 250, 0.005, 2.5, 8, source_ids=[0] * 3, target_ids=[0] * 3, flags=pa._lib.FLAG_UPSTREAM)
    for k in range(3):
        assert np.array_equal(poses[k], res.poses[k]) and residual[k] == res.residual[k], k       # the same library underneath: same bits
        o = ora.icp_reference_upstream(model, scene, inits[k], 250, 0.005, 2.5, 8, nthreads=ora.max_threads())
        assert np.linalg.norm(poses[k] - o["pose"]) <= 1e-5 and o["passes"] == res.passes[k]
        assert abs(residual[k] - o["residual"]) <= 1e-9 * max(1.0, o["residual"])
