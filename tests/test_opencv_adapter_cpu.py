"""pose_estimation_amd/host/opencv_adapter.hpp - the drop-in for opencv_surface_match.cpp:85,94 - needs OpenCV headers, which
exist neither here nor on the GPU box.  So that it is at least COMPILED (round 1 shipped it never compiled), tests/cpp/
adapter_check.cpp builds it against the test doubles of tests/stubs/opencv2/ (the few cv::Mat / Matx44d / Pose3D members the
adapter touches; clearly not OpenCV), warnings as errors; without a GPU the program must fail loudly, not fall back."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "adapter_check")


def build():
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "stubs"), "-o", EXE,
                           os.path.join(ROOT, "tests", "cpp", "adapter_check.cpp"), "-L" + os.path.join(ROOT, "pose_estimation_amd", "lib"),
                           "-lmi355_icp", "-Wl,-rpath," + os.path.join(ROOT, "pose_estimation_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def _gpu_visible():
    try:
        return "gfx9" in subprocess.run(["rocminfo"], capture_output=True, text=True, timeout=20).stdout
    except Exception:
        return False


def test_adapter_compiles_and_fails_loudly_without_a_gpu():
    exe = build()
    text = open(os.path.join(ROOT, "pose_estimation_amd", "host", "opencv_adapter.hpp")).read()
    assert "tests/stubs" not in text and "stubs" not in text       # the product header knows nothing of the test doubles
    if not _gpu_visible():
        r = subprocess.run([exe, "100", "100"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and "no CPU fallback" in r.stderr
