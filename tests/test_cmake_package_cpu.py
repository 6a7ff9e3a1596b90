"""The CMake package file (cmake/mi355_icpConfig.cmake) a maintainer of the reference would point find_package at: configure and
build a miniature consumer against it and run it (host-only entry points: no GPU needed)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_find_package_and_link(tmp_path):
    cmake = shutil.which("cmake")
    if cmake is None:
        pytest.skip("cmake is not installed")
    src = os.path.join(ROOT, "tests", "cmake_consumer")
    build = str(tmp_path / "build")
    cfg = subprocess.run([cmake, "-S", src, "-B", build, f"-Dmi355_icp_DIR={os.path.join(ROOT, 'cmake')}", "-DCMAKE_BUILD_TYPE=Release"],
                         capture_output=True, text=True, timeout=120)
    assert cfg.returncode == 0, cfg.stdout + cfg.stderr
    bld = subprocess.run([cmake, "--build", build], capture_output=True, text=True, timeout=200)
    assert bld.returncode == 0, bld.stdout + bld.stderr
    run = subprocess.run([os.path.join(build, "consumer")], capture_output=True, text=True, timeout=60)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "max_iterations 250" in run.stdout
