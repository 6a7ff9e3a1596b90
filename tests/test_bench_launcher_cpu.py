"""bench.py's own rank launcher: a plain `python bench.py --gpus N` with no WORLD_SIZE must start
N rank processes itself, rendezvous them, and relay rank 0's single JSON line; a rank count that does not match
--gpus must end non-zero instead of printing a mislabelled line.  Driven here with 2 gloo ranks and no GPU
(`--launcher-selftest` replaces the GPU body of the ranks; the spawn / environment / relay code is the real one)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


@pytest.mark.timeout(300)
def test_plain_invocation_spawns_the_ranks_and_relays_one_json_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-selftest"], env=dict(_clean_env(), MI355_BENCH_SKIP_DEVICE_CHECK="1"),
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["launcher_selftest"] is True and out["n_ranks"] == 2
    assert out["local_ranks_plus_1"] == [1, 2]          # rank r ran with LOCAL_RANK r: one device per rank


def test_world_size_that_contradicts_gpus_is_refused():
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "mislabelled" in r.stderr and not r.stdout.strip()


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_launch():
    sys.path.insert(0, ROOT)
    import bench
    # rank bodies that die at once (unknown flag): the launcher must report failure, not success with no line
    code = bench.launch_ranks(2, ["--gpus", "2", "--no-such-flag"], check_devices=False)
    assert code == 1


@pytest.mark.timeout(120)
def test_one_dead_rank_does_not_leave_the_others_waiting():
    """Rank 1 dies at once, rank 0 would sit in the rendezvous for minutes: the launcher must end the run promptly."""
    sys.path.insert(0, ROOT)
    import time
    import bench
    t0 = time.time()
    code = bench.launch_ranks(2, ["--gpus", "2", "--launcher-selftest"], extra_env={"MI355_BENCH_SELFTEST_DIE_RANK": "1"}, check_devices=False)
    assert code == 1 and time.time() - t0 < 60


def test_more_gpus_than_visible_is_refused_before_any_rank_starts(monkeypatch, capsys):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: 1)
    assert bench.launch_ranks(2, ["--gpus", "2"]) == 2
    assert "only 1 GPU(s) are visible" in capsys.readouterr().err
