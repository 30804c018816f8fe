"""`python bench.py --gpus N` as typed, without a launcher: the parent spawns the N ranks before
it imports torch, they rendezvous, gather their tiles to rank 0 every step, time the steps between barriers with the
max over ranks, and rank 0's single JSON line comes back through the parent.  CPU rehearsal (--dry-run: gloo, the
render replaced by filling the owned pixels); the GPU path shares everything but the backend and the render call."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=240)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_bench_spawns_its_own_ranks(n):
    r = _run(["--gpus", str(n), "--steps", "3", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == n and out["steps"] == 3 and out["warmup"] == 1 and out["frames_assembled_correctly"] is True


def test_bench_under_a_launcher_environment_does_not_spawn_again():
    """RANK / WORLD_SIZE set (torch.distributed.run did the spawning): the process is a rank, not a parent"""
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--dry-run"],
             env_extra={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29512"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 1


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--dry-run"], env_extra={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_a_rank_that_dies_ends_the_whole_run_at_once():
    """one rank exits before the rendezvous (on a GPU box: an invalid device ordinal); the others wait for it in
    init_process_group for minutes -- the parent must stop them and report the status, not hang with them"""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run"], env_extra={"MIPT_BENCH_TEST_DIE_RANK": "1"})
    assert r.returncode == 3 and "rank 1 exited with status 3" in r.stderr, r.stderr[-2000:]
    assert time.time() - t0 < 60


def test_kernel_source_hash_guards_the_pmc_counters():
    """bench.py reports PMC-derived figures only when profiles/traffic.json was recorded for the current kernel sources"""
    sys.path.insert(0, ROOT)
    import bench
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    rec = json.load(open(tf))
    assert "kernel_source_hash" in rec  # written by tools/pmc_summary.py --traffic
