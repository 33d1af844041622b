"""`python bench.py --gpus N` without a launcher starts N fresh rank processes itself (before anything touches
the GPU), gives each RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, and returns their status.  LSM_BENCH_SPAWN_ONLY=1
runs only that launcher path: every rank joins a gloo group and the ranks are summed (no GPU needed)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update({"LSM_BENCH_SPAWN_ONLY": "1"}, **(extra_env or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2",
                           "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)


def test_bench_spawns_its_own_ranks():
    for n in (2, 3):
        p = _run(n)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1                                     # rank 0 only
        out = json.loads(lines[0])
        assert out["spawn_check"] and out["world"] == n and out["rank_sum"] == n * (n - 1) / 2


def test_bench_launcher_sets_its_own_rendezvous_and_rejects_bad_flags():
    p = _run(2, {"MASTER_ADDR": "256.0.0.1"})                      # the children get 127.0.0.1 whatever the caller had
    assert p.returncode == 0
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "nope"],
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0                                     # argparse error in the parent itself


def test_a_rank_that_dies_before_the_rendezvous_ends_the_run_promptly():
    """rank 1 exits 3 before init_process_group while ranks 0 and 2 wait in the rendezvous; the parent
    polls its children, terminates the survivors and returns that status -- within seconds, not after the
    process-group timeout (minutes)."""
    import time
    t0 = time.time()
    p = _run(3, {"LSM_BENCH_SPAWN_DIE": "1:3"})
    took = time.time() - t0
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    assert took < 90, took
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]      # nobody reported a result
