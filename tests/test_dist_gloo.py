"""The N>1 path on CPU: world_size 2 (and 3, uneven shards) over gloo.  Clips shard into contiguous
blocks, every rank ends up with all feature rows in dataset order, equal to the 1-process result."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n", [(2, 10), (2, 7), (3, 8)])
def test_sharded_feature_gather_equals_single_process(tmp_path, world, n):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _dist_worker
    import extract_lsm_features as ex
    rs = np.random.RandomState(1)
    clips = (rs.rand(n, 4, 6) < 0.4).astype(np.uint8)
    single = ex.extract_all_features(_dist_worker.FakeLsm(), clips, ["a", "b"], "")
    assert single.shape == (n, 6)

    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"),
                                       str(tmp_path), str(n)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    covered = []
    for rank in range(world):
        d = np.load(tmp_path / f"rank{rank}.npz")
        np.testing.assert_array_equal(d["feats"], single)          # all rows, dataset order, every rank
        covered += list(range(int(d["lo"]), int(d["hi"])))
        assert float(d["w"]) == 3.25                                # rank 0's value everywhere
        want = np.concatenate([np.arange((0 if r == 1 else 2 * r + 3) * 5, dtype=np.int32).reshape(-1, 5) + 1000 * r
                               for r in range(world)])
        np.testing.assert_array_equal(d["var"], want)               # ragged blocks, rank order, every rank
    assert covered == list(range(n))
    # bench.py's chunked exchange (dist.gather_step_chunk / rows_by_rank): asserted inside every rank; all ranks agree
    assert len({int(np.load(tmp_path / f"rank{rank}.npz")["digest"]) for rank in range(world)}) == 1
