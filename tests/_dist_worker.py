"""Worker for tests/test_dist_gloo.py (one process per rank, gloo backend, CPU tensors)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class FakeLsm:
    """Stands in for snn.SNN on a CPU box: 'features' are a deterministic function of the clip, so
    the sharding/gather logic of extract_all_features can be checked without a GPU."""
    device = torch.device("cpu")
    num_output_neurons = 3

    def run_batch(self, spikes, feature_keys):
        x = torch.from_numpy(np.ascontiguousarray(spikes)).float()
        base = torch.stack([x.sum(dim=(1, 2)), x[:, 0].sum(dim=1), x[:, :, 0].sum(dim=1)], dim=1)
        return torch.cat([base * (k + 1) for k in range(len(feature_keys))], dim=1), None, None


def real_lsm():
    """The real reservoir on cuda:0 (tests/test_gpu_hotpath.py: two ranks share the card, gloo moves the rows)."""
    from lsm_speech_classifier_amd.snn import SNN, SimulationParams
    return SNN(SimulationParams(num_neurons=300, num_output_neurons=100, small_world_graph_k=30, mean_weight=0.05),
               n_channels=4, device="cuda:0")


def main():
    out_dir, n = sys.argv[1], int(sys.argv[2])
    gpu = len(sys.argv) > 3 and sys.argv[3] == "gpu"
    from lsm_speech_classifier_amd import dist as lsm_dist
    import extract_lsm_features as ex
    rank, _, world = lsm_dist.init("gloo")
    rs = np.random.RandomState(1)
    clips = (rs.rand(n, 4, 6 if not gpu else 120) < 0.4).astype(np.uint8)
    if gpu:
        feats = ex.extract_all_features(real_lsm(), clips, ["spike_counts", "mean_isi"], "")
    else:
        feats = ex.extract_all_features(FakeLsm(), clips, ["a", "b"], "")
    lo, hi = lsm_dist.shard_range(n, rank, world)
    w = lsm_dist.broadcast_float(3.25 + rank, 0)
    # stage 1's exchange: blocks whose lengths the ranks do not know of each other (rank 1 contributes nothing)
    n_mine = 0 if rank == 1 else 2 * rank + 3
    block = (torch.arange(n_mine * 5, dtype=torch.int32).reshape(n_mine, 5) + 1000 * rank).to(torch.uint8 if gpu else torch.int32)
    if gpu:
        block = block.cuda()
    var = lsm_dist.gather_varrows(block).cpu().numpy()
    # the bench's exchange of a run of steps in chunks (tail chunk included) against one gather of everything
    steps, B, F = 7, 3, 4
    mine = (torch.arange(steps * B * F, dtype=torch.float32).reshape(steps, B, F) + 1000.0 * rank)
    if gpu:
        mine = mine.cuda()
    digests = []
    for chunk in (3, steps, 1):
        gathered = torch.full((world * steps * B, F), -1.0, dtype=torch.float32, device=mine.device)
        for c0, c1 in lsm_dist.chunk_bounds(steps, chunk):
            lsm_dist.gather_step_chunk(gathered, mine, c0, c1)
        blocks = lsm_dist.rows_by_rank(gathered, world, steps, chunk, B)
        for r in range(world):
            want = torch.arange(steps * B * F, dtype=torch.float32).reshape(steps * B, F) + 1000.0 * r
            assert torch.equal(blocks[r].cpu(), want), (chunk, r)
        digests.append(lsm_dist.rows_digest(blocks))
    assert digests[0] == digests[1] == digests[2] != 0
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), feats=feats, lo=lo, hi=hi, w=w, var=var, digest=digests[0])
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
