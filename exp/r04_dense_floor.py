"""Round 4: what bounds a LONE cfg2 launch of the dense-row kernel.  Every clip of the batch is launched
alone (one workgroup on an idle GPU = the clip's own 400-step chain, nothing to wait for but itself), then the whole batch.
If the batch takes what its slowest clip takes, the launch is bound by ONE clip's dependency chain and no scheduling, occupancy
or bandwidth change can shorten it -- only a shorter chain (fewer cycles per step) or fewer steps can."""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lsm_speech_classifier_amd  # noqa: F401,E402
import bench  # noqa: E402
from lsm_speech_classifier_amd import frontend, reservoir, snn  # noqa: E402

cfg = bench.CONFIGS["cfg2"]
B = cfg["batch"]
dev = torch.device("cuda", 0)
fe = frontend.SpikeFrontEnd(cfg["n_filters"], cfg["filterbank"], device=dev)
audio = torch.from_numpy(bench.make_audio(cfg["audio"], B, seed=1234)).to(dev)
r0 = fe.encode(audio)
wc = bench.w_critico(cfg["k"], 2.0, 2, r0)
params = reservoir.SimulationParams(num_neurons=cfg["N"], num_output_neurons=cfg["n_out"], small_world_graph_k=cfg["k"],
                                    mean_weight=wc * bench.MULTIPLIER)
net = snn.SNN(params, reservoir=reservoir.build_reservoir(params, fe.n_channels), device=dev)


def timed(x, wpc, reps=5):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        net.run_batch(x, bench.FEATURE_SET, waves_per_clip=wpc)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


st = torch.empty((B, 2), dtype=torch.int32, device=dev)
net.run_batch(r0, bench.FEATURE_SET, stats_out=st)
spikes = st[:, 1].cpu().numpy()
in_spikes = r0.sum(dim=(1, 2)).cpu().numpy()
for wpc in (8, 4, 16):
    batch_ms = timed(r0, wpc)
    empty_ms = timed(torch.zeros_like(r0[:1]), wpc)          # a silent clip: the bare 400-step loop (no rows, no spikes)
    per = np.array([timed(r0[b:b + 1], wpc, reps=3) for b in range(B)])
    order = np.argsort(per)
    rows_per_step = spikes / 400.0
    # cost of one row (one presynaptic spike) on the chain: slope of the single-clip time against its reservoir spikes
    slope, icpt = np.polyfit(spikes, per, 1)
    print(f"waves per clip {wpc}: batch of {B} clips {batch_ms:.4f} ms; single clips alone: min {per.min():.4f} median "
          f"{np.median(per):.4f} max {per.max():.4f} ms (clip {int(order[-1])}: {int(spikes[order[-1]])} reservoir spikes, "
          f"{rows_per_step[order[-1]]:.2f} rows/step; quietest clip {int(order[0])}: {int(spikes[order[0]])} spikes); "
          f"silent clip {empty_ms:.4f} ms = {empty_ms * 1e3 / 400:.3f} us per step; fit: {icpt:.4f} ms + {slope * 1e6:.1f} ns per "
          f"reservoir spike; batch / slowest clip = {batch_ms / per.max():.3f}", flush=True)
print(f"reservoir spikes per clip: min {spikes.min()} mean {spikes.mean():.0f} max {spikes.max()}; input spikes per clip: min "
      f"{in_spikes.min()} mean {in_spikes.mean():.0f} max {in_spikes.max()}")
