"""Round 4: when does each step of a run finish?  Completion time of every step (an event behind its last launch) relative to
the first launch, for a burst that starts right after a synchronisation -- cfg2, whole path / front ends alone / reservoir
alone.  Tells the clock ramp after the fence from pipeline fill: rate of steps 0-19, 20-39, ... of ONE run."""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lsm_speech_classifier_amd  # noqa: F401,E402
import bench  # noqa: E402
from lsm_speech_classifier_amd import frontend, reservoir, snn  # noqa: E402
from lsm_speech_classifier_amd.pipeline import HotPath  # noqa: E402

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
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for stage in ("full", "frontend", "reservoir"):
    hp = HotPath(fe, net, bench.FEATURE_SET)
    src = r0 if stage == "reservoir" else audio
    hp.prime(src, stage=stage, min_ms=40.0)
    for rep in range(2):
        hp.fork_from_current()
        for _ in range(5):
            hp.submit(src, stage=stage)
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True)
        t0.record()
        evs = []
        for i in range(steps):
            _, st = hp.submit(src, stage=stage)
            e = torch.cuda.Event(enable_timing=True)
            e.record(st)
            evs.append(e)
        torch.cuda.synchronize()
        done = np.sort(np.array([t0.elapsed_time(e) for e in evs]))
        # rate over windows of 20 completions
        marks = [0.0] + [float(done[k - 1]) for k in range(20, steps + 1, 20)]
        rates = [round((marks[k + 1] - marks[k]) / 20, 4) for k in range(len(marks) - 1)]
        print(f"{stage} rep {rep}: {steps} steps in {done[-1]:.3f} ms = {done[-1] / steps:.4f} ms/step; first completion at "
              f"{done[0]:.3f} ms; ms/step per window of 20 completions: {rates}", flush=True)
