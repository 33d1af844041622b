"""Round 5: reservoirs with a leak coefficient per neuron (the reference's --leak-variance-divisor, extract_lsm_features.py:174) at cfg4's
size: the pair-block kernel (masks + coefficients in registers, LEAKV) against the quad kernel (packed-entry input drive), 1024 clips."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lsm_speech_classifier_amd import reservoir as R, snn, synth
from oracle import ref_numpy as O
B, n, k = 1024, 4000, 800
rasters = synth.bernoulli_raster(B, 128, 400, 0.2, seed=n)
wc = O.w_critico(k, 2.0, 2, rasters[:64])
for div in (None, 5.0):
    p = R.SimulationParams(num_neurons=n, num_output_neurons=int(0.4 * n), small_world_graph_k=k, mean_weight=wc * 0.6,
                           leak_variance_divisor=div)
    net = snn.SNN(None, reservoir=R.build_reservoir(p, 128))
    dev = torch.from_numpy(rasters).cuda()
    ref = None
    for kernel in ("ring-quads", "ring"):
        net.set_kernel(kernel)
        f, _, _ = net.run_batch(dev); torch.cuda.synchronize()
        ms = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f, _, _ = net.run_batch(dev); e1.record(); torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        same = True if ref is None else bool(torch.equal(f, ref))
        ref = f if ref is None else ref
        print(f"N={n} leak divisor {div} {kernel}: {np.median(ms):.3f} ms for {B} clips, input mode {net.plan(B, 400, 0)['input_mode']}, "
              f"equal to the quad kernel: {same}, spikes/neuron {float(f[:, :int(0.4*n)].mean()):.2f}", flush=True)
