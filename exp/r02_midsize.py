"""Where does the ring-row kernel overtake dense rows?  Reservoir kernel alone at mid sizes (k = 0.2 N, 128 channels)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lsm_speech_classifier_amd import reservoir as R, snn, synth
from oracle import ref_numpy as O
B = 512
for n in (1536, 2048, 3072):
    k = int(0.2 * n)
    rasters = synth.bernoulli_raster(B, 128, 400, 0.2, seed=n)
    wc = O.w_critico(k, 2.0, 2, rasters[:64])
    p = R.SimulationParams(num_neurons=n, num_output_neurons=int(0.4 * n), small_world_graph_k=k, mean_weight=wc * 0.6)
    net = snn.SNN(None, reservoir=R.build_reservoir(p, 128))
    dev = torch.from_numpy(rasters).cuda()
    ref = None
    for kernel in ("dense", "ring-contiguous", "ring-quads", "ring"):       # (round 5: "ring" = pair blocks where the reservoir has them)
        try:
            net.set_kernel(kernel)
        except Exception as e:
            print(n, kernel, "unavailable"); continue
        f, _, _ = net.run_batch(dev); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f, _, _ = net.run_batch(dev); e1.record(); torch.cuda.synchronize()
        same = True if ref is None else bool(torch.equal(f, ref))
        ref = f if ref is None else ref
        print(f"N={n} {kernel}: {e0.elapsed_time(e1):.3f} ms for {B} clips, layout {net.layout(B, 400)}, equal to dense: {same}, spikes/neuron {float(f[:, :int(0.4*n)].mean()):.2f}", flush=True)
