"""Debug: first difference between the dense-row and band-row kernels on a small reservoir."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lsm_speech_classifier_amd import reservoir as R, snn, synth
from oracle import ref_numpy as O
N, k, C = 200, 40, 32
rasters = synth.bernoulli_raster(2, C, 60, 0.2, seed=200)
wc = O.w_critico(k, 2.0, 2, rasters)
res = R.build_reservoir(R.SimulationParams(num_neurons=N, num_output_neurons=80, small_world_graph_k=k, mean_weight=wc * 0.6), C)
net = snn.SNN(None, reservoir=res)
ptr, post = res.csc_ptr, res.csc_post
H = (len(post) // N + 1) // 2
for wpc in (1, 2, 4):
    net.set_kernel("dense"); f0, s0, v0 = net.run_batch(rasters, None, want_spike_matrix=True, want_v_trace=True, waves_per_clip=wpc)
    net.set_kernel("band"); f1, s1, v1 = net.run_batch(rasters, None, want_spike_matrix=True, want_v_trace=True, waves_per_clip=wpc)
    v0, v1, s0 = v0.cpu().numpy(), v1.cpu().numpy(), s0.cpu().numpy()
    d = np.argwhere(v0 != v1)
    print("wpc", wpc, "differing trace entries", len(d))
    if len(d):
        b, t, i = d[0]
        spk = np.nonzero(s0[b, t - 1])[0] if t > 0 else []
        print(" first diff clip", b, "step", t, "neuron", i, "dense", v0[b, t, i], "band", v1[b, t, i], "diff", v1[b,t,i]-v0[b,t,i])
        print(" spikers at t-1:", list(spk)[:20])
        for j in spk:
            tg = post[ptr[j]:ptr[j + 1]]
            if i in tg:
                q = (i - (j - H)) % N
                print("   synapse", j, "->", i, "weight", res.csc_w[ptr[j]:ptr[j+1]][list(tg).index(i)], "band index", q, "in band", q < 2 * H + 1)
        same_t = d[(d[:, 0] == b) & (d[:, 1] == t)][:, 2]
        print(" neurons differing at that step:", list(same_t)[:30])
