"""Round 4: extended fuzz of the reservoir kernels at the sizes the dense rows serve by default (N 64..2700), aimed at this round's
forms: refractory countdown in scalar masks (period 2) or in registers (others), coloured input masks (C <= 128, a colouring
exists) or natural ones, entry lists (C > 128), ring rows with masks (uniform leak) or entries.  Every kernel the reservoir
offers, chosen and forced layouts, against the C oracle (spike matrix, membrane trace, 8 feature vectors, statistics)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsm_speech_classifier_amd  # noqa: F401
from lsm_speech_classifier_amd import _lib, reservoir as R, snn, synth
from oracle import cport, ref_numpy as O
cport.build()
KEYS = ['spike_counts', 'spike_variances', 'mean_spike_times', 'first_spike_times', 'last_spike_times', 'mean_isi',
        'isi_variances', 'burst_counts']
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
runs, modes = 0, {}
for ci in range(n_cases):
    n = int(rng.choice([rng.randint(64, 2701), 1000, 1024, 500, 2048, 777]))
    k = int(2 * rng.randint(max(2, n // 40), max(3, n // 6)))
    c = int(rng.choice([1, 7, 31, 32, 33, 40, 64, 100, 128, 129, 200]))
    t = int(rng.choice([33, 100, 400]))
    dens = float(rng.choice([0.05, 0.25, 0.6]))
    refr = int(rng.choice([2, 2, 2, 0, 1, 3, 5]))
    div = None if rng.rand() < 0.6 else float(rng.choice([3.0, 10.0]))
    rasters = synth.bernoulli_raster(3, c, t, dens, seed=500 + ci)
    wc = O.w_critico(k, 2.0, refr, rasters)
    p = R.SimulationParams(num_neurons=n, num_output_neurons=int(rng.randint(1, n + 1)), small_world_graph_k=k,
                           small_world_graph_p=float(rng.choice([0.05, 0.1, 0.3])), mean_weight=wc * float(rng.choice([0.6, 1.2])),
                           refractory_period=refr, leak_variance_divisor=div)
    res = R.build_reservoir(p, c)
    net = snn.SNN(None, reservoir=res)
    refs = [cport.lif_run(res, rasters[b], KEYS, want_trace=True) for b in range(3)]
    kernels = ["dense", "sparse"]
    try:
        net.set_kernel("ring"); kernels += ["ring", "ring-contiguous"]
    except _lib.LsmHipError:
        pass
    done = []
    for kernel in kernels:
        net.set_kernel(kernel)
        for wpc in (0, 1, 2, 4, 8, 16):
            stats = torch.zeros((3, 2), dtype=torch.int32, device="cuda")
            try:
                mode = net.plan(3, t, wpc)["input_mode"]
                f, sm, vt = net.run_batch(rasters, KEYS, want_spike_matrix=True, want_v_trace=True, waves_per_clip=wpc,
                                          stats_out=stats)
            except _lib.LsmHipError as e:
                assert "layout" in str(e), str(e)
                continue
            f, sm, vt, st = f.cpu().numpy(), sm.cpu().numpy(), vt.cpu().numpy(), stats.cpu().numpy()
            for b in range(3):
                fr, smr, vtr = refs[b]
                assert np.array_equal(sm[b], smr) and np.array_equal(vt[b], vtr) and np.array_equal(f[b], fr), \
                    (n, k, c, t, refr, div, kernel, wpc, b, mode)
                assert st[b, 0] == int((smr.sum(0) > 0).sum()) and st[b, 1] == int(smr.sum()), (kernel, wpc, b, st[b])
            runs += 1
            modes[mode] = modes.get(mode, 0) + 1
            done.append(f"{kernel}/{wpc}:{mode}")
    print(f"case {ci}: N={n} k={k} C={c} T={t} dens={dens} refr={refr} div={div} spikes/clip={int(refs[0][1].sum())} ok "
          f"[{' '.join(done)}]", flush=True)
print(f"all equal to the oracle: {runs} kernel/layout runs; input modes used (0/1 dense entries, 2/3 dense masks natural/coloured, "
      f"10/11 ring entries, 12/13 ring masks, 20 sparse): {dict(sorted(modes.items()))}")
