"""One-off extended fuzz of the large-N reservoir kernels (beyond tests/test_gpu_fuzz.py, whose cases stop at
N = 2700): random N up to 8192, k, rewiring and input shapes; dense / ring / ring-contiguous at the chosen and forced
layouts against the C oracle (spike matrix, membrane trace, features, statistics), oracle run once per clip."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsm_speech_classifier_amd  # noqa: F401
from lsm_speech_classifier_amd import _lib, reservoir as R, snn, synth
from oracle import cport, ref_numpy as O
cport.build()
KEYS = ['spike_counts', 'spike_variances', 'mean_spike_times', 'first_spike_times', 'last_spike_times', 'mean_isi',
        'isi_variances', 'burst_counts']
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 10
checked = 0
for ci in range(n_cases):
    n = int(rng.choice([rng.randint(2700, 8193), 4096, 8192, 3072, 5000, 6144, 7777]))
    k = int(2 * rng.randint(max(8, n // 40), max(9, n // 8)))
    c = int(rng.choice([16, 40, 128, 200, 256]))
    t = int(rng.choice([37, 100, 400]))
    dens = float(rng.choice([0.05, 0.2, 0.5]))
    p_rew = float(rng.choice([0.02, 0.1, 0.25]))
    refr = int(rng.randint(0, 4))
    div = None if rng.rand() < 0.5 else 5.0
    rasters = synth.bernoulli_raster(2, c, t, dens, seed=100 + ci)
    wc = O.w_critico(k, 2.0, refr, rasters)
    p = R.SimulationParams(num_neurons=n, num_output_neurons=int(rng.randint(1, n + 1)), small_world_graph_k=k,
                           small_world_graph_p=p_rew, mean_weight=wc * float(rng.choice([0.6, 1.5])),
                           refractory_period=refr, leak_variance_divisor=div)
    t0 = time.time()
    res = R.build_reservoir(p, c)
    net = snn.SNN(None, reservoir=res)
    refs = [cport.lif_run(res, rasters[b], KEYS, want_trace=True) for b in range(2)]
    kernels = ["dense"]
    try:
        net.set_kernel("ring"); kernels += ["ring", "ring-contiguous"]
    except _lib.LsmHipError:
        pass
    done = []
    for kernel in kernels:
        try:
            net.set_kernel(kernel)
        except _lib.LsmHipError:
            continue
        for wpc in (0, 4, 8, 16):
            stats = torch.zeros((2, 2), dtype=torch.int32, device="cuda")
            try:
                f, sm, vt = net.run_batch(rasters, KEYS, want_spike_matrix=True, want_v_trace=True, waves_per_clip=wpc,
                                          stats_out=stats)
            except _lib.LsmHipError as e:          # a forced layout the reservoir does not have, or (many output neurons
                assert "layout" in str(e), str(e)  # at large N) a per-clip LDS image beyond 160 KB: refused loudly
                if wpc == 0:
                    done.append(f"{kernel}: no layout fits")
                continue
            f, sm, vt, st = f.cpu().numpy(), sm.cpu().numpy(), vt.cpu().numpy(), stats.cpu().numpy()
            for b in range(2):
                fr, smr, vtr = refs[b]
                assert np.array_equal(sm[b], smr) and np.array_equal(vt[b], vtr) and np.array_equal(f[b], fr), \
                    (n, k, c, t, kernel, wpc, b)
                assert st[b, 0] == int((smr.sum(0) > 0).sum()) and st[b, 1] == int(smr.sum()), (kernel, wpc, b, st[b])
            done.append(f"{kernel}/{wpc}:{net.layout(2, t, wpc)['waves_per_clip']}")
            checked += 1
    print(f"case {ci}: N={n} k={k} p={p_rew} C={c} T={t} dens={dens} refr={refr} spikes/clip={int(refs[0][1].sum())} "
          f"ok [{' '.join(done)}] ({time.time() - t0:.0f} s)", flush=True)
print("all equal to the oracle:", checked, "kernel/layout runs")
