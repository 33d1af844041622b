"""Round 5: extended fuzz of the pair-block ring kernel (csrc/lif_pair.h).  Random small-world reservoirs whose size gives a pair layout
(2*ceil(N/256) blocks a multiple of 4, 8 or 16 waves; sizes just above a block boundary = padding neurons, at one = none), k from
narrow to as wide as the waves allow, 1..128 channels (natural and coloured mask positions), refractory periods 0..5, one leak
coefficient or one per neuron (LEAKV), 33..400 steps, quiet to saturated drive, a clip of non-0/1 bytes.  Pair blocks at every wave
count the reservoir offers (forced by name) and the quad kernel, against the C oracle: spike matrix, membrane trace, all eight feature
vectors, statistics.  usage: python3 exp/r05_fuzz_pairs.py <seed> <cases>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsm_speech_classifier_amd  # noqa: F401
from lsm_speech_classifier_amd import _lib, reservoir as R, snn, synth
from oracle import cport, ref_numpy as O
cport.build()
KEYS = ['spike_counts', 'spike_variances', 'mean_spike_times', 'first_spike_times', 'last_spike_times', 'mean_isi',
        'isi_variances', 'burst_counts']
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 51)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
runs, forms = 0, {}
for ci in range(n_cases):
    quads = int(rng.choice([4, 6, 8, 12, 16, 24, 32]))                   # 2*quads blocks
    n = int(rng.choice([quads * 256, quads * 256 - rng.randint(1, 200), (quads - 1) * 256 + rng.randint(1, 60)]))
    k = int(2 * rng.randint(max(2, n // 40), max(3, n // 7)))
    c = int(rng.choice([1, 7, 31, 32, 33, 40, 64, 100, 128]))
    t = int(rng.choice([33, 100, 400])) if n <= 4096 else int(rng.choice([33, 60]))
    dens = float(rng.choice([0.05, 0.25, 0.6]))
    refr = int(rng.choice([2, 2, 2, 0, 1, 3, 5]))
    div = None if rng.rand() < 0.5 else float(rng.choice([3.0, 10.0]))
    rasters = synth.bernoulli_raster(3, c, t, dens, seed=900 + ci)
    rasters[1] = (rasters[1] * 201).astype(np.uint8)                      # any non-zero byte is a spike
    wc = O.w_critico(k, 2.0, refr, rasters)
    p = R.SimulationParams(num_neurons=n, num_output_neurons=int(rng.randint(1, n + 1)), small_world_graph_k=k,
                           small_world_graph_p=float(rng.choice([0.05, 0.1, 0.3])), mean_weight=wc * float(rng.choice([0.6, 1.2])),
                           refractory_period=refr, leak_variance_divisor=div)
    res = R.build_reservoir(p, c)
    net = snn.SNN(None, reservoir=res)
    refs = [cport.lif_run(res, rasters[b], KEYS, want_trace=True) for b in range(3)]
    done = []
    for kernel in ("ring-pairs", "ring-quads"):
        try:
            net.set_kernel(kernel)
        except _lib.LsmHipError:
            done.append(f"{kernel}:none")
            continue
        for wpc in ((4, 8, 16) if kernel == "ring-pairs" else (0,)):
            stats = torch.zeros((3, 2), dtype=torch.int32, device="cuda")
            try:
                plan = net.plan(3, t, wpc)
                f, sm, vt = net.run_batch(rasters, KEYS, want_spike_matrix=True, want_v_trace=True, waves_per_clip=wpc,
                                          stats_out=stats)
            except _lib.LsmHipError as e:
                assert "layout" in str(e), str(e)
                continue
            f, sm, vt, st = f.cpu().numpy(), sm.cpu().numpy(), vt.cpu().numpy(), stats.cpu().numpy()
            for b in range(3):
                fr, smr, vtr = refs[b]
                assert np.array_equal(sm[b], smr) and np.array_equal(vt[b], vtr) and np.array_equal(f[b], fr), \
                    (n, k, c, t, refr, div, kernel, wpc, b, plan)
                assert st[b, 0] == int((smr.sum(0) > 0).sum()) and st[b, 1] == int(smr.sum()), (kernel, wpc, b, st[b])
            runs += 1
            if kernel == "ring-pairs":
                assert plan["input_mode"] in (14, 15), plan
                key = (plan["slots_per_lane"] // 2, plan["waves_per_clip"], plan["input_mode"], div is not None)
                forms[key] = forms.get(key, 0) + 1
            done.append(f"{kernel}/{plan['waves_per_clip']}x{plan['slots_per_lane']}:{plan['input_mode']}")
    print(f"case {ci}: N={n} k={k} C={c} T={t} dens={dens} refr={refr} div={div} spikes/clip={int(refs[0][1].sum())} ok "
          f"[{' '.join(done)}]", flush=True)
print(f"{runs} kernel/layout runs over {n_cases} reservoirs, all equal to the C oracle; pair-block forms "
      f"(blocks per wave, waves, input mode, per-neuron leak): {sorted(forms.items())}")
