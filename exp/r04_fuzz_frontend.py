"""Round 4: extended fuzz of the front end.  Random filter counts (1..320 and 512 / 640 / 1024), clip lengths (1-4 overlapping
windows, hops that are no multiple of 8), bin counts, threshold tables (1-8 thresholds, positive / zero / negative hysteresis gaps),
row repeats and signals (chirps + noise, silence, a constant, a single click, tiny and large amplitudes, clipped noise): the
one-launch gammatone front end in both layouts and the split launches against the C oracle, bit for bit; then the mel branch's one
launch against its three split launches (equal bit for bit; the mel oracle is a tolerance test, tests/test_gpu_mel.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lsm_speech_classifier_amd  # noqa: F401
from lsm_speech_classifier_amd import frontend, synth
from oracle import cport, ref_numpy as O
cport.build()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n_mel = int(sys.argv[3]) if len(sys.argv) > 3 else 20


def signals(n, n_samples, rng):
    base = synth.class_chirps(rng.randint(0, 12, size=n), seed=int(rng.randint(1 << 30)))
    a = np.ascontiguousarray(np.resize(base, (n, n_samples)).astype(np.float32))
    a += (0.02 * rng.randn(n, n_samples)).astype(np.float32)
    kinds = []
    for b in range(n):
        kind = rng.choice(["mix", "mix", "mix", "silent", "constant", "click", "tiny", "large", "clipped", "noise"])
        if kind == "silent":
            a[b] = 0.0
        elif kind == "constant":
            a[b] = np.float32(rng.uniform(-1, 1))
        elif kind == "click":
            a[b] = 0.0; a[b, rng.randint(n_samples)] = 1.0
        elif kind == "tiny":
            a[b] *= np.float32(1e-6)
        elif kind == "large":
            a[b] *= np.float32(300.0)
        elif kind == "clipped":
            a[b] = np.clip(3 * rng.randn(n_samples), -1, 1).astype(np.float32)
        elif kind == "noise":
            a[b] = (0.1 * rng.randn(n_samples)).astype(np.float32)
        kinds.append(kind)
    return a, kinds


t0 = time.time()
runs = 0
for ci in range(n_cases):
    nf = int(rng.choice([rng.randint(1, 321), rng.randint(1, 321), 64, 128, 129, 256, 512, 640, 1024]))
    tb = int(rng.choice([100, 100, rng.randint(20, 129)]))
    # window = 25 ms = 400 samples; hop = n_samples / time_bins: 1..4 windows alive need hop >= 100
    ns = int(rng.choice([16000, rng.randint(max(100 * tb, 4000), min(400 * tb, 48000) + 1)]))
    n_thr = int(rng.randint(1, 9))
    thr = sorted(float(x) for x in rng.uniform(0.05, 0.99, size=n_thr))
    if len(set(thr)) != n_thr:
        continue
    gap = float(rng.choice([0.1, 0.0, 0.02, -0.05, 0.3]))
    red = int(rng.choice([1, 1, 2, 3]))
    n = int(rng.choice([1, 2, 3, 5])) if nf <= 320 else 2
    audio, kinds = signals(n, ns, rng)
    fe = frontend.SpikeFrontEnd(nf, "gammatone", redundancy=red, thresholds=thr, gap=gap, time_bins=tb, n_samples=ns)
    nw = (fe.nwin + fe.hop - 1) // fe.hop
    if not 1 <= nw <= 4 or fe.ncols < 2:
        continue
    coefs = O.gammatone_coefs(16000, nf, 50)
    ref = np.stack([cport.encode_hysteresis(cport.normalise_resize(cport.gammatone_db(
        cport.gammatone_spec(a, coefs, fe.nwin, fe.hop, fe.ncols)), tb), thr, gap) for a in audio])
    ref = np.repeat(ref, red, axis=1)
    routes = {"fused": fe.encode(audio, fused=True), "fused, one chain per lane": fe.encode(audio, fused=True, low_latency=True),
              "split": fe.encode(audio, fused=False)}
    for name, got in routes.items():
        g = got.cpu().numpy()
        assert g.shape == ref.shape, (name, g.shape, ref.shape)
        bad = np.nonzero((g != ref).reshape(n, -1).any(1))[0]
        assert bad.size == 0, (ci, name, nf, ns, tb, thr, gap, red, [kinds[b] for b in bad])
        runs += 1
    print(f"case {ci}: filters {nf} samples {ns} bins {tb} windows {nw} (nwin {fe.nwin} hop {fe.hop} cols {fe.ncols}) thresholds {n_thr} "
          f"gap {gap} repeat {red} clips {kinds} spikes {int(ref.sum())} ok", flush=True)
print(f"gammatone: all equal to the oracle: {runs} route runs in {time.time() - t0:.0f} s", flush=True)

mel_runs = 0
for ci in range(n_mel):
    nf = int(rng.choice([40, 40, rng.randint(8, 129)]))
    tb = int(rng.choice([100, rng.randint(20, 129)]))
    ns = int(rng.choice([16000, rng.randint(8000, 32001)]))
    n_thr = int(rng.randint(1, 9))
    thr = sorted(float(x) for x in rng.uniform(0.05, 0.99, size=n_thr))
    gap = float(rng.choice([0.1, 0.0, -0.05]))
    red = int(rng.choice([1, 3]))
    n = int(rng.choice([1, 3, 7]))
    audio, kinds = signals(n, ns, rng)
    try:
        fe = frontend.SpikeFrontEnd(nf, "mel", redundancy=red, thresholds=thr, gap=gap, time_bins=tb, n_samples=ns)
    except Exception as e:                       # shapes the mel branch refuses are refused by both routes
        print(f"mel case {ci}: filters {nf} samples {ns} bins {tb}: refused ({type(e).__name__}: {str(e)[:80]})", flush=True)
        continue
    split = fe.encode(audio, fused=False)
    try:
        one = fe.encode(audio, fused=True)
    except Exception as e:
        print(f"mel case {ci}: filters {nf} samples {ns} bins {tb}: one launch refused ({str(e)[:80]})", flush=True)
        continue
    assert torch.equal(one, split), (ci, nf, ns, tb, thr, gap, red, kinds)
    mel_runs += 1
    print(f"mel case {ci}: filters {nf} samples {ns} bins {tb} thresholds {n_thr} gap {gap} repeat {red} clips {kinds} "
          f"spikes {int(split.sum().item())} ok", flush=True)
print(f"mel: one launch equal to the split launches: {mel_runs} cases", flush=True)
