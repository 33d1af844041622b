"""Round 3: extended fuzz of the one-launch front end (lsm_gammatone_spikes_f64) against the C oracle and the two split
launches -- random filter counts (1..300, incl. ragged last groups and > 256), clip lengths (window overlaps 1..4, hops
that are not multiples of 8), time bins (resize up, down, none), 1..8 thresholds, gaps, redundancy, batch sizes around
the workgroup size, both launch layouts; clips from silence to clipping noise.  Not part of the test suite (minutes of
oracle time); prints one line per case and a summary.
    python exp/r03_fuzz_fused.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lsm_speech_classifier_amd import frontend, synth           # noqa: E402
from oracle import cport, ref_numpy as O                        # noqa: E402

cport.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
bad = flips = cells = 0
t0 = time.time()
for case in range(n_cases):
    F = int(rng.choice([1, 2, 7, 31, 40, 63, 64, 65, 96, 127, 128, 129, 192, 200, 255, 256, 257, 300]))
    n_samples = int(rng.choice([4000, 6000, 8000, 12000, 13000, 16000, 20000, 24000]))
    time_bins = int(rng.choice([20, 37, 50, 77, 98, 100, 120]))
    hop_time = n_samples / (16000 * time_bins)
    nwin, hop, ncols = frontend.gtgram_strides(16000, frontend.GT_WINDOW_TIME, hop_time, n_samples)
    if not (1 <= (nwin + hop - 1) // hop <= 4) or ncols < 2 or hop < 1:
        continue
    n_thr = int(rng.randint(1, 9))
    thr = sorted(float(x) for x in rng.choice(np.arange(0.04, 0.99, 0.03), size=n_thr, replace=False))
    gap = float(rng.choice([0.01, 0.03, 0.05, 0.1, 0.2]))
    red = int(rng.randint(1, 4))
    B = int(rng.choice([1, 2, 3, 4, 5, 7, 9]))
    kind = rng.randint(0, 4)
    if kind == 0:
        audio = np.resize(synth.class_chirps(rng.randint(0, 12, size=B), seed=int(rng.randint(1 << 20))), (B, n_samples))
    elif kind == 1:
        audio = (rng.randn(B, n_samples) * float(rng.choice([1e-4, 0.05, 1.0, 30.0]))).astype(np.float32)
    elif kind == 2:
        t = np.arange(n_samples) / 16000.0
        audio = np.stack([np.sin(2 * np.pi * rng.uniform(60, 7000) * t) * rng.uniform(0.01, 1) for _ in range(B)])
    else:
        audio = np.zeros((B, n_samples))
        audio[:, rng.randint(0, n_samples)] = 1.0                       # an impulse
    audio = np.ascontiguousarray(audio, dtype=np.float32)
    if B > 1 and rng.rand() < 0.3:
        audio[0] = 0.0                                                  # a silent clip in the batch
    fe = frontend.SpikeFrontEnd(F, "gammatone", redundancy=red, thresholds=thr, gap=gap, time_bins=time_bins,
                                n_samples=n_samples)
    got = fe.encode(audio, fused=True)
    same = torch.equal(got, fe.encode(audio, fused=False)) and torch.equal(got, fe.encode(audio, fused=True, low_latency=True))
    coefs = O.gammatone_coefs(16000, F, 50)
    ref = np.stack([np.repeat(cport.encode_hysteresis(cport.normalise_resize(cport.gammatone_db(
        cport.gammatone_spec(a, coefs, nwin, hop, ncols)), time_bins), thr, gap), red, axis=0) for a in audio])
    diff = int((got.cpu().numpy() != ref).sum())
    flips += diff
    cells += ref.size
    ok = same and diff == 0
    bad += not ok
    print(f"case {case:3d} F={F:3d} L={n_samples:5d} bins={time_bins:3d} windows={(nwin + hop - 1) // hop} cols={ncols:3d} thr={n_thr} "
          f"gap={gap} R={red} B={B} kind={kind}: {'ok' if ok else 'MISMATCH'} (fused == split == low-latency: {same}; "
          f"cells differing from the oracle: {diff} of {ref.size}, spikes {int(ref.sum())})", flush=True)
print(f"{n_cases} cases, {bad} not ok, {flips} of {cells} raster cells differ from the oracle, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
