"""Synthetic inputs: no Speech Commands corpus exists on either box.

* ``white_noise``: the roofline-stress input of BASELINE.json configs[4].
* ``class_chirps``: class-structured, speech-like 1 s clips (three amplitude-modulated chirps per
  class + noise) so that accuracy-equality checks have something learnable.
* ``bernoulli_raster``: reservoir-only timing input, uint8 (B, C, T) at a given density.
"""
from __future__ import annotations

import numpy as np

SAMPLE_RATE = 16000
CLIP_SAMPLES = 16000


def white_noise(n_clips: int, seed: int = 1234, n_samples: int = CLIP_SAMPLES) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((n_clips, n_samples)) * 0.1).astype(np.float32)


def class_chirps(labels, seed: int = 1234, n_samples: int = CLIP_SAMPLES) -> np.ndarray:
    """One speech-like clip per entry of ``labels``: a harmonic source (pitch glide) shaped by
    three formant resonances whose start/end frequencies, onset and duration are fixed by the
    class; the clip index jitters them a little and adds 0.04-sigma background noise."""
    labels = np.asarray(labels, dtype=np.int64)
    t = np.arange(n_samples) / SAMPLE_RATE
    out = np.empty((len(labels), n_samples), dtype=np.float32)
    harm = np.arange(1, 41)[:, None]
    for n, c in enumerate(labels):
        crng = np.random.default_rng(seed + int(c))                 # class template
        f_start = crng.uniform(300.0, 3800.0, size=3)
        f_end = crng.uniform(300.0, 3800.0, size=3)
        onset = crng.uniform(0.05, 0.30)
        dur = crng.uniform(0.45, 0.65)
        pitch0, pitch1 = crng.uniform(100.0, 220.0, size=2)
        srng = np.random.default_rng((seed + 7919) * 1000003 + n)   # per-clip jitter
        on = onset + 0.02 * srng.standard_normal()
        u = np.clip((t - on) / dur, 0.0, 1.0)
        gate = (t >= on) & (t <= on + dur)
        env = (np.sin(np.pi * u) ** 2) * (0.6 + 0.4 * np.sin(2 * np.pi * 4.0 * t + c)) * gate
        pitch = (pitch0 + (pitch1 - pitch0) * u) * (1 + 0.02 * srng.standard_normal())
        phase = 2 * np.pi * np.cumsum(pitch) / SAMPLE_RATE
        hf = harm * pitch[None, :]
        gain = np.zeros_like(hf)
        for j in range(3):
            fc = (f_start[j] + (f_end[j] - f_start[j]) * u) * (1 + 0.03 * srng.standard_normal())
            gain += np.exp(-0.5 * ((hf - fc[None, :]) / 500.0) ** 2)
        gain *= hf < 0.45 * SAMPLE_RATE
        x = (gain * np.sin(harm * phase[None, :])).sum(axis=0) * env
        x *= 0.5 / max(1e-9, np.abs(x).max())
        x += 0.04 * srng.standard_normal(n_samples)
        out[n] = x.astype(np.float32)
    return out


def bernoulli_raster(n_clips: int, n_channels: int, n_steps: int = 400, density: float = 0.2,
                     seed: int = 1234) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return (rng.random((n_clips, n_channels, n_steps)) < density).astype(np.uint8)
