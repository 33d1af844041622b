"""Generate tests/golden/*.npz by running the REFERENCE's own in-tree NumPy functions.

Run in the build container only (needs /root/reference, which never travels):
    python tests/make_golden.py

The reference modules import three third-party packages that are absent here (librosa,
gammatone, snnpy).  They are replaced by stub modules so that the module bodies import; only
functions whose arithmetic is entirely in-tree are then called for real:

* create_dataset.convert_spectrogram_to_spikes_hysteresis   (create_dataset.py:81-98)
* create_dataset.create_pure_redundancy                      (create_dataset.py:101-104)
* create_dataset.audio_to_spectrogram lines 59-78 (dB/floor/min-max/zoom/crop) — reached by
  making the stubbed gtgram / melspectrogram / power_to_db return arrays supplied by this script,
  so everything after the third-party call is the reference's code
* extract_lsm_features.calculate_theoretical_w_critico       (extract_lsm_features.py:33-60)

The fixtures hold inputs and the reference's outputs only (data, no source text).
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_reference():
    holder = {}
    for name in ["librosa", "librosa.feature", "gammatone", "gammatone.gtgram", "snnpy", "snnpy.snn"]:
        sys.modules[name] = types.ModuleType(name)
    sys.modules["librosa"].feature = sys.modules["librosa.feature"]
    sys.modules["gammatone"].gtgram = sys.modules["gammatone.gtgram"]
    sys.modules["snnpy.snn"].SNN = object
    sys.modules["snnpy.snn"].SimulationParams = object
    sys.modules["gammatone.gtgram"].gtgram = lambda **kw: holder["spec"]
    sys.modules["librosa.feature"].melspectrogram = lambda **kw: holder["spec"]
    sys.modules["librosa"].power_to_db = lambda S, ref=None: holder["db"]
    sys.path.insert(0, REF)
    import create_dataset as cd
    import extract_lsm_features as ex
    return cd, ex, holder


def spectrogram_cases(rs):
    cases = {}
    t = np.linspace(0, 1, 100)
    for F in (1, 2, 40, 128, 256):
        cases[f"uniform_F{F}"] = rs.rand(F, 100)
    smooth = 0.5 + 0.5 * np.sin(2 * np.pi * (3 * t[None, :] + np.linspace(0, 1, 40)[:, None]))
    cases["smooth_F40"] = smooth
    cases["walk_F128"] = np.clip(0.75 + np.cumsum(rs.randn(128, 100) * 0.05, axis=1), 0, 1)
    # values sitting exactly on ON thresholds and on the Python-float OFF bounds
    thr = [0.70, 0.80, 0.90, 0.95]
    specials = np.array(thr + [x - 0.1 for x in thr] + [0.0, 1.0, 0.85, 0.6, 0.7000000000000001,
                        np.nextafter(0.95, 1), np.nextafter(0.95, 0), np.nextafter(0.85, 0)])
    cases["edges_F16"] = specials[rs.randint(0, len(specials), size=(16, 100))]
    cases["zeros_F4"] = np.zeros((4, 100))
    cases["ones_F4"] = np.ones((4, 100))
    return cases


def main():
    os.makedirs(OUT, exist_ok=True)
    cd, ex, holder = load_reference()
    rs = np.random.RandomState(20240611)

    # ---- (1) hysteresis encoder, float64 and float32 inputs ------------------------------
    enc = {}
    for name, spec in spectrogram_cases(rs).items():
        for dt in (np.float64, np.float32):
            x = spec.astype(dt)
            y = cd.convert_spectrogram_to_spikes_hysteresis(x, cd.SPIKE_THRESHOLDS, cd.HYSTERESIS_GAP)
            enc[f"{name}_{np.dtype(dt).name}_in"] = x
            enc[f"{name}_{np.dtype(dt).name}_out"] = y
    # non-default gap / threshold list (the function's own default gap is 0.05)
    x = rs.rand(8, 100)
    enc["gap005_float64_in"] = x
    enc["gap005_float64_out"] = cd.convert_spectrogram_to_spikes_hysteresis(x, [0.5, 0.9, 0.3])
    enc["thresholds"] = np.array(cd.SPIKE_THRESHOLDS)
    enc["gap"] = np.array(cd.HYSTERESIS_GAP)
    np.savez_compressed(os.path.join(OUT, "encoder.npz"), **enc)

    # ---- (2) redundancy ------------------------------------------------------------------
    r = (rs.rand(6, 40) < 0.3).astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "redundancy.npz"), x=r,
                        r1=cd.create_pure_redundancy(r, 1), r3=cd.create_pure_redundancy(r, 3))

    # ---- (3) post-filterbank part of audio_to_spectrogram --------------------------------
    post = {}
    audio = np.zeros(16000, dtype=np.float32)
    # gammatone branch: gtgram output is float64 (F, 98), positive magnitudes
    for name, F, scale in (("gt_a", 128, 1e-3), ("gt_b", 40, 1.0), ("gt_c", 16, 1e-6)):
        spec = np.abs(rs.randn(F, 98)) * scale * np.exp(rs.randn(F, 1) * 2)
        holder["spec"] = spec
        post[f"{name}_in"] = spec
        post[f"{name}_out"] = cd.audio_to_spectrogram(audio, F, "gammatone")
    holder["spec"] = np.full((8, 98), 0.25)                      # flat -> zeros branch
    post["gt_flat_in"] = holder["spec"]
    post["gt_flat_out"] = cd.audio_to_spectrogram(audio, 8, "gammatone")
    # mel branch: power_to_db output is float32 (F, 101), max 0, floor -80
    for name, F in (("mel_a", 128), ("mel_b", 40)):
        db = np.maximum(-np.abs(rs.randn(F, 101)) * 25.0, -80.0).astype(np.float32)
        db[rs.randint(F), rs.randint(101)] = 0.0
        holder["spec"] = None
        holder["db"] = db
        post[f"{name}_in"] = db
        post[f"{name}_out"] = cd.audio_to_spectrogram(audio, F, "mel")
    np.savez_compressed(os.path.join(OUT, "postfilter.npz"), **post)

    # ---- (4) w_critico -------------------------------------------------------------------
    wc = {}
    P = types.SimpleNamespace
    cases = [
        ("dense", (rs.rand(20, 128, 400) < 0.238).astype(np.uint8), P(small_world_graph_k=200, membrane_threshold=2.0, refractory_period=2)),
        ("sparse", (rs.rand(7, 40, 400) < 0.03).astype(np.uint8), P(small_world_graph_k=100, membrane_threshold=2.0, refractory_period=2)),
        ("many", (rs.rand(600, 4, 8) < 0.5).astype(np.uint8), P(small_world_graph_k=1600, membrane_threshold=1.5, refractory_period=3)),
        ("kzero", (rs.rand(3, 4, 8) < 0.5).astype(np.uint8), P(small_world_graph_k=0, membrane_threshold=2.0, refractory_period=2)),
        ("empty", np.zeros((0, 4, 8), dtype=np.uint8), P(small_world_graph_k=200, membrane_threshold=2.0, refractory_period=2)),
    ]
    for name, data, params in cases:
        with contextlib.redirect_stdout(io.StringIO()):
            w = ex.calculate_theoretical_w_critico(params, data)
        wc[f"{name}_in"] = data
        wc[f"{name}_params"] = np.array([params.small_world_graph_k, params.membrane_threshold,
                                         params.refractory_period], dtype=np.float64)
        wc[f"{name}_out"] = np.array(w, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "w_critico.npz"), **wc)

    # ---- (5) constants the boundary must reproduce ---------------------------------------
    np.savez_compressed(
        os.path.join(OUT, "constants.npz"),
        feature_set_names=np.array(list(ex.FEATURE_SETS.keys())),
        feature_set_all=np.array(ex.FEATURE_SETS["all"]),
        feature_set_original=np.array(ex.FEATURE_SETS["original"]),
        feature_set_rate=np.array(ex.FEATURE_SETS["rate"]),
        feature_set_timing=np.array(ex.FEATURE_SETS["timing"]),
        feature_set_rhythm=np.array(ex.FEATURE_SETS["rhythm"]),
        reservoir=np.array([ex.NUM_NEURONS, ex.NUM_OUTPUT_NEURONS, ex.LEAK_COEFFICIENT,
                            ex.REFRACTORY_PERIOD, ex.MEMBRANE_THRESHOLD, ex.SMALL_WORLD_P,
                            ex.SMALL_WORLD_K], dtype=np.float64),
        frontend=np.array([cd.SAMPLE_RATE, cd.DURATION, cd.TIME_BINS, cd.HYSTERESIS_GAP,
                           cd.MAX_SAMPLES_PER_CLASS, cd.REDUNDANCY_FACTOR], dtype=np.float64),
    )
    nonfinite(cd, holder)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


def nonfinite(cd, holder):
    """Round 5: the reference's post-filterbank code on spectrograms that hold a NaN or an Inf --
    `spec_db.max()` / `.min()` / `np.maximum` (create_dataset.py:59-67) propagate NaN, so the whole clip normalises to NaN
    (and its raster is all zeros).  Own random stream and own file: the other fixtures keep their bytes."""
    rs = np.random.RandomState(20261005)
    audio = np.zeros(16000, dtype=np.float32)
    post = {}
    with np.errstate(all="ignore"):
        for name, F, pos, val in (("gt_nan", 16, (3, 17), np.nan), ("gt_nan_first", 8, (0, 0), np.nan),
                                  ("gt_nan_last", 8, (7, 97), np.nan), ("gt_inf", 16, (5, 40), np.inf)):
            spec = np.abs(rs.randn(F, 98)) * 1e-2
            spec[pos] = val
            holder["spec"] = spec
            post[f"{name}_in"] = spec
            post[f"{name}_out"] = cd.audio_to_spectrogram(audio, F, "gammatone")
        for name, F, pos, val in (("mel_nan", 40, (11, 60), np.nan), ("mel_nan_first", 13, (0, 0), np.nan)):
            db = np.maximum(-np.abs(rs.randn(F, 101)) * 25.0, -80.0).astype(np.float32)
            db[rs.randint(F), rs.randint(101)] = 0.0
            db[pos] = val
            holder["spec"] = None
            holder["db"] = db
            post[f"{name}_in"] = db
            post[f"{name}_out"] = cd.audio_to_spectrogram(audio, F, "mel")
        for k in list(post):
            if k.endswith("_out"):
                post[k[:-4] + "_raster"] = cd.convert_spectrogram_to_spikes_hysteresis(
                    post[k], cd.SPIKE_THRESHOLDS, cd.HYSTERESIS_GAP)
    np.savez_compressed(os.path.join(OUT, "postfilter_nonfinite.npz"), **post)


if __name__ == "__main__":
    if sys.argv[1:] == ["nonfinite"]:
        cd_, _, holder_ = load_reference()
        nonfinite(cd_, holder_)
    else:
        main()
