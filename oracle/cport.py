"""ctypes front to oracle/liblsm_oracle.so (plain-C oracle) — TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblsm_oracle.so")
_lib = None

FEATURE_KEYS = ['spike_counts', 'spike_variances', 'mean_spike_times', 'first_spike_times',
                'last_spike_times', 'mean_isi', 'isi_variances', 'burst_counts']


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "lsm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liblsm_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def gammatone_spec(audio, coefs, nwin, hop, ncols):
    audio = np.ascontiguousarray(audio, dtype=np.float32)
    coefs = np.ascontiguousarray(coefs, dtype=np.float64)
    F = coefs.shape[0]
    out = np.empty((F, ncols), dtype=np.float64)
    rc = lib().orc_gammatone_spec(_p(audio, C.c_float), C.c_int(audio.shape[0]),
                                  _p(coefs, C.c_double), F, nwin, hop, ncols, _p(out, C.c_double))
    if rc:
        raise ValueError(f"orc_gammatone_spec failed: {rc}")
    return out


def gammatone_db(spec):
    out = np.array(spec, dtype=np.float64, order="C", copy=True)
    lib().orc_gammatone_db(_p(out, C.c_double), C.c_int(out.size))
    return out


def normalise_resize(db, time_bins=100):
    db = np.ascontiguousarray(db)
    F, ncols = db.shape
    if db.dtype == np.float32:
        out = np.empty((F, time_bins), dtype=np.float32)
        lib().orc_normalise_resize_f32(_p(db, C.c_float), F, ncols, time_bins, _p(out, C.c_float))
    else:
        db = db.astype(np.float64, copy=False)
        out = np.empty((F, time_bins), dtype=np.float64)
        lib().orc_normalise_resize_f64(_p(db, C.c_double), F, ncols, time_bins,
                                       _p(out, C.c_double))
    return out


def threshold_tables(thresholds, gap, dtype):
    """Descending ON thresholds and Python-float ``thr - gap`` OFF bounds, rounded to dtype."""
    thr = sorted(thresholds, reverse=True)
    on = np.array([dtype(t) for t in thr], dtype=dtype)
    off = np.array([dtype(t - gap) for t in thr], dtype=dtype)
    return on, off


def encode_hysteresis(spec, thresholds, gap):
    spec = np.ascontiguousarray(spec)
    F, Tb = spec.shape
    if spec.dtype == np.float32:
        on, off = threshold_tables(thresholds, gap, np.float32)
        out = np.empty((F, Tb * len(on)), dtype=np.uint8)
        lib().orc_encode_hysteresis_f32(_p(spec, C.c_float), F, Tb, _p(on, C.c_float),
                                        _p(off, C.c_float), len(on), _p(out, C.c_uint8))
    else:
        spec = spec.astype(np.float64, copy=False)
        on, off = threshold_tables(thresholds, gap, np.float64)
        out = np.empty((F, Tb * len(on)), dtype=np.uint8)
        lib().orc_encode_hysteresis_f64(_p(spec, C.c_double), F, Tb, _p(on, C.c_double),
                                        _p(off, C.c_double), len(on), _p(out, C.c_uint8))
    return out


def gammatone_frontend_batch(audio, coefs, nwin, hop, ncols, thresholds, gap, time_bins=100, n_threads=1):
    """(n, n_samples) float32 -> (n, F, time_bins*len(thresholds)) uint8: the whole gammatone front end of a
    batch, one clip per OpenMP thread (the per-clip functions above, chained in C)."""
    audio = np.ascontiguousarray(audio, dtype=np.float32)
    coefs = np.ascontiguousarray(coefs, dtype=np.float64)
    n, L = audio.shape
    F = coefs.shape[0]
    on, off = threshold_tables(thresholds, gap, np.float64)
    out = np.empty((n, F, time_bins * len(on)), dtype=np.uint8)
    rc = lib().orc_gammatone_frontend_batch(
        _p(audio, C.c_float), n, L, _p(coefs, C.c_double), F, nwin, hop, ncols, time_bins,
        _p(on, C.c_double), _p(off, C.c_double), len(on), int(n_threads), _p(out, C.c_uint8))
    if rc:
        raise ValueError(f"orc_gammatone_frontend_batch failed: {rc}")
    return out


def _res_args(res):
    return (res.num_neurons, res.n_channels)


def lif_run(res, raster, keys=None, want_spikes=True, want_trace=False):
    """One clip. Returns (features (n_keys*n_out,) float32, spike_matrix or None, v_trace or None)."""
    raster = np.ascontiguousarray(raster, dtype=np.uint8)
    Cn, T = raster.shape
    N = res.num_neurons
    keys = FEATURE_KEYS if keys is None else list(keys)
    key_ids = np.array([FEATURE_KEYS.index(k) for k in keys], dtype=np.int32)
    n_out = len(res.out_idx)
    feats = np.empty((len(keys), n_out), dtype=np.float32)
    sm = np.empty((T, N), dtype=np.uint8) if want_spikes else None
    vt = np.empty((T, N), dtype=np.float32) if want_trace else None
    rc = lib().orc_lif_run(
        N, Cn, T, _p(res.csr_ptr, C.c_int32), _p(res.csr_pre, C.c_int32), _p(res.csr_w, C.c_float),
        _p(res.in_ptr, C.c_int32), _p(res.in_chan, C.c_int32), C.c_float(float(res.w_in)),
        _p(res.leak, C.c_float), C.c_float(float(res.theta)), int(res.refractory_period),
        _p(raster, C.c_uint8), _p(sm, C.c_uint8), _p(vt, C.c_float),
        n_out, _p(res.out_idx, C.c_int32), int(res.burst_isi_max),
        len(keys), _p(key_ids, C.c_int32), _p(feats, C.c_float))
    if rc:
        raise ValueError(f"orc_lif_run failed: {rc}")
    return feats.reshape(-1), sm, vt


def lif_run_batch(res, rasters, keys=None, n_threads=1):
    """(B, C, T) uint8 -> (B, n_keys*n_out) float32; one clip per OpenMP thread."""
    rasters = np.ascontiguousarray(rasters, dtype=np.uint8)
    B, Cn, T = rasters.shape
    keys = FEATURE_KEYS if keys is None else list(keys)
    key_ids = np.array([FEATURE_KEYS.index(k) for k in keys], dtype=np.int32)
    n_out = len(res.out_idx)
    feats = np.empty((B, len(keys) * n_out), dtype=np.float32)
    rc = lib().orc_lif_run_batch(
        B, int(n_threads), res.num_neurons, Cn, T,
        _p(res.csr_ptr, C.c_int32), _p(res.csr_pre, C.c_int32), _p(res.csr_w, C.c_float),
        _p(res.in_ptr, C.c_int32), _p(res.in_chan, C.c_int32), C.c_float(float(res.w_in)),
        _p(res.leak, C.c_float), C.c_float(float(res.theta)), int(res.refractory_period),
        _p(rasters, C.c_uint8), n_out, _p(res.out_idx, C.c_int32), int(res.burst_isi_max),
        len(keys), _p(key_ids, C.c_int32), _p(feats, C.c_float))
    if rc:
        raise ValueError(f"orc_lif_run_batch failed: {rc}")
    return feats
