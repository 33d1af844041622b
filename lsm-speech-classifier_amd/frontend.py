"""Host mirror of the reference's front end (/root/reference/create_dataset.py:39-104) over the
HIP kernels: same function names, argument meaning and defaults; batched variants underneath.

Host code here only builds small parameter tables (filter coefficients, thresholds) and moves
pointers; all per-sample arithmetic runs in liblsm_hip.so.  No CPU fallback exists.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib

# create_dataset.py:10-17
SAMPLE_RATE = 16000
DURATION = 1.0
TIME_BINS = 100
SPIKE_THRESHOLDS = [0.70, 0.80, 0.90, 0.95]
HYSTERESIS_GAP = 0.1
REDUNDANCY_FACTOR = 1
# gtgram call at create_dataset.py:51-58
GT_WINDOW_TIME = 0.025
GT_F_MIN = 50

_EAR_Q = 9.26449          # Glasberg & Moore
_MIN_BW = 24.7


def gammatone_filter_table(fs: float, channels: int, f_min: float) -> np.ndarray:
    """(channels, 10) float64 rows [A0, A11, A12, A13, A14, A2, B0, B1, B2, gain], ordered from
    the lowest to the highest centre frequency: Slaney's ERB filterbank design (Apple TR #35) on
    ERB-spaced centre frequencies between f_min and fs/2, as gammatone==1.0.3 computes it for
    gtgram (the package itself is not available; SPEC.md §1.1)."""
    frac = np.arange(1, channels + 1) / channels
    c = _EAR_Q * _MIN_BW
    high = fs / 2
    cf = -c + np.exp(frac * (np.log(f_min + c) - np.log(high + c))) * (high + c)
    cf = cf[::-1].copy()
    T = 1.0 / fs
    B = 1.019 * 2 * np.pi * (cf / _EAR_Q + _MIN_BW)
    arg = 2 * cf * np.pi * T
    ebt = np.exp(B * T)
    vec = np.exp(2j * arg)
    rt_pos, rt_neg = np.sqrt(3 + 2 ** 1.5), np.sqrt(3 - 2 ** 1.5)
    common = -T * np.exp(-(B * T))
    ks = [np.cos(arg) + rt_pos * np.sin(arg), np.cos(arg) - rt_pos * np.sin(arg),
          np.cos(arg) + rt_neg * np.sin(arg), np.cos(arg) - rt_neg * np.sin(arg)]
    gain_arg = np.exp(1j * arg - B * T)
    gain = np.abs((vec - gain_arg * ks[0]) * (vec - gain_arg * ks[1]) * (vec - gain_arg * ks[2])
                  * (vec - gain_arg * ks[3]) * (T * ebt / (-1 / ebt + 1 + vec * (1 - ebt))) ** 4)
    tab = np.empty((channels, 10), dtype=np.float64)
    tab[:, 0] = T
    for q in range(4):
        tab[:, 1 + q] = common * ks[q]
    tab[:, 5] = 0.0
    tab[:, 6] = 1.0
    tab[:, 7] = -2 * np.cos(arg) / ebt
    tab[:, 8] = np.exp(-2 * B * T)
    tab[:, 9] = gain
    return tab


def coef_flags(tab: np.ndarray) -> int:
    """Properties of a coefficient table that let the kernel skip work without changing a bit:
    bit 0: every A2 is exactly zero; bit 1: no gain has an all-ones significand (the one case in
    which the FMA division sequence is not guaranteed to round like a true division); bit 2: A0/B0 is the
    same float64 for every channel (the fused kernel then forms b0*x once for the two channels of a lane)."""
    flags = 0
    if not np.any(tab[:, 5]):
        flags |= 1
    mant = np.ascontiguousarray(tab[:, 9]).view(np.uint64) & np.uint64((1 << 52) - 1)
    if np.all(np.isfinite(tab[:, 9])) and not np.any(mant == np.uint64((1 << 52) - 1)) \
            and np.all(np.abs(tab[:, 9]) > 1e-200) and np.all(np.abs(tab[:, 9]) < 1e200):
        flags |= 2
    b0 = tab[:, 0] / tab[:, 6]
    if np.all(np.isfinite(b0)) and np.all(b0.view(np.uint64) == b0[:1].view(np.uint64)):
        flags |= 4
    return flags


def _round_half_away(x: float) -> int:
    return int(np.sign(x) * np.floor(np.abs(x) + 0.5))


def gtgram_strides(fs: float, window_time: float, hop_time: float, n_samples: int):
    nwin = _round_half_away(window_time * fs)
    hop = _round_half_away(hop_time * fs)
    return nwin, hop, 1 + int(np.floor((n_samples - nwin) / hop))


def threshold_tables(thresholds, gap: float, dtype):
    """ON thresholds sorted descending and the Python-float ``thr - gap`` OFF bounds
    (create_dataset.py:87-89), rounded to the spectrogram dtype the comparison happens in."""
    thr = sorted(thresholds, reverse=True)
    on = np.array([dtype(t) for t in thr], dtype=dtype)
    off = np.array([dtype(t - gap) for t in thr], dtype=dtype)
    return on, off


def _stream(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _dev(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def _host(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


class SpikeFrontEnd:
    """Batched filterbank -> dB -> normalise -> resize -> hysteresis encoder on one GPU."""

    def __init__(self, n_filters: int, filterbank: str = "gammatone", device=None,
                 redundancy: int = REDUNDANCY_FACTOR, thresholds=None, gap: float = HYSTERESIS_GAP,
                 time_bins: int = TIME_BINS, n_samples: int = int(SAMPLE_RATE * DURATION)):
        if filterbank not in ("gammatone", "mel"):
            raise ValueError(f"filterbank must be 'mel' or 'gammatone', got {filterbank!r}")
        if int(n_filters) < 1:
            raise ValueError(f"n_filters must be >= 1, got {n_filters}")
        if filterbank == "gammatone" and int(n_filters) == 1:
            # SPEC.md 1.1: with ONE gammatone channel NumPy sums each 400-sample window pairwise (the fancy-index result
            # (1, 400) is contiguous along the reduced axis), with two or more element by element; the kernels implement
            # the latter.  One filter is refused rather than answered 6e-16 beside the reference.
            raise ValueError("the gammatone branch needs n_filters >= 2: with a single channel the reference's window sums "
                             "take NumPy's pairwise order, which the GPU filterbank does not restate (SPEC.md 1.1); use "
                             "--filterbank mel for one filter")
        _lib.require_gpu()
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type == "cuda" and self.device.index is None:      # pin the device NOW: later calls may
            self.device = torch.device("cuda", torch.cuda.current_device())   # come under another current device
        self.n_filters = int(n_filters)
        self.filterbank = filterbank
        self.redundancy = int(redundancy)
        self.thresholds = list(SPIKE_THRESHOLDS if thresholds is None else thresholds)
        self.gap = float(gap)
        self.time_bins = int(time_bins)
        self.n_samples = int(n_samples)
        if filterbank == "gammatone":
            hop_time = self.n_samples / (SAMPLE_RATE * self.time_bins)       # create_dataset.py:50
            self.nwin, self.hop, self.ncols = gtgram_strides(SAMPLE_RATE, GT_WINDOW_TIME, hop_time,
                                                             self.n_samples)
            tab = gammatone_filter_table(SAMPLE_RATE, self.n_filters, GT_F_MIN)
            self.coefs = torch.from_numpy(tab).to(self.device)
            self.coef_flags = coef_flags(tab)
            self.dtype = torch.float64
        else:
            from . import mel as _mel
            self._mel = _mel.MelSpectrogram(self.n_filters, self.n_samples, self.time_bins, self.device)
            self.ncols = self._mel.n_frames
            self.dtype = torch.float32

    @property
    def n_channels(self) -> int:
        return self.n_filters * self.redundancy

    @property
    def n_steps(self) -> int:
        return self.time_bins * len(self.thresholds)

    def _stream(self) -> int:
        """The current stream of THIS front end's device (not of whatever device is current)."""
        return torch.cuda.current_stream(self.device).cuda_stream

    def _indexed_device(self) -> torch.device:
        d = self.device                 # pinned by __init__; an instance built without it may still say "cuda"
        return d if d.index is not None or d.type != "cuda" else torch.device("cuda", torch.cuda.current_device())

    def workspace_elems(self, n_clips: int) -> int:
        """float64 elements of the fused launch's scratch for a batch of `n_clips` (0 for the mel branch)."""
        if self.filterbank != "gammatone":
            return 0
        return max(int(self.lib.lsm_gammatone_spikes_workspace(int(n_clips), self.n_filters, self.ncols)), 8) // 8

    def will_fuse(self) -> bool:
        """Whether `encode()` takes the one-launch route by default.  ONE decision for `encode()` and for
        `pipeline.HotPath`, which hands its own raster buffers to the fused launch only.  Gammatone: up to 1024 filters
        (`lsm_gammatone_spikes_f64`).  Mel: the one-launch route (`lsm_mel_spikes_f32`, `encode(fused=True)`) exists and
        gives the same rasters, but measured slower than the three split launches at the reference's CPU-runnable shape
        (40 filters, 200 clips: 0.318 against 0.305 ms per step, profiles/r04_mel_one_launch_ab.txt), so it is taken by
        default only with LSM_MEL_ONE_LAUNCH=1.
        LSM_FRONTEND_SPLIT=1: diagnostic switch for same-box A/B runs of the two routes (exp/r03_fused_sweep.sh)."""
        if os.environ.get("LSM_FRONTEND_SPLIT") == "1":
            return False
        if self.filterbank == "mel":
            return (os.environ.get("LSM_MEL_ONE_LAUNCH") == "1"
                    and self._mel.fits_one_launch(self.time_bins, len(self.thresholds)))
        return self.n_filters <= 1024

    def new_workspace(self, n_clips: int) -> torch.Tensor:
        """Scratch of the one-launch route for batches of up to `n_clips` clips, to pass as `encode(workspace=...)`: one per
        stream when launches may overlap (the mel scratch carries per-clip arrival counters and starts zeroed)."""
        if self.filterbank == "mel":
            return self._mel.new_workspace(n_clips)
        return torch.empty((self.workspace_elems(n_clips),), dtype=torch.float64, device=self.device)

    def _audio(self, audio) -> torch.Tensor:
        if isinstance(audio, np.ndarray):
            audio = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32))
        audio = audio.to(self.device, dtype=torch.float32).contiguous()
        if audio.dim() == 1:
            audio = audio[None]
        if audio.shape[1] != self.n_samples:
            raise ValueError(f"clips must have {self.n_samples} samples, got {audio.shape[1]}")
        return audio

    def spectrogram_db(self, audio, want_spec: bool = False):
        """(B, n_samples) float32 -> dB spectrogram (B, F, ncols) in the filterbank's dtype
        (gammatone: float64 20*log10(gtgram+1e-9), un-floored; mel: float32 power_to_db)."""
        audio = self._audio(audio)
        B = audio.shape[0]
        if self.filterbank == "mel":
            with torch.cuda.device(self.device):
                return self._mel.power_db(audio), None
        with torch.cuda.device(self.device):       # launch on THIS device's current stream
            db = torch.empty((B, self.n_filters, self.ncols), dtype=torch.float64, device=self.device)
            spec = torch.empty_like(db) if want_spec else None
            _lib.check(self.lib.lsm_gammatone_spec_f64(
                _dev(audio), B, self.n_samples, _dev(self.coefs), self.n_filters, self.nwin, self.hop,
                self.ncols, _dev(spec) if want_spec else None, _dev(db), self.coef_flags, self._stream()),
                "lsm_gammatone_spec_f64")
        return db, spec

    def spikes_from_db(self, db: torch.Tensor, want_norm: bool = False, want_raster: bool = True):
        """dB spectrogram -> (uint8 raster (B, C, n_steps), normalised spectrogram or None)."""
        B = db.shape[0]
        f64 = db.dtype == torch.float64
        np_dt = np.float64 if f64 else np.float32
        on, off = threshold_tables(self.thresholds, self.gap, np_dt)
        if db.device != self._indexed_device():
            raise ValueError(f"spectrogram on {db.device}, front end on {self.device}")
        db = db.contiguous()
        with torch.cuda.device(self.device):
            raster = (torch.empty((B, self.n_channels, self.n_steps), dtype=torch.uint8, device=self.device)
                      if want_raster else None)
            norm = (torch.empty((B, self.n_filters, self.time_bins), dtype=db.dtype, device=self.device)
                    if want_norm else None)
            fn = self.lib.lsm_spec_to_spikes_f64 if f64 else self.lib.lsm_spec_to_spikes_f32
            _lib.check(fn(_dev(db), B, self.n_filters, db.shape[2], self.time_bins,
                          1 if self.filterbank == "gammatone" else 0, _host(on), _host(off), len(on),
                          self.redundancy, _dev(raster) if want_raster else None,
                          _dev(norm) if want_norm else None, self._stream()), "lsm_spec_to_spikes")
        return raster, norm

    def encode(self, audio, fused: bool | None = None, low_latency: bool = False,
               raster_out: torch.Tensor | None = None, workspace: torch.Tensor | None = None,
               share_lds: bool = False) -> torch.Tensor:
        """audio (B, n_samples) -> uint8 spike raster (B, C, n_steps) on the device.  The gammatone branch is one
        launch (`lsm_gammatone_spikes_f64`); `fused=False` takes the split entry points (identical rasters), which is
        also what the mel branch takes by default (its one-launch route, `fused=True` -> `lsm_mel_spikes_f32`, measured
        slower) and what shapes too large for the one-launch kernels use (`will_fuse()`).  `low_latency`: the
        fused launch in its one-chain layout (twice the waves, each half as long) -- for a batch that meets an idle
        GPU; `pipeline.HotPath` asks for it when none of its front ends is in flight.  `share_lds`: the launch runs
        beside LDS-hungry workgroups of another kernel (the ring-row reservoir kernel, two 64 KB clips per CU): it then
        asks for the 27 KB of LDS it uses instead of reserving half a CU's (the reservation is a placement tool for small
        launches, `lsm_gammatone_spikes_f64` flag bit 1).  `raster_out` / `workspace`
        (fused launch only): caller-owned uint8 (B, C, n_steps) output and scratch from `new_workspace(B)` (gammatone:
        float64, at least `workspace_elems(B)` elements; mel: zero-initialised bytes), so that a steady stream of
        batches makes no allocator call at all."""
        if fused is None:
            fused = self.will_fuse()
        if not fused:
            if raster_out is not None or workspace is not None:
                # the split route allocates its own outputs on the current stream: a caller that owns the raster
                # buffer (pipeline.HotPath's rings) must not be handed another tensor silently
                raise ValueError("raster_out / workspace belong to the fused launch; this front end takes the split "
                                 "route (mel branch, > 1024 filters or LSM_FRONTEND_SPLIT=1)")
            db, _ = self.spectrogram_db(audio)
            raster, _ = self.spikes_from_db(db)
            return raster
        audio = self._audio(audio)
        B = audio.shape[0]
        if self.filterbank == "mel":
            if not self._mel.fits_one_launch(self.time_bins, len(self.thresholds)):
                raise ValueError(f"{self.n_filters} mel filters do not fit the one-launch front end; use fused=False")
            on32, off32 = threshold_tables(self.thresholds, self.gap, np.float32)
            with torch.cuda.device(self.device):
                return self._mel.spikes(audio, on32, off32, self.time_bins, self.redundancy, raster_out, workspace)
        on, off = threshold_tables(self.thresholds, self.gap, np.float64)
        with torch.cuda.device(self.device):
            shape = (B, self.n_channels, self.n_steps)
            if raster_out is not None:
                if (raster_out.dtype != torch.uint8 or tuple(raster_out.shape) != shape or not raster_out.is_contiguous()
                        or raster_out.device != self.device):
                    raise ValueError(f"raster_out must be a contiguous uint8 {shape} tensor on {self.device}")
                raster = raster_out
            else:
                raster = torch.empty(shape, dtype=torch.uint8, device=self.device)
            ws_bytes = int(self.lib.lsm_gammatone_spikes_workspace(B, self.n_filters, self.ncols))
            if workspace is not None:
                if (workspace.dtype != torch.float64 or workspace.numel() * 8 < ws_bytes or not workspace.is_contiguous()
                        or workspace.device != self.device):
                    raise ValueError(f"workspace must be a contiguous float64 tensor of >= {ws_bytes // 8} elements on {self.device}")
                ws = workspace
            else:
                ws = torch.empty((max(ws_bytes, 8) // 8,), dtype=torch.float64, device=self.device)
            _lib.check(self.lib.lsm_gammatone_spikes_f64(
                _dev(audio), B, self.n_samples, _dev(self.coefs), self.n_filters, self.nwin, self.hop,
                self.ncols, self.time_bins, _host(on), _host(off), len(on), self.redundancy, _dev(raster),
                _dev(ws), ws_bytes, self.coef_flags, (1 if low_latency else 0) | (2 if share_lds else 0), self._stream()),
                "lsm_gammatone_spikes_f64")
        return raster


_FRONT_ENDS: dict = {}


def _front_end(n_filters: int, filterbank: str, n_samples: int) -> SpikeFrontEnd:
    key = (int(n_filters), filterbank, int(n_samples))
    if key not in _FRONT_ENDS:
        _FRONT_ENDS[key] = SpikeFrontEnd(n_filters, filterbank, n_samples=n_samples)
    return _FRONT_ENDS[key]


def audio_to_spectrogram(audio: np.ndarray, n_filters: int, filterbank: str) -> np.ndarray:
    """create_dataset.py:39-78 for one clip: (n_samples,) -> (n_filters, TIME_BINS) in [0, 1]
    (float64 for gammatone, float32 for mel and for the flat-input zeros, as in the reference)."""
    fe = _front_end(n_filters, filterbank, len(audio))
    db, _ = fe.spectrogram_db(np.asarray(audio, dtype=np.float32))
    _, norm = fe.spikes_from_db(db, want_norm=True, want_raster=False)
    hi, lo = db.max(), db.min()
    if filterbank == "gammatone":
        lo = torch.maximum(lo, hi - 80.0)                  # create_dataset.py:60
    if float(hi - lo) < 1e-8:                              # create_dataset.py:64-65
        return np.zeros((n_filters, TIME_BINS), dtype=np.float32)
    return norm[0].cpu().numpy()


def convert_spectrogram_to_spikes_hysteresis(spectrogram, thresholds, hysteresis_gap=0.05):
    """create_dataset.py:81-98: (F, n_time) float -> (F, n_time*len(thresholds)) uint8."""
    _lib.require_gpu()
    lib = _lib.load()
    spec = np.ascontiguousarray(spectrogram)
    if spec.dtype not in (np.float32, np.float64):
        spec = spec.astype(np.float64)
    F, n_time = spec.shape
    on, off = threshold_tables(thresholds, hysteresis_gap, spec.dtype.type)
    dspec = torch.from_numpy(spec).cuda()
    out = torch.empty((F, n_time * len(on)), dtype=torch.uint8, device="cuda")
    fn = lib.lsm_encode_hysteresis_f64 if spec.dtype == np.float64 else lib.lsm_encode_hysteresis_f32
    _lib.check(fn(_dev(dspec), F, n_time, _host(on), _host(off), len(on), _dev(out), _stream()),
               "lsm_encode_hysteresis")
    return out.cpu().numpy()


def create_pure_redundancy(spike_train: np.ndarray, redundancy_factor: int) -> np.ndarray:
    """create_dataset.py:101-104 (host copy; the batched path repeats rows inside the kernel)."""
    return np.repeat(spike_train, redundancy_factor, axis=0)


# ---- bit-packed rasters: the packed variant of File 1 --------------------------

def pack_raster(raster: torch.Tensor) -> torch.Tensor:
    """(..., T) uint8 on the GPU (any non-zero byte is a spike) -> (..., ceil(T/8)) uint8, time step
    8q+k in bit k of byte q (``numpy.packbits(x != 0, axis=-1, bitorder="little")``)."""
    _lib.require_gpu()
    if raster.dtype != torch.uint8 or not raster.is_cuda:
        raise ValueError("pack_raster: expected a uint8 CUDA tensor")
    raster = raster.contiguous()
    T = raster.shape[-1]
    rows = raster.numel() // T if T else 0
    out = torch.empty(raster.shape[:-1] + ((T + 7) // 8,), dtype=torch.uint8, device=raster.device)
    _lib.check(_lib.load().lsm_raster_pack_bits(_dev(raster), rows, T, _dev(out), _stream()),
               "lsm_raster_pack_bits")
    return out


def unpack_raster(packed: torch.Tensor, n_steps: int) -> torch.Tensor:
    """Inverse of :func:`pack_raster`: (..., ceil(T/8)) uint8 on the GPU -> (..., T) bytes of 0/1."""
    _lib.require_gpu()
    if packed.dtype != torch.uint8 or not packed.is_cuda:
        raise ValueError("unpack_raster: expected a uint8 CUDA tensor")
    if packed.shape[-1] != (n_steps + 7) // 8:
        raise ValueError(f"unpack_raster: last dimension {packed.shape[-1]} does not hold {n_steps} steps")
    packed = packed.contiguous()
    rows = packed.numel() // packed.shape[-1] if packed.shape[-1] else 0
    out = torch.empty(packed.shape[:-1] + (n_steps,), dtype=torch.uint8, device=packed.device)
    _lib.check(_lib.load().lsm_raster_unpack_bits(_dev(packed), rows, n_steps, _dev(out), _stream()),
               "lsm_raster_unpack_bits")
    return out
