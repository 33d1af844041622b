"""Host side of the mel branch (/root/reference/create_dataset.py:43-48): parameter tables for the
HIP kernels (Hann window, FFT twiddles, Slaney mel basis as librosa 0.11 builds it by default) and
the batched launch.  librosa itself is not available; SPEC.md §1.5 states what is restated."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

SAMPLE_RATE = 16000
N_FFT = 2048
AMIN = 1e-10
TOP_DB = 80.0


def _slaney_hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f * 3.0 / 200.0
    log = 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) / (np.log(6.4) / 27.0)
    return np.where(f >= 1000.0, log, lin)


def _slaney_mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    lin = m * 200.0 / 3.0
    log = 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0))
    return np.where(m >= 15.0, log, lin)


def mel_basis(sr: float, n_fft: int, n_mels: int):
    """Triangular Slaney-normalised mel filters, float32 (n_mels, 1 + n_fft//2), plus the
    half-open non-zero bin range [lo, hi) of every filter."""
    freqs = np.arange(1 + n_fft // 2, dtype=np.float64) * (sr / n_fft)
    edges = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(0.0), _slaney_hz_to_mel(sr / 2.0), n_mels + 2))
    width = np.diff(edges)
    w = np.zeros((n_mels, len(freqs)), dtype=np.float64)
    for i in range(n_mels):
        rise = (freqs - edges[i]) / width[i]
        fall = (edges[i + 2] - freqs) / width[i + 1]
        w[i] = np.maximum(0.0, np.minimum(rise, fall))
    w *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    w32 = w.astype(np.float32)
    lo = np.zeros(n_mels, dtype=np.int32)
    hi = np.zeros(n_mels, dtype=np.int32)
    for i in range(n_mels):
        nz = np.nonzero(w32[i])[0]
        if len(nz):
            lo[i], hi[i] = nz[0], nz[-1] + 1
    return w32, lo, hi


class MelSpectrogram:
    def __init__(self, n_mels: int, n_samples: int, time_bins: int, device):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.n_mels = int(n_mels)
        self.n_samples = int(n_samples)
        self.hop = max(1, int(n_samples / time_bins))              # create_dataset.py:44
        self.n_frames = 1 + n_samples // self.hop                  # centred: 1 + floor(L / hop)
        n = np.arange(N_FFT)
        window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)       # periodic Hann, float64
        k = np.arange(N_FFT // 2)
        tw = np.stack([np.cos(2.0 * np.pi * k / N_FFT), -np.sin(2.0 * np.pi * k / N_FFT)], axis=1)
        basis, lo, hi = mel_basis(SAMPLE_RATE, N_FFT, self.n_mels)
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self.window, self.twiddle = to(window), to(tw)
        self.basis, self.lo, self.hi = to(basis), to(lo), to(hi)
        # scratch of the one-launch front end per (batch size, stream).  A workspace carries the clips' arrival counters,
        # so two launches that may overlap (different streams) must not share one; pipeline.HotPath owns one per stream.
        self._ws = {}

    def power(self, audio: torch.Tensor) -> torch.Tensor:
        B = audio.shape[0]
        out = torch.empty((B, self.n_mels, self.n_frames), dtype=torch.float32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(self.lib.lsm_mel_power_f32(
            p(audio), B, self.n_samples, N_FFT, self.hop, self.n_frames, p(self.window),
            p(self.twiddle), p(self.basis), p(self.lo), p(self.hi), self.n_mels, p(out),
            torch.cuda.current_stream(self.device).cuda_stream), "lsm_mel_power_f32")
        return out

    def workspace_bytes(self, n_clips: int) -> int:
        return int(self.lib.lsm_mel_spikes_workspace(int(n_clips), self.n_mels, self.n_frames))

    def new_workspace(self, n_clips: int) -> torch.Tensor:
        """Zeroed scratch of the one-launch front end for batches of up to `n_clips` clips (the per-clip counters at its
        start must be zero on entry; every launch leaves them zero, so one workspace serves a stream of batches)."""
        return torch.zeros((max(self.workspace_bytes(n_clips), 256),), dtype=torch.uint8, device=self.device)

    def fits_one_launch(self, time_bins: int, n_thr: int) -> bool:
        """The finishing workgroup keeps its latch bit rows (64 rows x thresholds x time-bin words, twice) in the four waves'
        point buffers (csrc/spikes_body.h: spikes_lds_bytes; csrc/mel.hip: 4 x 17 408 bytes)."""
        return 128 + 2 * 64 * max(n_thr, 1) * ((time_bins + 31) // 32) * 4 <= 4 * 17408

    def spikes(self, audio: torch.Tensor, on: np.ndarray, off: np.ndarray, time_bins: int, redundancy: int,
               raster_out: torch.Tensor | None = None, workspace: torch.Tensor | None = None) -> torch.Tensor:
        """(B, n_samples) float32 -> uint8 raster (B, n_mels*redundancy, time_bins*len(on)) in ONE launch
        (`lsm_mel_spikes_f32`): mel power -> dB -> normalise -> resize -> hysteresis encoder."""
        B = audio.shape[0]
        shape = (B, self.n_mels * redundancy, time_bins * len(on))
        raster = raster_out if raster_out is not None else torch.empty(shape, dtype=torch.uint8, device=self.device)
        if tuple(raster.shape) != shape or raster.dtype != torch.uint8 or not raster.is_contiguous():
            raise ValueError(f"raster_out must be a contiguous uint8 {shape} tensor")
        need = self.workspace_bytes(B)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if workspace is None:
            # launches on ONE stream run in order: they may share a workspace.  Keyed by the stream OBJECT (a raw handle
            # can be handed out again after its stream has died) and bounded: the oldest entry goes
            key = (B, torch.cuda.current_stream(self.device))
            ws = self._ws.get(key)
            if ws is None:
                while len(self._ws) >= 16:
                    self._ws.pop(next(iter(self._ws)))
                ws = self._ws[key] = self.new_workspace(B)
        else:
            ws = workspace
            if ws.dtype != torch.uint8 or ws.numel() < need or ws.device != raster.device or ws.data_ptr() % 256:
                raise ValueError(f"workspace must be a zero-initialised, 256-byte aligned uint8 tensor of >= {need} bytes")
        p = lambda t: C.c_void_p(t.data_ptr())
        h = lambda a: C.c_void_p(a.ctypes.data)
        on = np.ascontiguousarray(on, dtype=np.float32)
        off = np.ascontiguousarray(off, dtype=np.float32)
        _lib.check(self.lib.lsm_mel_spikes_f32(
            p(audio), B, self.n_samples, N_FFT, self.hop, self.n_frames, p(self.window), p(self.twiddle),
            p(self.basis), p(self.lo), p(self.hi), self.n_mels, C.c_float(AMIN), C.c_float(TOP_DB), int(time_bins),
            h(on), h(off), len(on), int(redundancy), p(raster), p(ws), int(ws.numel()), stream), "lsm_mel_spikes_f32")
        return raster

    def power_db(self, audio: torch.Tensor) -> torch.Tensor:
        """(B, n_samples) float32 -> power_to_db(melspectrogram) float32 (B, n_mels, n_frames)."""
        S = self.power(audio)
        out = torch.empty_like(S)
        _lib.check(self.lib.lsm_power_to_db_f32(
            C.c_void_p(S.data_ptr()), S.shape[0], S.shape[1] * S.shape[2], C.c_float(AMIN),
            C.c_float(TOP_DB), C.c_void_p(out.data_ptr()), torch.cuda.current_stream(self.device).cuda_stream),
            "lsm_power_to_db_f32")
        return out
