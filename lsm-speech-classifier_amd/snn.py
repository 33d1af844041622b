"""Host mirror of the reference's reservoir object protocol over the HIP reservoir kernel.

``SNN`` / ``SimulationParams`` stand in for ``snnpy.snn.SNN`` / ``SimulationParams`` as the
reference uses them (/root/reference/extract_lsm_features.py:2,79-83,100,109-116,164-188):
``SNN(simulation_params=P)``, ``.reset()``, ``.set_input_spike_times(u8 (C, T))``, ``.simulate()``,
``.extract_features_from_spikes() -> dict``, attributes ``.num_neurons`` and ``.spike_matrix``
(T, N).  ``run_batch`` is the batched path ``extract_all_features`` uses.  All arithmetic runs
in liblsm_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .reservoir import Reservoir, SimulationParams, build_reservoir

# FEATURE_SETS['all'] order (extract_lsm_features.py:20-22) = key ids of the C ABI
FEATURE_KEYS = ['spike_counts', 'spike_variances', 'mean_spike_times', 'first_spike_times',
                'last_spike_times', 'mean_isi', 'isi_variances', 'burst_counts']

__all__ = ["SNN", "SimulationParams", "FEATURE_KEYS"]


def _dev(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _host(a):
    return C.c_void_p(a.ctypes.data)


class SNN:
    def __init__(self, simulation_params: SimulationParams, n_channels: int | None = None,
                 device=None, reservoir: Reservoir | None = None):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.params = simulation_params
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type == "cuda" and self.device.index is None:        # pin it now (as frontend.SpikeFrontEnd)
            self.device = torch.device("cuda", torch.cuda.current_device())
        if reservoir is None:
            if n_channels is None:
                ist = simulation_params.input_spike_times
                if ist is None:
                    raise ValueError("SNN needs n_channels or simulation_params.input_spike_times")
                n_channels = int(np.asarray(ist).shape[0])
            reservoir = build_reservoir(simulation_params, n_channels)
        self.reservoir = reservoir
        self.num_neurons = reservoir.num_neurons
        self.n_channels = reservoir.n_channels
        self.num_output_neurons = len(reservoir.out_idx)
        self._handle = C.c_void_p()
        self._compute_units = torch.cuda.get_device_properties(self.device).multi_processor_count
        r = reservoir
        with torch.cuda.device(self.device):
            _lib.check(self.lib.lsm_reservoir_create(
                C.byref(self._handle), r.num_neurons, r.n_channels, _host(r.csc_ptr),
                _host(r.csc_post), _host(r.csc_w), _host(r.leak),
                _host(np.ascontiguousarray(r.in_tgt, dtype=np.int32)), int(r.in_fanout),
                float(r.w_in), _host(r.out_idx), len(r.out_idx), float(r.theta),
                int(r.refractory_period), int(r.burst_isi_max)), "lsm_reservoir_create")
        self._input = None
        self.spike_matrix = None
        self._features = None

    def __del__(self):
        try:
            h = getattr(self, "_handle", None)
            if h is not None and h.value:
                self.lib.lsm_reservoir_destroy(h)
                self._handle = None
        except Exception:          # interpreter shutdown: modules may already be torn down
            pass

    # ---- batched path -------------------------------------------------------------------
    def run_batch(self, spikes, feature_keys=None, want_spike_matrix=False, want_v_trace=False,
                  waves_per_clip: int = 0, packed_time_steps: int = 0, stats_out=None, features_out=None,
                  longest_first: bool | None = None):
        """spikes: uint8 (B, C, T) torch tensor on this device (or NumPy, copied).  Returns
        (features float32 (B, n_keys*N_out) device tensor, spike_matrix or None, v_trace or None);
        NaN entries are already 0 and keys are concatenated in the given order
        (extract_lsm_features.py:85-87).  With ``packed_time_steps=T`` the input is the bit-packed
        form (B, C, ceil(T/8)) of ``spikefile``: it is uploaded as it is and unpacked on the GPU.
        ``stats_out``: optional int32 (B, 2) device tensor that receives, per clip, the number of neurons
        that fired at least once and the spikes of the whole reservoir (accumulated inside the kernel).
        ``waves_per_clip``: 0 = the library's layout for a lone launch, -1 = its layout for a launch that
        shares the GPU with other kernels (``pipeline.HotPath``), else 1, 2, 4, 8 or 16.
        ``features_out``: optional contiguous float32 (B, n_keys*N_out) device tensor to write the rows into
        (e.g. this step's slice of a gather buffer) instead of a fresh one.
        ``longest_first``: start the clips with the most input spikes first (``lsm_reservoir_run_ordered``: a
        clip's time grows with its activity, and a launch of several rounds is otherwise as long as whichever
        clip starts last); results are the same either way.  Default: on when the batch has more clips than the
        GPU has compute units (``LSM_RESERVOIR_ORDER=0`` turns the default off)."""
        if isinstance(spikes, np.ndarray):
            spikes = torch.from_numpy(np.ascontiguousarray(spikes, dtype=np.uint8))
        spikes = spikes.to(self.device, dtype=torch.uint8).contiguous()
        if packed_time_steps:
            from .frontend import unpack_raster
            with torch.cuda.device(self.device):
                spikes = unpack_raster(spikes, int(packed_time_steps))
        if spikes.dim() != 3 or spikes.shape[1] != self.n_channels:
            raise ValueError(f"spikes must be (B, {self.n_channels}, T), got {tuple(spikes.shape)}")
        B, _, T = spikes.shape
        keys = FEATURE_KEYS if feature_keys is None else [k for k in feature_keys if k in FEATURE_KEYS]
        key_ids = np.array([FEATURE_KEYS.index(k) for k in keys], dtype=np.int32)
        if features_out is not None:
            if (features_out.dtype != torch.float32 or tuple(features_out.shape) != (B, len(keys) * self.num_output_neurons)
                    or not features_out.is_contiguous() or features_out.device != spikes.device):
                raise ValueError(f"features_out must be a contiguous float32 ({B}, {len(keys) * self.num_output_neurons}) "
                                 f"tensor on {spikes.device}")
            feats = features_out
        else:
            feats = torch.empty((B, len(keys) * self.num_output_neurons), dtype=torch.float32,
                                device=self.device)
        sm = (torch.empty((B, T, self.num_neurons), dtype=torch.uint8, device=self.device)
              if want_spike_matrix else None)
        vt = (torch.empty((B, T, self.num_neurons), dtype=torch.float32, device=self.device)
              if want_v_trace else None)
        if stats_out is not None and (stats_out.dtype != torch.int32 or tuple(stats_out.shape) != (B, 2)
                                      or not stats_out.is_contiguous() or stats_out.device != spikes.device):
            raise ValueError(f"stats_out must be a contiguous int32 ({B}, 2) tensor on {spikes.device}")
        if longest_first is None:
            longest_first = self.longest_first_default(B)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            if longest_first:
                # scratch of this call alone: the caching allocator hands a block back to the stream it was taken on
                need = self.lib.lsm_reservoir_order_workspace(B)
                ws = torch.empty((need + 3) // 4, dtype=torch.int32, device=self.device)
                _lib.check(self.lib.lsm_reservoir_run_ordered(
                    self._handle, _dev(spikes), B, T, _host(key_ids), len(keys), _dev(feats), _dev(sm),
                    _dev(vt), _dev(stats_out), int(waves_per_clip), _dev(ws), need, stream),
                    "lsm_reservoir_run_ordered")
            else:
                _lib.check(self.lib.lsm_reservoir_run(
                    self._handle, _dev(spikes), B, T, _host(key_ids), len(keys), _dev(feats), _dev(sm),
                    _dev(vt), _dev(stats_out), int(waves_per_clip), stream),
                    "lsm_reservoir_run")
        return feats, sm, vt

    def longest_first_default(self, n_clips: int) -> bool:
        """Whether ``run_batch`` starts the clips of a batch of this size longest first by default."""
        return n_clips > self._compute_units and os.environ.get("LSM_RESERVOIR_ORDER", "1") != "0"

    def diagnostics(self, spikes) -> dict:
        """Batched health statistics (the quantities /root/reference/extract_lsm_features.py:119-133
        derives per clip from ``lsm.spike_matrix``): per clip the share of neurons that fired at least
        once, the number of silent neurons and the mean spikes per neuron.  The two integers behind them
        come out of the reservoir kernel itself (one flag per neuron, one count per wave): no (T, N) spike
        matrix is written or reduced."""
        B = int(spikes.shape[0])
        stats = torch.empty((B, 2), dtype=torch.int32, device=self.device)
        self.run_batch(spikes, ['spike_counts'], stats_out=stats)
        st = stats.cpu().numpy().astype(np.int64)                        # exact integers from here on
        active, total = st[:, 0], st[:, 1]
        return {
            "participation": active / self.num_neurons * 100,           # same expression as the reference
            "dead_neurons": self.num_neurons - active,
            "mean_spikes_per_neuron": total / self.num_neurons,
        }

    # 'band': round-1 name of 'ring'; 'ring-contiguous': ring rows with contiguous quad ownership only (tests)
    # 'ring-pairs' / 'ring-quads': ring rows shared out in 128-neuron pair blocks (csrc/lif_pair.h) / in 256-neuron quads
    # (csrc/lif_ring.h) only -- 'ring' takes pair blocks where the reservoir has them (tests, same-box A/B runs)
    KERNEL_MODES = {"auto": 0, "sparse": 1, "dense": 2, "ring": 3, "band": 3, "ring-contiguous": 4, "ring-pairs": 5,
                    "ring-quads": 6}

    def set_kernel(self, mode: str = "auto"):
        """'auto' (register accumulation over dense presynaptic rows; over ring rows -- dense ring window
        plus a list of the rewired synapses -- for ring-like reservoirs whose dense table exceeds the L2
        caches), 'sparse' (CSC scatter through LDS), 'dense' or 'ring'.  'dense' on a reservoir that 'auto' serves
        with ring rows builds the dense table now (one allocation + synchronisation on this device)."""
        with torch.cuda.device(self.device):
            return self._set_kernel(mode)

    def _set_kernel(self, mode: str = "auto"):
        _lib.check(self.lib.lsm_reservoir_set_kernel(self._handle, self.KERNEL_MODES[mode]),
                   "lsm_reservoir_set_kernel")

    def kernel_in_use(self) -> str:
        return {1: "sparse", 2: "dense", 3: "ring"}[self.lib.lsm_reservoir_kernel_in_use(self._handle)]

    def plan(self, n_clips: int, n_steps: int, waves_per_clip: int = 0) -> dict:
        """What `run_batch` would launch for this batch: kernel, layout, LDS bytes per clip and the bytes of the
        weight table that kernel gathers from (`lsm_reservoir_plan`), plus the bytes one spike requests from it, mean
        over the presynaptic neurons (`lsm_reservoir_row_request_bytes`)."""
        k, wpc, sl, lds, tab = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_long()
        _lib.check(self.lib.lsm_reservoir_plan(self._handle, n_clips, n_steps, waves_per_clip, C.byref(k),
                                               C.byref(wpc), C.byref(sl), C.byref(lds), C.byref(tab)),
                   "lsm_reservoir_plan")
        rb = C.c_double()
        _lib.check(self.lib.lsm_reservoir_row_request_bytes(self._handle, n_clips, n_steps, waves_per_clip,
                                                            C.byref(rb)), "lsm_reservoir_row_request_bytes")
        return {"kernel": {1: "sparse", 2: "dense", 3: "ring"}[k.value], "waves_per_clip": wpc.value,
                "slots_per_lane": sl.value, "lds_bytes": lds.value, "table_bytes": tab.value,
                "row_request_bytes": rb.value,
                "input_mode": self._input_mode(n_clips, n_steps, waves_per_clip)}

    def _input_mode(self, n_clips: int, n_steps: int, waves_per_clip: int) -> int:
        mode = int(self.lib.lsm_reservoir_input_mode(self._handle, n_clips, n_steps, waves_per_clip))
        if mode < 0:                         # an LSM_ERR_* code is not a mode
            _lib.check(mode, "lsm_reservoir_input_mode")
        return mode

    def layout(self, n_clips: int, n_steps: int, waves_per_clip: int = 0):
        wpc, sl, lds = C.c_int(), C.c_int(), C.c_int()
        _lib.check(self.lib.lsm_reservoir_layout(self._handle, n_clips, n_steps, waves_per_clip,
                                                 C.byref(wpc), C.byref(sl), C.byref(lds)))
        return {"waves_per_clip": wpc.value, "slots_per_lane": sl.value, "lds_bytes": lds.value}

    # ---- the reference's single-clip protocol -------------------------------------------
    def reset(self):
        self._input = None
        self.spike_matrix = None
        self._features = None

    def set_input_spike_times(self, sample):
        self._input = np.ascontiguousarray(sample, dtype=np.uint8)

    def simulate(self):
        if self._input is None:
            raise RuntimeError("simulate() before set_input_spike_times()")
        feats, sm, _ = self.run_batch(self._input[None], FEATURE_KEYS, want_spike_matrix=True)
        self.spike_matrix = sm[0].cpu().numpy()
        self._features = feats[0].cpu().numpy().reshape(len(FEATURE_KEYS), self.num_output_neurons)

    def extract_features_from_spikes(self) -> dict:
        """dict[str -> (N_out,) float32]; entries that are undefined for a silent neuron are NaN
        here (the batched path returns them as 0, which is what the caller's nan_to_num makes)."""
        if self._features is None:
            raise RuntimeError("extract_features_from_spikes() before simulate()")
        f = {k: self._features[i].copy() for i, k in enumerate(FEATURE_KEYS)}
        n = f['spike_counts']
        for k in ('mean_spike_times', 'first_spike_times', 'last_spike_times'):
            f[k][n < 1] = np.nan
        for k in ('mean_isi', 'isi_variances'):
            f[k][n < 2] = np.nan
        return f
