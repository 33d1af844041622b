"""The overlapped hot path as a library component: audio batches -> reservoir feature rows.

The reference chains its stages through files and serial Python loops
(/root/reference/main.py:19-27; clip loops at create_dataset.py:143 and extract_lsm_features.py:78).
Here one STEP = one batch of clips through filterbank -> dB/normalise/resize -> hysteresis encoder ->
LIF reservoir -> features, all on the GPU -- two launches since round 3, the fused front end
(`lsm_gammatone_spikes_f64`) and the reservoir -- and consecutive steps are issued on rotating HIP streams: the
work of a step is ordered on its own stream, the float64 filterbank of step s+1 runs beside the
latency-bound reservoir kernel of step s and refills the CUs that clips finishing early leave idle.
Every step still does all its work; per-step latency grows, throughput rises.

`HotPath` owns what used to live in bench.py: the stream rotation, the layout hint for the reservoir
kernel (a launch that shares the chip with other kernels prefers fewer, fatter waves than a lone one:
`waves_per_clip=-1`, decided inside liblsm_hip.so) and the hardware-queue count.
"""
from __future__ import annotations

import os

import numpy as np
import torch

DEFAULT_STREAMS = 6
# front-end streams of their own (0 = rotation: every step keeps to one stream).  Measured at 128 filters / 1000
# neurons / 256 clips (profiles/r03_stream_topology.txt, r03_priorities_and_stream_counts.txt): 5 + 6 streams 0.65 ms
# per step over 200 steps and 0.80-0.81 over the driver's 20, 4 + 6 streams 0.66 / 0.82-0.83, the rotation over 6
# streams 0.70 / 0.86-0.90 on the same box -- a rotation stream cannot issue its next front end before its own
# reservoir kernel has finished; four one-launch front ends (64 workgroups each, one per CU) cover the chip and the
# fifth is already queued when the first CUs come free.  5 + 6 + the default stream = the 12 hardware queues.
DEFAULT_FE_STREAMS = int(os.environ.get("LSM_FE_STREAMS", "5"))
DEFAULT_HW_QUEUES = 12
RASTER_DEPTH = 4                     # raster buffers per front-end stream (two-stage topology)
# the last batches of a finite list take the low-latency layouts (HotPath.submit(tail=True)); measured at cfg2,
# 20-step bursts (profiles/r04_tail_steps.txt)
TAIL_STEPS = int(os.environ.get("LSM_TAIL_STEPS", "2"))
STAGES = ("full", "frontend", "reservoir")


def configure_hardware_queues(n: int = DEFAULT_HW_QUEUES) -> int:
    """The HIP runtime multiplexes streams onto 4 hardware queues by default, and kernels of streams that
    share a queue serialise: with 4 queues three streams are the optimum, with 8-12 six streams overlap
    (0.96 -> 0.72 ms per step; 12 also leave RCCL's stream a queue of its own).  The variable is read when
    HIP initialises, so it must be set BEFORE the first CUDA/HIP call of the process: importing the package
    does that (`lsm_speech_classifier_amd.__init__`, an existing GPU_MAX_HW_QUEUES wins; a warning when HIP
    was already up).  Returns the number of queues IN FORCE for this process, which is what `HotPath.hw_queues`
    and the bench line report -- 4 when the package was imported too late, whatever the variable says now."""
    from . import EFFECTIVE_HW_QUEUES
    return int(EFFECTIVE_HW_QUEUES)


class HotPath:
    """fe: frontend.SpikeFrontEnd, net: snn.SNN (same device).  `streams` = reservoir streams (1 = serial:
    everything on the current stream); `fe_streams` = front-end streams of their own (0 = rotation: a step keeps to
    one of the `streams`; None = DEFAULT_FE_STREAMS)."""

    def __init__(self, fe, net, feature_keys=None, streams: int = DEFAULT_STREAMS,
                 waves_per_clip: int | None = None, time_reservoir: bool = False,
                 fe_streams: int | None = None):
        def norm(d):      # a tensor's device always carries its index, `torch.device("cuda")` does not
            return d if d.index is not None or d.type != "cuda" else torch.device("cuda", torch.cuda.current_device())
        if norm(fe.device) != norm(net.device):
            raise ValueError(f"front end on {fe.device}, reservoir on {net.device}")
        if fe.n_channels != net.n_channels:
            raise ValueError(f"front end has {fe.n_channels} channels, reservoir expects {net.n_channels}")
        self.fe, self.net = fe, net
        self.device = norm(fe.device)     # so that a device-resident batch is recognised as such (not copied again)
        self.feature_keys = feature_keys
        self.n_streams = max(1, int(streams))
        # inside the rotation the reservoir launch shares the chip: let the library pick for that case
        self.waves_per_clip = (-1 if self.n_streams > 1 else 0) if waves_per_clip is None else int(waves_per_clip)
        self.time_reservoir = bool(time_reservoir)
        self.hw_queues = configure_hardware_queues()
        # Stream topology.  fe_streams == 0 ("rotation"): step s runs front end AND reservoir on stream s % streams, so
        # the stream cannot issue its next front end before this step's reservoir kernel has finished.
        # fe_streams > 0 ("two stages"): the front ends go round `fe_streams` streams of their own and never wait
        # for a reservoir kernel; step s's reservoir launch goes to reservoir stream s % streams behind an event.
        self.n_fe_streams = (DEFAULT_FE_STREAMS if fe_streams is None else max(0, int(fe_streams))) if self.n_streams > 1 else 0
        with torch.cuda.device(self.device):
            self.streams = ([torch.cuda.Stream(device=self.device) for _ in range(self.n_streams)]
                            if self.n_streams > 1 else [None])
            self.fe_streams = [torch.cuda.Stream(device=self.device) for _ in range(self.n_fe_streams)]
        # A front end that meets an idle GPU (none of this pipeline's front ends still in flight: the first step
        # after a synchronisation, or a caller that submits rarely) is launched in the low-latency layout: it is done
        # after 1.47 instead of 2.43 ms, and -- what matters for a short burst of steps -- the burst's front ends no
        # longer run in lockstep rounds of four that all end, and all release their reservoir launches, together.
        self.wide_when_idle = os.environ.get("LSM_FE_WIDE_WHEN_IDLE", "1") != "0"
        self.tail_lone_layout = os.environ.get("LSM_TAIL_LONE_LAYOUT", "1") != "0"   # diagnostic: tail steps' reservoir layout
        self._wide_below = int(os.environ.get("LSM_FE_WIDE_BELOW", "1"))     # diagnostic: front ends in flight below which the wide layout goes out
        self._fe_done = []                  # events of the front ends issued, newest last (two-stage topology)
        # Back-pressure: the host enqueues a step in ~0.1 ms, the GPU runs it in ~0.64: without a bound a long run is
        # enqueued thousands of steps ahead and everything a step allocates stays reserved until the GPU gets there
        # (measured before the pipeline owned its raster buffers: 12 GB after 3000 steps).  submit() therefore waits
        # for step s - max_ahead before it issues step s; the GPU keeps max_ahead steps queued, far more than it overlaps.
        self.max_ahead = max(16, 4 * (self.n_streams + self.n_fe_streams))
        # Two-stage topology: the rasters and the front end's scratch live in buffers the pipeline owns, RASTER_DEPTH
        # rasters and one scratch per front-end stream, handed round behind events -- a raster buffer is written again
        # only after the reservoir launch that read it has finished (a GPU-side wait of the front-end stream, never
        # a host wait).  Allocating them per step from torch's allocator needs the block of step s to stay reserved
        # until its reservoir launch has finished on ANOTHER stream (`record_stream`): memory then grows with the
        # steps enqueued ahead and a burst allocates inside the timed path.  Throughput is the same either way
        # (profiles/r03_pipeline_buffer_pool.txt); the pool makes the steady path free of allocator calls.
        self.pool = os.environ.get("LSM_HOTPATH_POOL", "1") != "0"
        self._rasters = {}                  # (front-end slot, batch size) -> [buffers], [events "free again"], next index
        self._ws = {}                       # (front-end slot, batch size) -> scratch
        self._in_flight = []                # one event per submitted step, on the stream its last launch went to
        self._step = 0
        self.reservoir_events = []          # (start, end) HIP event pairs, one per submitted step (time_reservoir)
        self._h2d = {}

    # ---- one step ---------------------------------------------------------------------------
    def _one(self, x, stats_out, out, stage, tail=False):
        if stage == "frontend":
            return self.fe.encode(x)
        rasters = x if stage == "reservoir" else self.fe.encode(x)
        if self.time_reservoir:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        # a tail step's reservoir launch meets a draining GPU: the layout of a lone launch (more, thinner waves)
        wpc = 0 if (tail and self.waves_per_clip == -1 and self.tail_lone_layout) else self.waves_per_clip
        feats, _, _ = self.net.run_batch(rasters, self.feature_keys, waves_per_clip=wpc,
                                         stats_out=stats_out, features_out=out)
        if self.time_reservoir:
            e1.record()
            self.reservoir_events.append((e0, e1))
        return feats

    def submit(self, audio, stats_out=None, after=None, out=None, stage: str = "full", tail: bool = False):
        """Issue one step on the next stream of the rotation.  `audio`: (B, n_samples) float32 -- device tensor,
        or pinned/pageable host tensor / NumPy array, uploaded on the step's own stream.  Returns (features
        (B, n_keys*N_out) device tensor, stream it is produced on); the caller waits (`stream.synchronize()`,
        `HotPath.synchronize()`, or an event) before reading the features.

        Ordering of a DEVICE input: the step's stream waits for `after` when that event is given, otherwise for
        everything the current stream of the device has issued so far (so a batch produced there just before
        `submit()` is never read early); host inputs are copied on the step's stream itself.
        `out`: write the rows into this (B, n_feat) float32 tensor (a slice of a gather buffer) instead of a new one.
        `stage`: "full"; "frontend" (returns the uint8 rasters; `out`/`stats_out` unused); "reservoir" (`audio` IS a
        uint8 raster batch (B, C, T) on the device) -- the same rotation, launches and layout hint as a full step,
        so that per-stage timings go through the code the headline goes through.
        `tail`: the caller has (almost) nothing to submit behind this step -- one of the last batches of a finite list
        (`run()` sets it for the last `TAIL_STEPS` batches).  The step then takes the low-latency layouts: the front end
        with one chain per lane (twice the waves, each half as long), the reservoir launch as a lone launch would be laid
        out.  At the end of a burst the last front end otherwise runs alone on a quarter of the chip for ~2 ms while the
        rest drains (profiles/r04_tail_steps.txt); inside a long run the normal layouts use fewer CU-cycles.
        Back-pressure: when `max_ahead` earlier steps have not finished, submit() first waits for the oldest."""
        res, st = self._submit(audio, stats_out, after, out, stage, tail)
        if st is not None and self.n_streams > 1:
            ev = torch.cuda.Event()
            ev.record(st)
            self._in_flight.append(ev)
        return res, st

    def _submit(self, audio, stats_out=None, after=None, out=None, stage: str = "full", tail: bool = False):
        """submit() without the in-flight bookkeeping."""
        if stage not in STAGES:
            raise ValueError(f"stage must be one of {STAGES}, got {stage!r}")
        slot = self._step % self.n_streams
        st = self.streams[slot]
        self._step += 1
        while len(self._in_flight) >= self.max_ahead:
            self._in_flight.pop(0).synchronize()
        cur = torch.cuda.current_stream(self.device)
        on_device = torch.is_tensor(audio) and audio.device == self.device
        if st is None:
            if after is not None:
                cur.wait_event(after)
            with torch.cuda.device(self.device):
                x = audio if stage == "reservoir" else self._to_device(audio, slot)
                return self._one(x, stats_out, out, stage, tail), cur
        if self.n_fe_streams and stage != "reservoir":
            # two stages: front end on its own stream, the reservoir launch behind an event on another
            fslot = (self._step - 1) % self.n_fe_streams
            fst = self.fe_streams[fslot]
            with torch.cuda.device(self.device):
                with torch.cuda.stream(fst):
                    if after is not None:
                        fst.wait_event(after)
                    elif on_device:
                        fst.wait_stream(cur)
                    self._fe_done = [ev for ev in self._fe_done[-8:] if not ev.query()]
                    # the front end's own decision (mel branch, > 1024 filters and LSM_FRONTEND_SPLIT=1 take the split
                    # route, which allocates its raster on the front-end stream): only the fused launch writes
                    # into the pipeline's rings, everything else goes through record_stream below
                    fuses = bool(getattr(self.fe, "will_fuse", lambda: False)())
                    idle = fuses and ((self.wide_when_idle and len(self._fe_done) < self._wide_below) or tail)
                    x = self._to_device(audio, ("fe", fslot))
                    pooled = self.pool and stage == "full" and fuses
                    if pooled:
                        key = (fslot, int(x.shape[0]))
                        ring = self._rasters.get(key)
                        if ring is None:
                            ring = self._rasters[key] = [
                                [torch.empty((x.shape[0], self.fe.n_channels, self.fe.n_steps), dtype=torch.uint8,
                                             device=self.device) for _ in range(RASTER_DEPTH)],
                                [None] * RASTER_DEPTH, 0]
                            self._ws[key] = self.fe.new_workspace(x.shape[0])        # one per front-end stream
                        bi = ring[2]
                        ring[2] = (bi + 1) % RASTER_DEPTH
                        if ring[1][bi] is not None:
                            fst.wait_event(ring[1][bi])          # its last reader (a reservoir launch) has finished
                        gt = self.fe.filterbank == "gammatone"
                        rasters = self.fe.encode(x, low_latency=idle and gt, raster_out=ring[0][bi], workspace=self._ws[key],
                                                 share_lds=gt and self._share_lds(int(x.shape[0])))
                    else:
                        lowlat = idle and getattr(self.fe, "filterbank", "") == "gammatone"
                        rasters = self.fe.encode(x, low_latency=True) if lowlat else self.fe.encode(x)
                        pooled = False
                    done = torch.cuda.Event()
                    done.record(fst)
                    self._fe_done.append(done)
                    if stage == "frontend":
                        return rasters, fst
                if not pooled:
                    rasters.record_stream(st)      # allocated on the front-end stream, read on the reservoir stream
                with torch.cuda.stream(st):
                    st.wait_event(done)
                    feats = self._one(rasters, stats_out, out, "reservoir", tail)
                    if pooled:
                        free = torch.cuda.Event()
                        free.record(st)
                        ring[1][bi] = free
                    return feats, st
        with torch.cuda.device(self.device), torch.cuda.stream(st):
            if after is not None:
                st.wait_event(after)
            elif on_device:
                st.wait_stream(cur)
            x = audio if stage == "reservoir" else self._to_device(audio, slot)
            return self._one(x, stats_out, out, stage, tail), st

    def _share_lds(self, n_clips: int) -> bool:
        """Whether the fused front end gives up its LDS reservation (`encode(share_lds=True)`).  Measured and NOT taken by
        default: beside the ring-row kernel at 4000 neurons (64 KB per clip, two per CU; the front end needs 27 KB per
        workgroup since round 4) the whole path runs 6.37-6.40 ms per step without the reservation against 6.26-6.28 with
        it -- without it the dispatcher stacks several front-end workgroups on one CU
        (profiles/r04_cfg4_frontend_lds_placement.txt).  LSM_FE_SHARE_LDS=1 switches it on for A/B runs."""
        return os.environ.get("LSM_FE_SHARE_LDS") == "1"

    def _to_device(self, audio, slot):
        if isinstance(audio, np.ndarray):
            audio = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32))
        if audio.device == self.device:
            return audio
        # host batch: asynchronous copy on the step's stream into this rotation slot's own buffer
        buf = self._h2d.get((slot, tuple(audio.shape)))
        if buf is None:
            buf = self._h2d[(slot, tuple(audio.shape))] = torch.empty(audio.shape, dtype=torch.float32,
                                                                      device=self.device)
        buf.copy_(audio, non_blocking=True)
        return buf

    def prime(self, audio, stage: str = "full", min_ms: float = 0.0):
        """One untimed step on EVERY stream of the pipeline, then a synchronisation: each stream's first use
        pays for its allocator pool (torch caches device memory per stream: the first `encode` on a stream
        calls hipMalloc, which stalls every queue), for the pipeline's own buffers, for the first launch on its
        hardware queue and for the kernels' one-off attribute calls.  Part of set-up; afterwards a step makes no
        runtime call but launches and (host inputs) one asynchronous copy.

        `min_ms` > 0: keep submitting untimed steps until that much time has passed -- the GPU's clock governor
        needs tens of milliseconds of load to reach the clock the pipeline then holds; a burst of steps that
        starts right after a set-up phase (seconds of host work, an idle GPU) otherwise runs its first ~15 ms
        6-9 % slower than the same steps in the middle of a long run (profiles/r03_clock_ramp.txt)."""
        import time
        self.fork_from_current()
        events = self.reservoir_events
        self.reservoir_events = []
        t0 = time.perf_counter()
        n = 0
        while n < max(self.n_streams, self.n_fe_streams) or (time.perf_counter() - t0) * 1e3 < min_ms:
            self.submit(audio, stage=stage)
            n += 1
            if n % 8 == 0:                       # let the GPU catch up: the loop is timed by work done, not enqueued
                self._in_flight and self._in_flight[-1].synchronize()
        self.synchronize()
        self.reservoir_events = events
        return n

    def fork_from_current(self):
        """Make every stream of the rotation wait for what the current stream has issued so far
        (inputs produced there, e.g. an upload or a reservoir build)."""
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams + self.fe_streams:
            if st is not None:
                st.wait_stream(cur)

    def join_to_current(self):
        """The reverse edge: the current stream waits for everything the rotation has issued (no host
        synchronisation) -- e.g. before ONE collective over the rows of many steps."""
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams + self.fe_streams:
            if st is not None:
                cur.wait_stream(st)

    def synchronize(self):
        self._in_flight.clear()
        for st in self.streams + self.fe_streams:
            if st is not None:
                st.synchronize()
        torch.cuda.current_stream(self.device).synchronize()

    # ---- many steps -------------------------------------------------------------------------
    def run(self, audio_batches, out: torch.Tensor | None = None) -> torch.Tensor:
        """Feature rows of every batch, in order, as one (n_clips, n_feat) device tensor.  `audio_batches`
        is an iterable of (B_i, n_samples) arrays/tensors (host or device)."""
        self.fork_from_current()
        # The last TAIL_STEPS batches are submitted in the low-latency layouts: that needs a look-ahead of TAIL_STEPS
        # batches, not the whole list -- a lazily produced stream of (pinned or device) batches is consumed as it comes and
        # never held in full.
        from collections import deque
        parts, ahead, it = [], deque(), iter(audio_batches)
        for a in it:
            ahead.append(a)
            if len(ahead) > TAIL_STEPS:
                parts.append(self.submit(ahead.popleft(), tail=False)[0])
        while ahead:
            parts.append(self.submit(ahead.popleft(), tail=True)[0])
        self.synchronize()
        if not parts:
            n_keys = len(self.feature_keys) if self.feature_keys is not None else 8
            return torch.empty((0, n_keys * self.net.num_output_neurons), dtype=torch.float32, device=self.device)
        if out is None:
            return torch.cat(parts)
        torch.cat(parts, out=out)
        return out


def features_from_audio(audio: np.ndarray, fe, net, feature_keys, batch: int = 1024,
                        streams: int = DEFAULT_STREAMS, device_out: bool = False):
    """Host convenience for the drop-in scripts' in-memory path: (n, n_samples) float32 on the host ->
    (n, n_feat) float32 on the host (or, `device_out`, still on the GPU for a gather), batches of `batch` clips
    through the overlapped pipeline (pinned staging, uploads on the steps' streams).  n = 0 (an empty shard)
    gives an empty (0, n_feat) block."""
    hp = HotPath(fe, net, feature_keys, streams=streams)
    pinned = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32))
    if len(pinned):
        try:
            pinned = pinned.pin_memory()
        except RuntimeError:
            pass
    with torch.cuda.device(hp.device):
        feats = hp.run(pinned[lo:lo + batch] for lo in range(0, len(pinned), batch))
    return feats if device_out else feats.cpu().numpy()
