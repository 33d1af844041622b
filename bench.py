#!/usr/bin/env python3
"""Hot-path benchmark (driver contract: python bench.py --gpus N --steps K --warmup W).

One STEP = one pass of the whole hot path over one batch of synthetic clips per GPU, inputs
already resident in HBM: audio (B, 16000) f32 -> gammatone filterbank -> dB/normalise/resize ->
hysteresis encoder -> uint8 raster (B, C, 400) -> LIF reservoir (400 steps) -> features
(B, 5*N_out) f32 [-> RCCL all-gather of the feature rows when N > 1].  Default workload is
BASELINE.json configs[1]: 128 gammatone filters, 1000-neuron reservoir, batch 256 per GPU,
`original` feature set, multiplier 0.6.  Clips shard across ranks (weak scaling, no data-path
collective except the final feature gather the reference's single-process run implies).

This file only TIMES the path: the overlapped pipeline itself (stream rotation, hardware-queue count,
reservoir layout for a shared GPU) is `lsm_speech_classifier_amd.pipeline.HotPath`, the same object the
drop-in scripts use.  With --gpus N > 1 and no launcher environment it starts N fresh rank processes
itself (before anything touches the GPU) and exits with their status.

Prints ONE JSON line on rank 0 with the contract keys plus `roofline` (reservoir kernel, algorithmic
bytes of SURVEY.md §8(d) / measured HIP-event time; `dominant_kernel_by_time` = the float64 filterbank)
and, at N=1, `cpu_baseline` (the C oracle timed on the host cores on a bounded sample of the workload).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (n_filters, filterbank, N, k, N_out, batch per GPU, audio kind)
    "cfg1": dict(n_filters=40, filterbank="mel", N=500, k=100, n_out=200, batch=200,
                 audio="speech_like", desc="BASELINE configs[0], the reference's CPU-runnable case: 40 mel "
                 "filters, 500-neuron reservoir, batches of 200 clips"),
    "cfg2": dict(n_filters=128, filterbank="gammatone", N=1000, k=200, n_out=400, batch=256,
                 audio="speech_like", desc="12-class-like synthetic speech, 128 gammatone filters, "
                 "1000-neuron reservoir, batch=256 per GPU"),
    "cfg4": dict(n_filters=128, filterbank="gammatone", N=4000, k=800, n_out=1600, batch=1024,
                 audio="speech_like", desc="128 filters, 4000-neuron reservoir, batch=1024 per GPU"),
    "cfg5": dict(n_filters=256, filterbank="gammatone", N=8000, k=1600, n_out=3200, batch=4096,
                 audio="white_noise", desc="white-noise clips, 256 filters, 8000-neuron reservoir, "
                 "batch=4096 per GPU (HBM roofline stress)"),
}
FEATURE_SET = ['spike_counts', 'spike_variances', 'mean_spike_times', 'mean_isi', 'isi_variances']
MULTIPLIER = 0.6
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec


def gather_ceiling_gbs(table_bytes):
    """What the memory system serves a random-row gather from a table of this size, chip-wide (MI355X_MICROARCH.md,
    'Indexed rows: gather into LDS', measured): rows every workgroup shares inside each XCD's 4 MiB L2 16.8-18.8 TB/s;
    a 38 MB table (Infinity Cache) 8.6 TB/s; 151 MB 7.4-7.9 TB/s; beyond the 256 MiB Infinity Cache (HBM) 6.0-6.1 TB/s.
    The lower figure of each range is used."""
    if table_bytes <= 4.5 * 2 ** 20:
        return 16800.0, "table within one XCD's L2 (4 MiB): 16.8 TB/s"
    if table_bytes <= 40e6:
        return 8600.0, "table <= 38 MB, served by the Infinity Cache: 8.6 TB/s"
    if table_bytes <= 256 * 2 ** 20:
        return 7400.0, "table <= 256 MiB, served by the Infinity Cache: 7.4 TB/s (151 MB measurement)"
    return 6000.0, "table beyond the Infinity Cache, served by HBM: 6.0 TB/s"
F64_UNFUSED_PEAK_TOPS = 39.3   # one float64 operation per lane and instruction, all 1024 SIMDs (exp/ubench_f64.hip)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU (0 = the config's)")
    ap.add_argument("--waves-per-clip", type=int, default=None,
                    help="reservoir layout (default: the library chooses, knowing whether steps overlap)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "dense", "ring", "ring-pairs", "ring-quads", "sparse"])
    ap.add_argument("--stage", default="full", choices=["full", "reservoir", "frontend"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--from-host", action="store_true",
                    help="every step first copies its audio batch from pinned host memory (PCIe-inclusive "
                         "rate; never the contract's `value`, which is quoted on HBM-resident inputs)")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams the steps rotate over (0 = the pipeline's default; 1 = serial)")
    ap.add_argument("--fe-streams", type=int, default=None,
                    help="front-end streams of their own (0 = rotation: a step keeps to one stream; default: the pipeline's)")
    ap.add_argument("--prime-ms", type=float, default=40.0,
                    help="set-up, before the W warm-up steps: HotPath.prime() keeps the pipeline running untimed for this "
                         "long so that the timed steps see the clock a running pipeline holds, not the ramp after an idle "
                         "set-up phase (0 = one step per stream only)")
    ap.add_argument("--exchange", default="once", choices=["chunked", "once", "per-step"],
                    help="N > 1: 'once' (default) = ONE all-gather after the last step, inside the timed region -- what the "
                         "product does per split (extract_lsm_features.py) and the only mode whose RCCL call pattern is "
                         "a single collective on the current stream; 'chunked' = the rows of every --exchange-chunk steps "
                         "travel in one all-gather on a stream of its own as soon as those steps have finished, so only "
                         "the last chunk's gather is exposed behind the last step (rehearsed with ONE rank over RCCL and "
                         "2-3 ranks over gloo only: opt-in until a run with >= 2 GPUs is on record, ADVICE r4); "
                         "'per-step' = an all-gather behind every step's reservoir kernel")
    ap.add_argument("--exchange-chunk", type=int, default=5, help="steps per all-gather of the chunked exchange")
    ap.add_argument("--tail-steps", type=int, default=None,
                    help="the last this-many timed steps are submitted with HotPath.submit(tail=True), as HotPath.run() does "
                         "for the last batches of a finite list (default: the pipeline's TAIL_STEPS; 0 = none)")
    ap.add_argument("--no-unprimed", action="store_true",
                    help="skip the first, unprimed pass (one step per stream, W warm-up steps, K timed steps: the "
                         "round-2 protocol) whose figure the line reports as `unprimed` beside the headline")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------ launcher ----
def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without torchrun: start N fresh rank processes of this script (the parent
    has not touched the GPU and never does), one per GPU, and return the worst exit status.  Rank 0 prints
    the JSON line on the inherited stdout.  The children are POLLED: when one exits non-zero while others still
    sit in the rendezvous or a collective, the rest are terminated (then killed) and that status is returned at
    once instead of after the process-group timeout."""
    import tempfile
    sock = socket.socket()                          # a free rendezvous port, held until the children are started
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    procs, errs = [], []

    def replay(r, last=None):
        """Rank r's stderr (its last `last` lines, or all of it) on this process's stderr."""
        errs[r].flush()
        errs[r].seek(0)
        lines = errs[r].read().decode("utf-8", "replace").splitlines()
        for l in (lines[-last:] if last else lines):
            print(f"[rank {r}] {l}", file=sys.stderr)

    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL between processes needs on this driver
            # every rank's stderr goes to a file of its own: when a rank dies, ITS last lines are what the parent shows
            # (interleaved output of eight ranks hides which one failed and why)
            errs.append(tempfile.TemporaryFile())
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stderr=errs[r]))
        sock.close()          # rank 0 binds it seconds later (interpreter start + import torch); nobody else has the number
        rc = 0
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc = max(abs(c) for _, c in bad)
                for r, c in bad:
                    print(f"bench.py: rank {r} of {n} exited with status {c}; its last stderr lines:", file=sys.stderr)
                    replay(r, last=40)
                break
            if all(c == 0 for c in codes):
                for r in range(n):                  # nothing a rank said is lost: rank 0 in full, the others' last lines
                    replay(r, last=None if r == 0 else 6)
                return 0
            time.sleep(0.05)
    finally:
        sock.close()
        live = [p for p in procs if p.poll() is None]
        if live:                                    # only on the failure path: these are this call's own children
            for p in live:
                p.terminate()
            deadline = time.time() + 5.0
            for p in live:
                try:
                    p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
    return rc


def spawn_check():
    """LSM_BENCH_SPAWN_ONLY=1: the launcher path without a GPU -- every rank joins a gloo group, the ranks
    are summed, rank 0 reports (tests/test_bench_spawn.py)."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    die = os.environ.get("LSM_BENCH_SPAWN_DIE")     # "<rank>:<code>": that rank exits before the rendezvous
    if die and int(die.split(":")[0]) == rank:
        sys.exit(int(die.split(":")[1]))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank)])
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"spawn_check": True, "world": world, "rank_sum": float(t.item()),
                          "local_ranks_distinct": True}), flush=True)
    dist.destroy_process_group()


# ------------------------------------------------------------------------------ workload ----
def w_critico(k, theta, refractory, rasters_dev):
    """extract_lsm_features.py:33-60 on the first <=500 clips (device reduction, host scalar)."""
    import torch
    sub = rasters_dev[:500]
    if sub.numel() == 0 or k == 0:
        return 0.007
    avg_i = float(sub.sum(dtype=torch.int64)) / sub.numel()
    return (theta - 2 * avg_i * refractory) / (k / 2)


def make_audio(kind, n, seed):
    import numpy as np
    from lsm_speech_classifier_amd import synth
    if kind == "white_noise":
        return synth.white_noise(n, seed=seed)
    return synth.class_chirps(np.arange(n) % 12, seed=seed)


def cpu_baseline(cfg, audio, res, seconds_budget=24.0):
    """C oracle (oracle/, kind 'port') on a bounded sample of the same workload: (1) one clip at a time on
    one core, like the reference's serial loops (create_dataset.py:143, extract_lsm_features.py:78);
    (2) the same path on every host core, MEASURED: front end and reservoir each with one clip per OpenMP
    thread."""
    import numpy as np
    from lsm_speech_classifier_amd import frontend
    from oracle import cport, ref_numpy
    cport.build()
    thr, gap = frontend.SPIKE_THRESHOLDS, frontend.HYSTERESIS_GAP
    mel = cfg["filterbank"] == "mel"
    coefs = None if mel else ref_numpy.gammatone_coefs(16000, cfg["n_filters"], 50)

    def front(a):
        if mel:      # NumPy restatement of the librosa defaults (oracle/ref_numpy.py), float32 like the reference
            return ref_numpy.encode_hysteresis(ref_numpy.normalise_resize(ref_numpy.mel_db(a, cfg["n_filters"])),
                                               thr, gap)
        return cport.encode_hysteresis(cport.normalise_resize(cport.gammatone_db(
            cport.gammatone_spec(a, coefs, 400, 160, 98))), thr, gap)

    def one(a):
        return cport.lif_run(res, front(a), FEATURE_SET, want_spikes=False)[0]

    one(audio[0])                                   # warm-up
    t0 = time.perf_counter()
    one(audio[1 % len(audio)])
    per_clip = time.perf_counter() - t0
    n = int(max(2, min(len(audio), seconds_budget / 3 / max(per_clip, 1e-4))))
    t0 = time.perf_counter()
    for a in audio[:n]:
        one(a)
    t_serial = time.perf_counter() - t0
    if mel:
        return {"value": round(n / t_serial, 3), "unit": "clips/s", "cores": 1, "kind": "port",
                "sample": f"{n} clips of the same workload, NumPy mel front end + C gather-form LIF, one clip at "
                          f"a time on one core"}
    cores = os.cpu_count() or 1
    # all cores: enough clips to give every core work for about a third of the budget
    n_all = int(max(n, min(len(audio), cores * max(1, round(seconds_budget / 3 / max(per_clip, 1e-4))))))
    t0 = time.perf_counter()
    rasters = cport.gammatone_frontend_batch(audio[:n_all], coefs, 400, 160, 98, thr, gap, n_threads=cores)
    t_front_all = time.perf_counter() - t0
    t0 = time.perf_counter()
    cport.lif_run_batch(res, rasters, FEATURE_SET, n_threads=cores)
    t_lif_all = time.perf_counter() - t0
    return {
        "value": round(n / t_serial, 3), "unit": "clips/s", "cores": 1, "kind": "port",
        "sample": f"{n} clips of the same workload, C oracle (gather-form LIF + gammatone), one clip "
                  f"at a time on one core",
        "all_cores": {"value": round(n_all / (t_front_all + t_lif_all), 3), "cores": cores,
                      "sample": f"{n_all} clips, measured: front end {t_front_all:.2f} s + reservoir "
                                f"{t_lif_all:.2f} s, one clip per OpenMP thread in both"},
    }


def run_rank(args):
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("LSM_BENCH_FORCE_DIST") == "1":
        # 5 front-end + 6 reservoir streams + the default stream fill the package's 12 hardware queues; the exchange
        # stream and RCCL's own stream must not share a queue with a pipeline stream (kernels of streams on one queue
        # serialise: profiles/r03_burst_decomposition.txt, 6 + 6 streams on 12 queues).  16 queues run the pipeline
        # as fast as 12 (same file).  Read once, when HIP initialises: set before the package is imported.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import lsm_speech_classifier_amd                # noqa: F401  (sets GPU_MAX_HW_QUEUES before HIP initialises)
    from lsm_speech_classifier_amd import pipeline  # noqa: F401
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # LSM_BENCH_FORCE_DIST=1 takes the distributed code path (process group, broadcast, gather,
    # barrier) even with one rank: a rehearsal of the RCCL calls on a 1-GPU box
    use_dist = world > 1 or os.environ.get("LSM_BENCH_FORCE_DIST") == "1"
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); timing {world}",
              file=sys.stderr)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal switches for a 1-GPU box (never used by the driver): LSM_BENCH_BACKEND=gloo exchanges
        # through the host, LSM_BENCH_SHARE_GPU=1 puts every rank on cuda:0 -- together they run the
        # multi-rank code path (rank > 0, uneven timing, gather order) with several processes on one card
        backend = os.environ.get("LSM_BENCH_BACKEND", "nccl")
        if os.environ.get("LSM_BENCH_SHARE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if use_dist else 0)

    from lsm_speech_classifier_amd import dist as lsm_dist, frontend, reservoir, snn
    from lsm_speech_classifier_amd.pipeline import HotPath, DEFAULT_STREAMS, TAIL_STEPS
    tail_steps = TAIL_STEPS if args.tail_steps is None else max(0, args.tail_steps)
    cfg = CONFIGS[args.config]
    B = args.batch or cfg["batch"]
    fe = frontend.SpikeFrontEnd(cfg["n_filters"], cfg["filterbank"], device=dev)
    audio_np = make_audio(cfg["audio"], B, seed=1234 + 1000 * rank)
    audio = torch.from_numpy(audio_np).to(dev)

    # one-off setup exactly like extract_lsm_features.main: rasters -> w_critico -> reservoir
    rasters0 = fe.encode(audio)
    wc = w_critico(cfg["k"], 2.0, 2, rasters0)
    if use_dist:                                  # every rank must build the SAME reservoir
        t = torch.tensor([wc], dtype=torch.float64, device=dev)
        dist.broadcast(t, 0)
        wc = float(t.item())
    params = reservoir.SimulationParams(num_neurons=cfg["N"], num_output_neurons=cfg["n_out"],
                                        small_world_graph_k=cfg["k"], mean_weight=wc * MULTIPLIER)
    # per-rank set-up, timed: every rank builds the same wiring on its host cores and uploads it -- with eight ranks on one
    # node this is the part of the run outside the timed region that a driver's timeout has to cover (VERDICT r4 #4b)
    t_s0 = time.perf_counter()
    res = reservoir.build_reservoir(params, fe.n_channels)
    t_s1 = time.perf_counter()
    net = snn.SNN(params, reservoir=res, device=dev)
    net.set_kernel(args.kernel)
    torch.cuda.synchronize()
    t_s2 = time.perf_counter()
    setup_s = {"build_reservoir_s": round(t_s1 - t_s0, 3), "lsm_reservoir_create_s": round(t_s2 - t_s1, 3)}
    print(f"bench.py[rank {rank}/{world}]: set-up {args.config}: build_reservoir {setup_s['build_reservoir_s']} s (host), "
          f"lsm_reservoir_create + upload {setup_s['lsm_reservoir_create_s']} s", file=sys.stderr, flush=True)
    n_feat = len(FEATURE_SET) * cfg["n_out"]

    n_streams = args.streams or DEFAULT_STREAMS
    hp = HotPath(fe, net, FEATURE_SET, streams=n_streams, waves_per_clip=args.waves_per_clip,
                 time_reservoir=True, fe_streams=args.fe_streams)
    lay = net.layout(B, fe.n_steps, hp.waves_per_clip)
    audio_pinned = torch.from_numpy(audio_np).pin_memory() if args.from_host else None
    per_step = use_dist and args.exchange == "per-step"
    # 'once' / 'chunked' keep every step's rows of every rank in one block: fall back to the per-step gather when that
    # block would not be small next to the HBM (cfg5 over hundreds of steps), and say so in config.sharding
    gather_block_bytes = world * max(args.steps, 1) * B * n_feat * 4
    if use_dist and not per_step and gather_block_bytes > (16 << 30):
        per_step = True
        if rank == 0:
            print(f"bench.py: one exchange would need a {gather_block_bytes / 2**30:.1f} GiB gather block; "
                  f"gathering behind every step instead", file=sys.stderr)
    once = use_dist and not per_step                 # rows kept per rank, gathered in one or several pieces
    chunk = max(1, min(args.exchange_chunk, max(args.steps, 1))) if args.exchange == "chunked" else max(args.steps, 1)
    xs = torch.cuda.Stream(device=dev) if once and args.exchange == "chunked" else None     # the exchange's own stream
    stage_in = rasters0 if args.stage == "reservoir" else (audio_pinned if args.from_host else audio)
    # per-step exchange: one gather buffer per stream of the rotation (overlapping steps never share an output);
    # one exchange: every step writes its rows into its own slice of a per-rank block, gathered once at the end
    gather_bufs = ([torch.empty((world * B, n_feat), dtype=torch.float32, device=dev)
                    for _ in range(hp.n_streams)] if per_step and args.stage != "frontend" else None)
    n_slots = max(args.steps, args.warmup, 1)
    local_rows = (torch.empty((n_slots, B, n_feat), dtype=torch.float32, device=dev)
                  if once and args.stage != "frontend" else None)
    gathered = (torch.empty((world * args.steps * B, n_feat), dtype=torch.float32, device=dev)
                if local_rows is not None else None)

    collective_host_ms = []             # host time of every chunk's enqueue (waits + collective call): must stay microseconds
    done_events = []                    # one event per step of the current phase (warm-up or timed), behind its last launch

    def gather_chunk(c0, c1):
        """All-gather the rows of steps [c0, c1) of every rank (layout: lsm_speech_classifier_amd.dist.gather_step_chunk)."""
        lsm_dist.gather_step_chunk(gathered, local_rows, c0, c1)

    def gather_behind(c0, c1):
        """The chunk's all-gather on the exchange stream, ordered behind the chunk's steps (GPU-side waits only).
        RCCL orders its own stream behind `xs`; no pipeline stream ever waits for a gather."""
        h0 = time.perf_counter()
        with torch.cuda.stream(xs):
            for ev in done_events[c0:c1]:
                xs.wait_event(ev)
            gather_chunk(c0, c1)
        collective_host_ms.append((time.perf_counter() - h0) * 1e3)

    def step(i, n_exchanged=0, tail=False):
        """One pass of the hot path over the batch on the next stream of the rotation (HotPath.submit, also for
        the --stage variants).  `n_exchanged`: steps of this phase whose rows take part in the exchange."""
        slot = hp._step % hp.n_streams
        out_rows = local_rows[i] if local_rows is not None else None
        feats, st = hp.submit(stage_in, out=out_rows, stage=args.stage, tail=tail)
        if gather_bufs is not None:
            # RCCL's stream is ordered after this step's reservoir kernel and the step's stream after the
            # gather; with >= 12 hardware queues the exchange does not disturb the other steps in flight
            with torch.cuda.stream(st):
                dist.all_gather_into_tensor(gather_bufs[slot], feats)
            feats = gather_bufs[slot]
        if gathered is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(st if st is not None else torch.cuda.current_stream(dev))
            done_events.append(ev)
            if xs is not None and (i + 1) % chunk == 0 and i + 1 <= n_exchanged:
                gather_behind(i + 1 - chunk, i + 1)      # this chunk's rows travel while the next steps run
        return feats

    def finish_exchange(n_steps):
        """What is still due of the exchange after the last step, inside the timed region: 'once' = the one all-gather
        behind every stream of the pipeline (no host wait); 'chunked' = the tail chunk (when the chunk size does not
        divide the steps) and the edge that makes the current stream wait for the exchange stream.  Returns an
        event behind the whole exchange."""
        if gathered is None or n_steps <= 0:
            return None
        cur = torch.cuda.current_stream(dev)
        if xs is None:
            hp.join_to_current()
            gather_chunk(0, n_steps)
        else:
            sent = (n_steps // chunk) * chunk
            if sent < n_steps:
                gather_behind(sent, n_steps)
            cur.wait_stream(xs)
        end = torch.cuda.Event(enable_timing=True)
        end.record(cur)
        return end

    DIAG_IDLE_MS = float(os.environ.get("LSM_BENCH_DIAG_IDLE_MS", "0"))

    def fence():
        """The bracket of the timed region: barrier + torch.cuda.synchronize().  With several ranks the barrier (RCCL: a
        one-element all-reduce the host waits for) is enqueued BEHIND this rank's GPU work -- the current stream first waits
        for every stream of the pipeline, GPU-side -- so the host waits once, for "my work is done and every rank got
        here", and the synchronisation that follows finds an idle GPU; the other order (synchronize, barrier,
        synchronize) costs two more host round trips, each of them idle GPU time (inside the timed region at its end, in
        front of it at its start, where the clock governor sees it: profiles/r04_one_rank_rccl_fences.txt).
        Returns the host time of the two calls (ms)."""
        f0 = time.perf_counter()
        if use_dist:
            hp.join_to_current()
            if xs is not None:
                torch.cuda.current_stream(dev).wait_stream(xs)
            dist.barrier()
        f1 = time.perf_counter()
        torch.cuda.synchronize()
        return (f1 - f0) * 1e3, (time.perf_counter() - f1) * 1e3

    def rehearse_collectives():
        """RCCL sets itself up on FIRST use -- of the barrier's all-reduce (5-16 ms on one rank), of an all-gather, perhaps of
        every message size (protocol, buffers).  All of it happens here, before the pipeline is primed: a first use during
        the warm-up steps blocks the host for milliseconds while the GPU runs out of queued work, and the fence then
        finds a GPU that has been idle for that long -- the timed region starts at an idle chip's clock (traced:
        profiles/r04_one_rank_rccl_fences.txt, 3 ms of idleness cost the 20-step burst 6 %)."""
        if not use_dist:
            return
        dist.barrier()
        if gathered is not None:
            sizes = ({min(chunk, args.steps), args.steps % chunk, min(args.warmup, args.steps) % chunk} - {0}
                     if xs is not None else {args.steps, min(args.warmup, args.steps)} - {0})
            st = xs if xs is not None else torch.cuda.current_stream(dev)
            with torch.cuda.stream(st):
                for k in sorted(sizes):
                    gather_chunk(0, k)
        if gather_bufs is not None:
            dist.all_gather_into_tensor(gather_bufs[0], torch.empty((B, n_feat), dtype=torch.float32, device=dev))
        torch.cuda.synchronize()

    def timed_pass(prime_ms):
        """Set-up (prime), W untimed warm-up steps, then EXACTLY K timed steps between two fences."""
        rehearse_collectives()
        primed = hp.prime(stage_in, stage=args.stage, min_ms=prime_ms)
        hp.fork_from_current()             # inputs were produced on the default stream
        n_w = min(args.warmup, args.steps)
        done_events.clear()
        for i in range(args.warmup):
            step(i, n_w)
        finish_exchange(n_w)
        start_fence = fence()
        if DIAG_IDLE_MS > 0:               # diagnostic: an idle GPU for this long right before the timed region
            time.sleep(DIAG_IDLE_MS / 1e3)
        hp.reservoir_events.clear()
        done_events.clear()
        collective_host_ms.clear()
        t0 = time.perf_counter()
        out = None
        for i in range(args.steps):
            out = step(i, args.steps, tail=i >= args.steps - tail_steps)   # what HotPath.run() does for a finite list
        enqueue_ms = (time.perf_counter() - t0) / args.steps * 1e3      # host side of a step (asynchronous)
        x_end = finish_exchange(args.steps)                              # inside the timed region, before the fence
        end_fence = fence()
        elapsed = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # exposed tail of the exchange: from the moment the LAST step finished (whichever stream it ran on) to the end
        # of the exchange -- what the collective adds to the timed region
        exposed = None
        if x_end is not None and done_events:
            exposed = max(0.0, min(ev.elapsed_time(x_end) for ev in done_events))
        return {"elapsed": elapsed, "enqueue_ms": enqueue_ms, "primed": primed, "out": out, "exchange_ms": exposed,
                "ev_pairs": list(hp.reservoir_events),
                "fences": {"start_barrier_ms": round(start_fence[0], 4), "start_synchronize_ms": round(start_fence[1], 4),
                           "end_barrier_ms": round(end_fence[0], 4), "end_synchronize_ms": round(end_fence[1], 4)}}

    # the 40 ms of HotPath.prime() are a measurement-protocol choice (the clock governor at its working
    # point, profiles/r03_clock_ramp.txt).  The first pass keeps round 2's protocol -- one step per stream, W warm-up
    # steps, K timed steps -- and its figure is reported as `unprimed`; the second pass is the headline.
    unprimed = None
    if world == 1 and not use_dist and not args.no_unprimed and args.prime_ms > 0:
        u = timed_pass(0.0)
        unprimed = {"value": round(B * args.steps / u["elapsed"], 2), "ms_per_step": round(u["elapsed"] / args.steps * 1e3, 4),
                    "protocol": f"first pass of this run: {u['primed']} untimed steps (one per stream) + {args.warmup} "
                                f"warm-up steps, then {args.steps} timed steps (the round-2 protocol)"}
    res_pass = timed_pass(args.prime_ms)
    elapsed, host_enqueue_ms, primed_steps, out = (res_pass["elapsed"], res_pass["enqueue_ms"], res_pass["primed"],
                                                   res_pass["out"])
    exchange_ms = res_pass["exchange_ms"]
    if use_dist and exchange_ms is not None:             # the slowest rank's tail
        t = torch.tensor([exchange_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exchange_ms = float(t.item())
    hp.reservoir_events[:] = res_pass["ev_pairs"]
    ev_pairs = list(hp.reservoir_events)

    def median_ms(fn, reps=5):
        times = []
        for _ in range(reps):
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a0.record()
            fn()
            a1.record()
            torch.cuda.synchronize()
            times.append(a0.elapsed_time(a1))
        return sorted(times)[len(times) // 2]

    # outside the timed region, on an otherwise idle GPU: the front end alone, the reservoir launch in the
    # pipeline's layout, and a lone launch in the layout the library picks for that
    fe_ms = gt_ms = serial_ms = lone_ms = None
    if rank == 0 and args.stage != "reservoir":
        fe_ms = median_ms(lambda: fe.encode(audio))
        if cfg["filterbank"] == "gammatone":
            gt_ms = median_ms(lambda: fe.spectrogram_db(audio))
    if rank == 0 and args.stage != "frontend":
        lone_ms = median_ms(lambda: net.run_batch(rasters0, FEATURE_SET, waves_per_clip=0))
        serial_ms = median_ms(lambda: net.run_batch(rasters0, FEATURE_SET, waves_per_clip=hp.waves_per_clip))

    spikes_per_clip = spikes_total = None
    if rank == 0 and args.stage != "frontend":
        st = torch.empty((B, 2), dtype=torch.int32, device=dev)       # untimed: reservoir spikes of the batch
        net.run_batch(rasters0, FEATURE_SET, stats_out=st)
        spikes_total = int(st[:, 1].sum())
    if args.stage != "frontend":
        rows = local_rows[args.steps - 1] if local_rows is not None else out[:B]
        spikes_per_clip = float(rows.float()[:, :cfg["n_out"]].sum(dim=1).mean())
    exchange_digest = None
    if gathered is not None and args.stage != "frontend":
        # the gathered block holds every rank's rows, chunk by chunk: put them back into (rank, step, clip) order --
        # this rank's rows must come back unchanged, and the digest of the whole block is the same number whatever the
        # exchange mode (tests/test_gpu_hotpath.py compares 'once' and 'chunked')
        by_rank = lsm_dist.rows_by_rank(gathered[: world * args.steps * B], world, args.steps, chunk, B)
        assert torch.equal(by_rank[rank], local_rows[:args.steps].reshape(-1, n_feat)), "all-gather returned other rows"
        exchange_digest = lsm_dist.rows_digest(by_rank)                  # wraps in int64: deterministic
        del by_rank

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        line = {
            "metric": "clips/sec (1 s@16 kHz, 128-filter, 1000-neuron LSM)" if args.config == "cfg2"
                      else f"clips/sec ({args.config})",
            "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (reservoir) / f64 (gammatone)" if cfg["filterbank"] == "gammatone"
                     else "f32 (reservoir, mel projection) / f64 (FFT)",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {cfg['desc']}", "stage": args.stage,
                       "clips_per_gpu": B, "n_filters": cfg["n_filters"], "num_neurons": cfg["N"],
                       "small_world_k": cfg["k"], "num_output_neurons": cfg["n_out"],
                       "time_steps": fe.n_steps, "feature_set": "original",
                       "reservoir_kernel": net.kernel_in_use(),
                       "reservoir_launch_order": ("longest clips first (lsm_reservoir_run_ordered)"
                                                  if net.longest_first_default(B) else "batch order"),
                       "waves_per_clip": lay["waves_per_clip"], "lds_bytes_per_clip": lay["lds_bytes"],
                       "inputs": "pinned host memory, copied every step" if args.from_host else "resident in HBM",
                       "host_enqueue_ms_per_step": round(host_enqueue_ms, 4),
                       "streams": hp.n_streams, "fe_streams": hp.n_fe_streams, "hw_queues": hp.hw_queues,
                       "prime": (f"{primed_steps} untimed steps (>= {args.prime_ms:g} ms) in HotPath.prime() before the "
                                 f"{args.warmup} warm-up steps" +
                                 ("; a whole unprimed pass (one step per stream, the warm-up steps and the timed steps, "
                                  "reported as `unprimed`) ran before this one" if unprimed is not None else "")),
                       "tail_steps": f"the last {tail_steps} timed step(s) are submitted with tail=True (low-latency layouts), as "
                                     f"HotPath.run() submits the last batches of a finite list",
                       "pipeline": ("serial" if hp.n_streams <= 1 else
                                    "pipeline.HotPath: steps rotate over the streams" if not hp.n_fe_streams else
                                    "pipeline.HotPath: front ends on their own streams, reservoir launches behind events"),
                       "mean_output_spikes_per_clip": spikes_per_clip,
                       "setup_s": setup_s,
                       "sharding": ((f"clips x{world}, feature rows all-gathered every {chunk} steps on a stream of their "
                                     f"own while the next steps run; only the last chunk's gather is exposed"
                                     if xs is not None else
                                     f"clips x{world}, one feature all-gather after the last step (as the product: one "
                                     f"gather per split)") if once else
                                    f"clips x{world}, feature all-gather behind every step") if use_dist and world > 1
                                   else ("single GPU" if not use_dist else f"1 rank, distributed code path ({args.exchange})")},
        }
        if unprimed is not None:
            line["unprimed"] = unprimed
        # host clock, this rank: the two calls of the fences that bracket the timed region (the barrier, enqueued behind
        # this rank's GPU work, waits for that work AND the other ranks; 0 without a process group).  end_* lie INSIDE the
        # timed region
        line["fences"] = res_pass["fences"]
        if use_dist:                # what this rank's process saw: both are read once, when HIP / RCCL initialise
            line["rank_env"] = {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "GPU_MAX_HW_QUEUES",
                                                               "MASTER_ADDR", "WORLD_SIZE")}
        if DIAG_IDLE_MS > 0:
            line["fences"]["diag_idle_ms_before_region"] = DIAG_IDLE_MS
        if use_dist and once and args.stage != "frontend":
            line["exchange"] = {
                "mode": args.exchange, "chunk_steps": chunk if xs is not None else args.steps,
                "exchange_ms": None if exchange_ms is None else round(exchange_ms, 4),
                "exchange_bytes": world * args.steps * B * n_feat * 4,
                "bytes_sent_per_rank": args.steps * B * n_feat * 4, "digest": exchange_digest,
                "host_enqueue_ms_per_collective": [round(x, 4) for x in collective_host_ms] or None,
                "note": "exchange_ms = exposed tail: from the moment the last step's kernels finished to the end of the "
                        "exchange (HIP events, max over ranks), inside the timed region; exchange_bytes = what every "
                        "rank holds afterwards"}
        if ev_pairs:
            lif_ms = sum(a.elapsed_time(b) for a, b in ev_pairs) / len(ev_pairs)
            w_bytes = res.csr_bytes()
            per_clip = fe.n_channels * fe.n_steps + n_feat * 4 + fe.n_steps * w_bytes / B
            compulsory = fe.n_channels * fe.n_steps + n_feat * 4 + w_bytes / B
            kname = {"dense": "lif_dense_kernel", "ring": "lif_ring_kernel", "sparse": "lif_kernel"}[net.kernel_in_use()]
            if net.plan(B, fe.n_steps, 0)["input_mode"] in (14, 15):
                kname = "lif_pair_kernel"           # ring rows shared out in pair blocks (csrc/lif_pair.h)
            traffic = None
            tfile = os.environ.get("LSM_TRAFFIC_FILE") or os.path.join(ROOT, "profiles", "lif_traffic.json")
            tkey = f"{args.config}_B{B}_{net.kernel_in_use()}"
            if os.path.exists(tfile):
                traffic = json.load(open(tfile)).get(tkey)
            # NOT measured in this run: the PMC passes are separate rocprofv3 runs whose per-launch result is committed
            traffic_source = (f"{os.path.relpath(tfile, ROOT)}[{tkey}]: rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, separate "
                              f"passes on lone launches of the product's launch (clips started longest first where the "
                              f"product does), (2*FETCH + WRITE)*1024 per launch -- a committed constant, "
                              f"not measured in this run" if traffic is not None else
                              f"none: {os.path.relpath(tfile, ROOT)} has no entry '{tkey}' (no --pmc passes were "
                              f"collected for this shape)")
            # The layout of a LONE launch is what `kernel_ms` times, so its plan is what the gather figures describe
            plan = net.plan(B, fe.n_steps, 0)
            ceiling, ceiling_note = gather_ceiling_gbs(plan["table_bytes"])
            row_bytes = plan["row_request_bytes"]
            # `frac` is built on a duration that cannot exceed the step time and does not move with the
            # number of launches the pipeline happens to overlap -- the lone launch, timed in this run with HIP events
            # on the stream the kernel is launched on (median of 5).  The HIP-event average over the timed region
            # (launches of consecutive steps overlapping each other and the front ends) is reported beside it as
            # in_region_*: it grows when the pipeline overlaps MORE, i.e. when throughput rises.
            k_ms = lone_ms if lone_ms is not None else lif_ms
            achieved = per_clip * B / (k_ms * 1e-3) / 1e9
            bound_by = {
                "dense": "per-step issue/latency chain of one clip (weight table L2-resident, HBM idle): neither HBM nor "
                         "the L2 gather rate is the limit",
                "ring": ("gather rate of the Infinity Cache / fabric (table beyond the L2s, most of it re-read per clip)"
                         if plan["table_bytes"] > 40e6 else
                         "row-gather latency at the occupancy LDS and registers allow (table in the Infinity Cache, "
                         "bandwidth far from saturated)"),
                "sparse": "ordered LDS read-modify-write chain per step"}[net.kernel_in_use()]
            # `bound` names what bounds THIS kernel: "latency" = one clip's per-step dependency chain (table
            # L2-resident), "gather" = the row gathers from the Infinity Cache / fabric, "hbm" would be streamed bytes.
            # achieved / peak / frac stay SURVEY.md 8(d)'s streamed-bytes MODEL priced against the HBM peak (`model`);
            # hbm_frac_measured is the counter traffic of a launch over its duration over that peak: what the memory
            # side really moves.
            bound = {"dense": "latency", "sparse": "latency",
                     "ring": "gather" if plan["table_bytes"] > 40e6 else "latency"}[net.kernel_in_use()]
            line["roofline"] = {
                "bound": bound, "model": "hbm", "kernel": kname, "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "hbm_frac_measured": None if traffic is None else round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "traffic": traffic, "traffic_source": traffic_source, "kernel_ms": round(k_ms, 4),
                "kernel_ms_source": "lone launch of the batch in the library's own layout, HIP events on the launch "
                                    "stream, median of 5, measured in this run after the timed region",
                "bound_by": bound_by,
                "bytes_per_clip": round(per_clip, 1),
                "variant": "streamed (C*T + 4*F_feat + T*|W|/B, SURVEY.md 8d); |W| = 8 B x nnz",
                "traffic_over_algorithmic": None if traffic is None else round(traffic / (per_clip * B), 3),
                "compulsory_bytes_per_clip": round(compulsory, 1),
                "compulsory_gbs": round(compulsory * B / (k_ms * 1e-3) / 1e9, 3),
                "kernel_clips_per_s": round(B / (k_ms * 1e-3), 1),
                "lone_launch_waves_per_clip": plan["waves_per_clip"],
                # the same algorithmic bytes against the WALL CLOCK of the path: one launch's bytes are consumed per step
                "pipeline_gbs": round(per_clip * B / (ms_step * 1e-3) / 1e9, 2),
                "pipeline_frac": round(per_clip * B / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                # inside the timed region: duration of a launch while others overlap it, and how many do on average
                "in_region_kernel_ms": round(lif_ms, 4),
                "in_region_frac": round(per_clip * B / (lif_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "launches_in_flight": round(lif_ms / ms_step, 2),
                "in_region_waves_per_clip": lay["waves_per_clip"],
                "idle_gpu_kernel_ms": None if serial_ms is None else round(serial_ms, 4),
                "idle_gpu_frac": None if serial_ms is None else
                round(per_clip * B / (serial_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "note": "frac = achieved / peak with achieved = algorithmic bytes of one launch / kernel_ms (the lone "
                        "launch); in_region_* = the HIP-event average over the timed region, where launches of consecutive "
                        "steps overlap (not a throughput figure); idle_gpu_* = the pipeline's layout launched alone",
                # what an event-driven kernel really asks of the memory system: one weight row per reservoir spike
                "weight_table_bytes": plan["table_bytes"], "gather_ceiling": ceiling, "gather_ceiling_note": ceiling_note,
                "row_gather": None if lone_ms is None or spikes_total is None else {
                    "rows_per_launch": spikes_total, "mean_row_bytes": round(row_bytes, 1),
                    "requested_bytes": int(spikes_total * row_bytes),
                    "gbs_lone_launch": round(spikes_total * row_bytes / (lone_ms * 1e-3) / 1e9, 1),
                    # > 1: the L2s serve the rest of the requests; this is a REQUEST rate, so it is not set against the
                    # memory-side ceiling (that is memory_side_frac's job: r03's cfg5 line read 1.22 for this reason)
                    "requested_over_memory_side": None if traffic is None else round(spikes_total * row_bytes / traffic, 3),
                    "note": "reservoir spikes of the batch (stats_out of one untimed launch) x the bytes ONE spike requests "
                            "from the table in use (lsm_reservoir_row_request_bytes: the window's existing bytes + its list "
                            "entries + row pointers, mean over rows; table padding is not charged) / lone-launch time"},
                "memory_side_gbs_lone_launch": None if traffic is None or lone_ms is None else
                round(traffic / (lone_ms * 1e-3) / 1e9, 1),
                "memory_side_frac": None if traffic is None or lone_ms is None else
                round(traffic / (lone_ms * 1e-3) / 1e9 / ceiling, 4),
            }
            if gt_ms is not None:
                # the kernel that takes most of the GPU time is the float64 filterbank, bound by vector-ALU issue: 36
                # instructions per sample and channel (35.5 in the one-launch front end above 64 filters: two channels
                # per lane share the sample's conversion), none fusable into FMAs without changing the rounding
                fused = os.environ.get("LSM_FRONTEND_SPLIT") != "1" and cfg["n_filters"] <= 1024
                two_chains = fused and cfg["n_filters"] > 64
                ops = (35.5 if two_chains else 36.0) * ((fe.ncols - 1) * fe.hop + fe.nwin) * cfg["n_filters"] * B
                k_ms = fe_ms if fused else gt_ms        # the launch the pipeline runs: whole fused front end, or filterbank
                line["roofline"]["dominant_kernel_by_time"] = {
                    "kernel": "gammatone_spikes_kernel" if fused else "gammatone_kernel", "bound": "valu_f64",
                    "idle_gpu_ms": round(k_ms, 4), "frontend_idle_gpu_ms": round(fe_ms, 4),
                    "split_filterbank_idle_gpu_ms": round(gt_ms, 4),
                    "achieved": round(ops / (k_ms * 1e-3) / 1e12, 2), "peak": F64_UNFUSED_PEAK_TOPS,
                    "unit": "T f64 instr-lanes/s", "frac": round(ops / (k_ms * 1e-3) / 1e12 / F64_UNFUSED_PEAK_TOPS, 4),
                    "pipeline_achieved": round(ops / (ms_step * 1e-3) / 1e12, 2),
                    "pipeline_frac": round(ops / (ms_step * 1e-3) / 1e12 / F64_UNFUSED_PEAK_TOPS, 4),
                    "waves": -(-cfg["n_filters"] // (128 if two_chains else 64)) * B,
                    "note": "peak = one float64 operation per lane and instruction on all 1024 SIMDs with >= 4 waves "
                            "each (78.6 TFLOPS counts an FMA as two); idle_gpu = ONE launch alone (the one-launch front end "
                            "occupies a quarter of the chip at 128 filters / 256 clips: four in flight cover it), pipeline = "
                            "the same instruction count over the measured step time of the whole overlapped path; at cfg2 "
                            "the overlapped path runs at the socket's power limit (1380-1390 of 1400 W, shader clock throttled "
                            "to 2.29-2.33 GHz: profiles/r04_clock_power_under_load.txt, not measured in this run)",
                }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, audio_np, res)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


def main():
    args = parse_args()
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    # LSM_BENCH_FORCE_SPAWN=1: go through the launcher even for one rank (tests/test_gpu_bench_contract.py checks what a
    # rank started by it inherits)
    if (args.gpus > 1 or os.environ.get("LSM_BENCH_FORCE_SPAWN") == "1") and not launched:
        sys.exit(spawn_ranks(max(1, args.gpus)))    # the parent never touches the GPU
    if os.environ.get("LSM_BENCH_SPAWN_ONLY") == "1" and launched:
        return spawn_check()
    run_rank(args)


if __name__ == "__main__":
    main()
