#!/usr/bin/env python3
"""Hot-path benchmark (driver contract: python bench.py --gpus N --steps K --warmup W).

One STEP = one pass of the whole hot path over one batch of synthetic clips per GPU, inputs
already resident in HBM: audio (B, 16000) f32 -> gammatone filterbank -> dB/normalise/resize ->
hysteresis encoder -> uint8 raster (B, C, 400) -> LIF reservoir (400 steps) -> features
(B, 5*N_out) f32 [-> RCCL all-gather of the feature rows when N > 1].  Default workload is
BASELINE.json configs[1]: 128 gammatone filters, 1000-neuron reservoir, batch 256 per GPU,
`original` feature set, multiplier 0.6.  Clips shard across ranks (weak scaling, no data-path
collective except the final feature gather the reference's single-process run implies).

Prints ONE JSON line on rank 0 with the contract keys plus `roofline` (LIF kernel, algorithmic
bytes of SURVEY.md §8(d) / measured HIP-event time) and, at N=1, `cpu_baseline` (the C oracle
timed on the host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The steps rotate over several HIP streams; the runtime multiplexes streams onto 4 hardware queues by
# default and kernels of streams that share a queue serialise.  With 8 or more queues six streams overlap
# (measured at cfg2: 0.96 -> 0.82 ms per step; exp/hwq_sweep.sh); 12 also leave RCCL's stream a queue of
# its own when N > 1 (exp/dist_rehearsal.sh).  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

import numpy as np
import torch

CONFIGS = {
    # name: (n_filters, filterbank, N, k, N_out, batch per GPU, audio kind)
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


def w_critico(k, theta, refractory, rasters_dev):
    """extract_lsm_features.py:33-60 on the first <=500 clips (device reduction, host scalar)."""
    sub = rasters_dev[:500]
    if sub.numel() == 0 or k == 0:
        return 0.007
    avg_i = float(sub.sum(dtype=torch.int64)) / sub.numel()
    return (theta - 2 * avg_i * refractory) / (k / 2)


def make_audio(kind, n, seed):
    from lsm_speech_classifier_amd import synth
    if kind == "white_noise":
        return synth.white_noise(n, seed=seed)
    return synth.class_chirps(np.arange(n) % 12, seed=seed)


def cpu_baseline(cfg, audio, res, seconds_budget=20.0):
    """C oracle (oracle/, kind 'port') on a bounded sample: one clip at a time on one core, like
    the reference's serial loops (create_dataset.py:143, extract_lsm_features.py:78); then the
    same sample with one clip per core."""
    from lsm_speech_classifier_amd import frontend
    from oracle import cport, ref_numpy
    cport.build()
    coefs = ref_numpy.gammatone_coefs(16000, cfg["n_filters"], 50)

    def front(a):
        return cport.encode_hysteresis(cport.normalise_resize(cport.gammatone_db(
            cport.gammatone_spec(a, coefs, 400, 160, 98))), frontend.SPIKE_THRESHOLDS,
            frontend.HYSTERESIS_GAP)

    def one(a):
        return cport.lif_run(res, front(a), FEATURE_SET, want_spikes=False)[0]

    one(audio[0])                                   # warm-up
    t0 = time.perf_counter()
    one(audio[1 % len(audio)])
    per_clip = time.perf_counter() - t0
    n = int(max(2, min(len(audio), seconds_budget / 2 / max(per_clip, 1e-4))))
    t0 = time.perf_counter()
    for a in audio[:n]:
        one(a)
    t_serial = time.perf_counter() - t0
    cores = os.cpu_count() or 1
    rasters = np.stack([front(a) for a in audio[:n]])
    t0 = time.perf_counter()
    cport.lif_run_batch(res, rasters, FEATURE_SET, n_threads=cores)
    t_lif_all = time.perf_counter() - t0
    t1 = time.perf_counter()
    for a in audio[:8]:
        front(a)
    t_front = (time.perf_counter() - t1) / min(8, len(audio))
    return {
        "value": round(n / t_serial, 3), "unit": "clips/s", "cores": 1, "kind": "port",
        "sample": f"{n} clips of the same workload, C oracle (gather-form LIF + gammatone), one clip "
                  f"at a time on one core; front end {t_front * 1e3:.1f} ms/clip",
        "all_cores": {"value": round(n / (t_lif_all + n * t_front / cores), 3), "cores": cores,
                      "note": "reservoir with one clip per OpenMP thread; front-end time divided by cores"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU (0 = the config's)")
    ap.add_argument("--waves-per-clip", type=int, default=0)
    ap.add_argument("--stage", default="full", choices=["full", "reservoir", "frontend"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--from-host", action="store_true",
                    help="every step first copies its audio batch from pinned host memory (PCIe-inclusive "
                         "rate; never the contract's `value`, which is quoted on HBM-resident inputs)")
    ap.add_argument("--streams", type=int, default=6,
                    help="HIP streams the steps rotate over (consecutive steps overlap; 1 = serial)")
    ap.add_argument("--pipeline", default="rotate", choices=["rotate", "split"],
                    help="rotate: whole steps round-robin over --streams streams; split: one stream for "
                         "the front end, one for the reservoir, linked by events")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # LSM_BENCH_FORCE_DIST=1 takes the distributed code path (process group, broadcast, gather,
    # barrier) even with one rank: a rehearsal of the RCCL calls on a 1-GPU box
    use_dist = world > 1 or os.environ.get("LSM_BENCH_FORCE_DIST") == "1"
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
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank if use_dist else 0)

    from lsm_speech_classifier_amd import frontend, reservoir, snn
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
    res = reservoir.build_reservoir(params, fe.n_channels)
    net = snn.SNN(params, reservoir=res, device=dev)
    n_feat = len(FEATURE_SET) * cfg["n_out"]
    # Layout of a clip in the reservoir kernel.  The library's own choice (8 waves per clip at B = 256)
    # minimises the duration of a lone launch; inside the overlapped pipeline, which is bound by vector-ALU
    # issue, 4 waves per clip (4 neurons per lane) spend fewer instructions on per-wave overheads and the
    # whole pipeline is 3.5 % faster in a same-box A/B (exp/combo_sweep.sh).
    if args.waves_per_clip == 0 and max(1, args.streams) > 1 and cfg["N"] <= 1024 and args.stage == "full":
        args.waves_per_clip = 4
    lay = net.layout(B, fe.n_steps, args.waves_per_clip)

    ev_pairs = []
    n_streams = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 else [None]
    step_no = [0]
    audio_pinned = torch.from_numpy(audio_np).pin_memory() if args.from_host else None
    h2d_bufs = [torch.empty_like(audio) for _ in range(n_streams)] if args.from_host else None
    # one gather buffer per stream of the rotation: overlapping steps never share an output
    gather_bufs = ([torch.empty((world * B, n_feat), dtype=torch.float32, device=dev)
                    for _ in range(n_streams)] if use_dist else None)

    split = args.pipeline == "split" and args.stage == "full" and n_streams > 1

    def step(timed):
        """One pass of the hot path over the batch, issued on the next stream of the rotation: the
        work of a step is ordered on its own stream, consecutive steps overlap on the GPU."""
        if split:
            return split_step(timed)
        st = streams[step_no[0] % n_streams]
        step_no[0] += 1
        if st is None:
            return one_step(timed)
        with torch.cuda.stream(st):
            return one_step(timed)

    def split_step(timed):
        """Front end of every step on streams[0], reservoir of every step on streams[1]; the
        reservoir of step s waits (event) for the front end of step s only."""
        s_fe, s_lif = streams[0], streams[1]
        with torch.cuda.stream(s_fe):
            rasters = fe.encode(audio)
            ready = torch.cuda.Event()
            ready.record()
        rasters.record_stream(s_lif)
        with torch.cuda.stream(s_lif):
            s_lif.wait_event(ready)
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            feats, _, _ = net.run_batch(rasters, FEATURE_SET, waves_per_clip=args.waves_per_clip)
            if timed:
                e1.record()
                ev_pairs.append((e0, e1))
            if use_dist:
                step_no[0] += 1
                gathered = gather_bufs[step_no[0] % n_streams]
                gather(gathered, feats)
                return gathered
            return feats

    def one_step(timed):
        if args.stage == "reservoir":
            rasters = rasters0
        else:
            src = audio
            if args.from_host:                      # asynchronous H2D on this step's stream, own buffer
                src = h2d_bufs[(step_no[0] - 1) % n_streams] if n_streams > 1 else h2d_bufs[0]
                src.copy_(audio_pinned, non_blocking=True)
            rasters = fe.encode(src)
        if args.stage == "frontend":
            return rasters
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        feats, _, _ = net.run_batch(rasters, FEATURE_SET, waves_per_clip=args.waves_per_clip)
        if timed:
            e1.record()
            ev_pairs.append((e0, e1))
        if use_dist:
            gathered = gather_bufs[(step_no[0] - 1) % n_streams]
            gather(gathered, feats)
            return gathered
        return feats

    def gather(dst, src):
        # RCCL's stream is ordered after this step's reservoir kernel and the step's stream after the
        # gather; with >= 12 hardware queues the exchange does not disturb the other steps in flight
        # (single-rank rehearsal: 0.785 ms per step with the gather, 0.77 without; exp/dist_rehearsal.sh)
        dist.all_gather_into_tensor(dst, src)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for st in streams:                 # inputs were produced on the default stream
        if st is not None:
            st.wait_stream(torch.cuda.current_stream())
    for _ in range(args.warmup):
        out = step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    host_enqueue_ms = (time.perf_counter() - t0) / args.steps * 1e3      # host side of a step (asynchronous)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # outside the timed region: the reservoir kernel alone on an otherwise idle GPU (for reference
    # next to the in-region average, which includes sharing the chip with the overlapped steps)
    fe_ms = None
    if args.stage != "reservoir" and rank == 0:
        pairs = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            fe.encode(audio)
            e1.record()
            torch.cuda.synchronize()
            pairs.append(e0.elapsed_time(e1))
        fe_ms = sorted(pairs)[len(pairs) // 2]
    serial_ms = None
    lone_ms = None
    if args.stage != "frontend" and rank == 0:
        def lone_launch(wpc):
            times = []
            for _ in range(5):
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                a0.record()
                net.run_batch(rasters0, FEATURE_SET, waves_per_clip=wpc)
                a1.record()
                torch.cuda.synchronize()
                times.append(a0.elapsed_time(a1))
            return sorted(times)[len(times) // 2]
        # the library's own layout for a lone launch (8 waves per clip at B = 256), for reference
        lone_ms = lone_launch(0)
        pairs = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            net.run_batch(rasters0, FEATURE_SET, waves_per_clip=args.waves_per_clip)
            e1.record()
            torch.cuda.synchronize()
            pairs.append(e0.elapsed_time(e1))
        serial_ms = sorted(pairs)[len(pairs) // 2]

    spikes_per_clip = None
    if args.stage != "frontend":
        fl = out[:B].float()
        spikes_per_clip = float(fl[:, :cfg["n_out"]].sum(dim=1).mean())

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        line = {
            "metric": "clips/sec (1 s@16 kHz, 128-filter, 1000-neuron LSM)" if args.config == "cfg2"
                      else f"clips/sec ({args.config})",
            "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 (reservoir) / f64 (gammatone)",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {cfg['desc']}", "stage": args.stage,
                       "clips_per_gpu": B, "n_filters": cfg["n_filters"], "num_neurons": cfg["N"],
                       "small_world_k": cfg["k"], "num_output_neurons": cfg["n_out"],
                       "time_steps": fe.n_steps, "feature_set": "original",
                       "waves_per_clip": lay["waves_per_clip"], "lds_bytes_per_clip": lay["lds_bytes"],
                       "inputs": "pinned host memory, copied every step" if args.from_host else "resident in HBM",
                       "host_enqueue_ms_per_step": round(host_enqueue_ms, 4),
                       "streams": n_streams, "hw_queues": int(os.environ["GPU_MAX_HW_QUEUES"]),
                       "pipeline": args.pipeline if n_streams > 1 else "serial",
                       "mean_output_spikes_per_clip": spikes_per_clip,
                       "sharding": f"clips x{world}, feature all-gather" if world > 1 else "single GPU"},
        }
        if ev_pairs:
            lif_ms = sum(a.elapsed_time(b) for a, b in ev_pairs) / len(ev_pairs)
            w_bytes = res.csr_bytes()
            per_clip = fe.n_channels * fe.n_steps + n_feat * 4 + fe.n_steps * w_bytes / B
            compulsory = fe.n_channels * fe.n_steps + n_feat * 4 + w_bytes / B
            achieved = per_clip * B / (lif_ms * 1e-3) / 1e9
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "lif_traffic.json")
            if os.path.exists(tfile):
                traffic = json.load(open(tfile)).get(f"{args.config}_B{B}")
            line["roofline"] = {
                "bound": "hbm", "kernel": "lif_dense_kernel" if cfg["N"] <= 8192 else "lif_kernel", "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic, "kernel_ms": round(lif_ms, 4),
                # the measured memory-side bytes over the duration of the launch they were measured on (a lone
                # launch, one stream): what the kernel really pulls through the memory side of L2
                "traffic_gbs": None if (traffic is None or serial_ms is None) else
                round(traffic / (serial_ms * 1e-3) / 1e9, 1),
                "traffic_frac_of_peak": None if (traffic is None or serial_ms is None) else
                round(traffic / (serial_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "bytes_per_clip": round(per_clip, 1),
                "variant": "streamed (C*T + 4*F_feat + T*|W|/B, SURVEY.md 8d); |W| = 8 B x nnz",
                "compulsory_bytes_per_clip": round(compulsory, 1),
                "compulsory_gbs": round(compulsory * B / (lif_ms * 1e-3) / 1e9, 3),
                "kernel_clips_per_s": round(B / (lif_ms * 1e-3), 1),
                "note": "kernel_ms is the HIP-event average over the timed region, where launches of "
                        "consecutive steps overlap on the GPU; idle_gpu_* is the same launch (same layout) alone; "
                        "lone_launch_* is a lone launch in the layout the library picks for it",
                "lone_launch_kernel_ms": None if lone_ms is None else round(lone_ms, 4),
                "lone_launch_frac": None if lone_ms is None else
                round(per_clip * B / (lone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "lone_launch_waves_per_clip": net.layout(B, fe.n_steps, 0)["waves_per_clip"],
                "idle_gpu_kernel_ms": None if serial_ms is None else round(serial_ms, 4),
                "idle_gpu_frac": None if serial_ms is None else
                round(per_clip * B / (serial_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            }
        if fe_ms is not None and cfg["filterbank"] == "gammatone":
            # front end alone on an idle GPU: float64 VALU-issue bound (36 instructions per sample and
            # channel, none of them fusable into FMAs without changing the rounding), not HBM bound
            ops = 36.0 * ((fe.ncols - 1) * fe.hop + fe.nwin) * cfg["n_filters"] * B
            gt_waves = -(-cfg["n_filters"] // 64) * B
            line["frontend"] = {
                "kernels": "gammatone_kernel + spec_to_spikes_kernel", "idle_gpu_ms": round(fe_ms, 4),
                "bound": "valu_f64", "achieved_tflops": round(ops / (fe_ms * 1e-3) / 1e12, 2),
                "peak_tflops_fma_counted": 78.6, "peak_tops_unfused": 39.3,
                # the same instruction count over the measured step time of the whole (overlapped) pipeline:
                # how much of the chip's f64 issue rate the benchmark as a whole sustains
                "pipeline_tops": round(ops / (ms_step * 1e-3) / 1e12, 2),
                "pipeline_frac_of_unfused_peak": round(ops / (ms_step * 1e-3) / 1e12 / 39.3, 4),
                "note": "one float64 operation per lane and instruction (no FMA contraction allowed): the "
                        "ceiling for this instruction mix is 39.3 Tops/s with all 1024 SIMDs busy and >= 4 "
                        f"waves per SIMD (exp/ubench_f64.hip); this launch has {gt_waves} waves",
            }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, audio_np, res)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
