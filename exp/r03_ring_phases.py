"""Round 3: where a step of the ring-row reservoir kernel spends its time (cfg4 / cfg5).

Runs bench.py's workload through a diagnostic build (-DLSM_RING_PHASES=1, LSM_HIP_LIB=...liblsm_hip_phases.so): every wave sums
the core-clock cycles (s_memtime) between fixed points of its time step and writes the sums over its clip's feature row.
Prints cycles per wave and step, averaged over the batch, per phase.  The feature values of such a run are NOT features.
"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from lsm_speech_classifier_amd import frontend, reservoir, snn

PHASES = ["quad counts + total", "chunk set-up (prefix, list, geometry, list pointers)", "first P rows requested",
          "rows applied (incl. waiting for row loads)", "input counts (incl. fetching the input-map entries)",
          "update + lists + features", "barrier"]

def main():
    name, B = sys.argv[1], int(sys.argv[2])
    cfg = bench.CONFIGS[name]
    dev = torch.device("cuda", 0)
    fe = frontend.SpikeFrontEnd(cfg["n_filters"], cfg["filterbank"], device=dev)
    audio = torch.from_numpy(bench.make_audio(cfg["audio"], B, seed=1234)).to(dev)
    rasters = fe.encode(audio)
    wc = bench.w_critico(cfg["k"], 2.0, 2, rasters)
    params = reservoir.SimulationParams(num_neurons=cfg["N"], num_output_neurons=cfg["n_out"],
                                        small_world_graph_k=cfg["k"], mean_weight=wc * bench.MULTIPLIER)
    net = snn.SNN(params, reservoir=reservoir.build_reservoir(params, fe.n_channels), device=dev)
    lay = net.layout(B, fe.n_steps, 0)
    for _ in range(2):
        feats = net.run_batch(rasters, bench.FEATURE_SET)
        feats = feats[0] if isinstance(feats, tuple) else feats
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); net.run_batch(rasters, bench.FEATURE_SET); ev1.record(); torch.cuda.synchronize()
    wpc = lay["waves_per_clip"] if isinstance(lay, dict) else 8
    f = feats.cpu().numpy()[:, :wpc * 8].reshape(B, wpc, 8)
    T = fe.n_steps
    per = f[:, :, :7].mean(axis=(0, 1)) / T
    rows = f[:, :, 7].mean() / T
    tot = per.sum()
    print(f"{name} B={B} kernel {net.kernel_in_use()} layout {lay}: launch {ev0.elapsed_time(ev1):.3f} ms; "
          f"{rows:.1f} rows per step; {tot:.0f} cycles per wave and step")
    for k, p in enumerate(PHASES):
        print(f"  {per[k]:8.0f} cycles  {100 * per[k] / tot:5.1f} %  {p}")
    slow = f[:, :, :7].sum(axis=2).max(axis=1)        # a clip's slowest wave = the clip's length in cycles
    print(f"  per clip: mean {slow.mean() / T:.0f} cycles per step, slowest clip {slow.max() / T:.0f}, fastest {slow.min() / T:.0f}")
    b = int(slow.argmax())
    pb = f[b, :, :7].mean(axis=0) / T
    print(f"  slowest clip ({f[b, :, 7].mean() / T:.1f} rows per step): " + ", ".join(f"{v:.0f}" for v in pb))

    # what the launch order is worth: the workgroups of a launch are handed to the first free slot (two clips per CU at cfg4),
    # restated as a greedy list schedule over the measured per-clip durations
    import heapq
    def makespan(order, slots):
        h = [0.0] * slots
        heapq.heapify(h)
        for c in order:
            heapq.heappush(h, heapq.heappop(h) + float(slow[c]))
        return max(h)
    lds = lay["lds_bytes"]
    per_cu = max(1, min(160 * 1024 // lds, 32 // wpc))
    slots = per_cu * torch.cuda.get_device_properties(0).multi_processor_count
    keys = rasters.reshape(B, -1).sum(dim=1).cpu().numpy()
    os.makedirs("gpurun_out", exist_ok=True)
    np.save(f"gpurun_out/r03_clip_cycles_{name}_B{B}.npy", np.stack([keys.astype(np.float64), slow.astype(np.float64)]))
    lpt = np.argsort(-keys, kind="stable")
    ideal = np.argsort(-slow, kind="stable")
    tot = slow.sum() / slots
    print(f"  {slots} slots: work / slots = {tot / 1e6:.2f} Mcycles, longest clip {slow.max() / 1e6:.2f}; list schedule in batch order "
          f"{makespan(range(B), slots) / 1e6:.2f}, longest input first {makespan(lpt, slots) / 1e6:.2f}, longest clip first "
          f"{makespan(ideal, slots) / 1e6:.2f} Mcycles")
    print(f"  rank correlation input spikes / clip cycles: {np.corrcoef(np.argsort(np.argsort(keys)), np.argsort(np.argsort(slow)))[0, 1]:.3f}")


if __name__ == "__main__":
    main()
