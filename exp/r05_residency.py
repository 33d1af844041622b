"""Round 5 (VERDICT r4 #1a, "predict before building"): what do more resident clips per compute unit buy the pair-block ring kernel?

Time-only ablation.  Pass 1 (`record`, product library): cfg4's batch through the reservoir kernel with the spike matrix written
out -> /tmp/r05_sm.npy.  Pass 2 (`replay`, a -DLSM_PAIR_REPLAY=1 build named by LSM_HIP_LIB): the same launch, every neuron's
spikes taken from that record, so the rows applied in every step are those of the real run whatever the build leaves out (LEAN:
no input masks, no slot/refractory register, no feature records in LDS -- the registers and LDS of a kernel whose input counts are
precomputed and whose features accumulate outside LDS).  The builds differ in registers (waves per SIMD) and LDS (clips per CU).
Features of a replay run are garbage; only the launch time is read.
"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from lsm_speech_classifier_amd import frontend, reservoir, snn, _lib
from lsm_speech_classifier_amd.snn import _dev, _host

SM = "/tmp/r05_sm.npy"


def setup(name, B):
    cfg = bench.CONFIGS[name]
    dev = torch.device("cuda", 0)
    fe = frontend.SpikeFrontEnd(cfg["n_filters"], cfg["filterbank"], device=dev)
    audio = torch.from_numpy(bench.make_audio(cfg["audio"], B, seed=1234)).to(dev)
    rasters = fe.encode(audio)
    wc = bench.w_critico(cfg["k"], 2.0, 2, rasters)
    params = reservoir.SimulationParams(num_neurons=cfg["N"], num_output_neurons=cfg["n_out"],
                                        small_world_graph_k=cfg["k"], mean_weight=wc * bench.MULTIPLIER)
    net = snn.SNN(params, reservoir=reservoir.build_reservoir(params, fe.n_channels), device=dev)
    return fe, rasters, net


def launch(net, rasters, sm, keys, reps):
    B, _, T = rasters.shape
    key_ids = np.array([snn.FEATURE_KEYS.index(k) for k in keys], dtype=np.int32)
    feats = torch.empty((B, len(keys) * net.num_output_neurons), dtype=torch.float32, device=rasters.device)
    need = net.lib.lsm_reservoir_order_workspace(B)
    ws = torch.empty((need + 3) // 4, dtype=torch.int32, device=rasters.device)
    stream = torch.cuda.current_stream().cuda_stream
    ms = []
    for r in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(net.lib.lsm_reservoir_run_ordered(net._handle, _dev(rasters), B, T, _host(key_ids), len(keys), _dev(feats),
                                                     _dev(sm), None, None, 0, _dev(ws), need, stream), "run")
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            ms.append(e0.elapsed_time(e1))
    return float(np.median(ms)), feats


def main():
    mode, name, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
    fe, rasters, net = setup(name, B)
    T = fe.n_steps
    if mode == "record":
        sm = torch.empty((B, T, net.num_neurons), dtype=torch.uint8, device=rasters.device)
        ms, _ = launch(net, rasters, sm, bench.FEATURE_SET, 3)
        np.save(SM, sm.cpu().numpy())
        ms0, _ = launch(net, rasters, None, bench.FEATURE_SET, 5)
        print(f"record: {name} B={B}: product launch {ms0:.3f} ms ({ms:.3f} ms writing the spike matrix); "
              f"{float(sm.sum()) / B / T:.2f} rows per step; plan {net.plan(B, T, 0)}")
    else:
        sm = torch.from_numpy(np.load(SM)).to(rasters.device)
        assert tuple(sm.shape) == (B, T, net.num_neurons)
        ms, _ = launch(net, rasters, sm, bench.FEATURE_SET, 7)
        lay = net.layout(B, T, 0)
        per_cu = min(160 * 1024 // lay["lds_bytes"], 32 // lay["waves_per_clip"])
        print(f"replay {os.path.basename(os.environ.get('LSM_HIP_LIB', 'product'))}: launch {ms:.3f} ms; LDS {lay['lds_bytes']} B per clip "
              f"-> at most {per_cu} clips per CU by LDS and wave slots")


if __name__ == "__main__":
    main()
