"""Large-N validation: build, layout, parity spot check vs the C oracle, reservoir timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lsm_speech_classifier_amd import frontend, reservoir as R, snn, synth
from oracle import cport, ref_numpy as O
import bench
name = sys.argv[1]; B = int(sys.argv[2]); ncheck = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = bench.CONFIGS[name]
t0 = time.time()
fe = frontend.SpikeFrontEnd(cfg["n_filters"], "gammatone")
audio = bench.make_audio(cfg["audio"], B, 1234)          # B distinct clips (tiled copies would share L2 lines)
r = fe.encode(torch.from_numpy(audio).cuda()); torch.cuda.synchronize()
print(f"front end ok {tuple(r.shape)} density {float(r.float().mean()):.3f} ({time.time()-t0:.1f}s)", flush=True)
wc = bench.w_critico(cfg["k"], 2.0, 2, r)
p = R.SimulationParams(num_neurons=cfg["N"], num_output_neurons=cfg["n_out"], small_world_graph_k=cfg["k"], mean_weight=wc * 0.6)
t0 = time.time(); res = R.build_reservoir(p, fe.n_channels); print(f"reservoir built nnz={res.nnz} ({time.time()-t0:.1f}s)", flush=True)
net = snn.SNN(p, reservoir=res)
net.set_kernel(os.environ.get('LSM_KERNEL', 'auto'))
print("layout", net.layout(B, 400), flush=True)
for wpc in ([0] if len(sys.argv) < 5 else [int(x) for x in sys.argv[4].split(",")]):
    f, _, _ = net.run_batch(r, bench.FEATURE_SET, waves_per_clip=wpc); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f, _, _ = net.run_batch(r, bench.FEATURE_SET, waves_per_clip=wpc); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    per_clip = fe.n_channels * 400 + f.shape[1] * 4 + 400 * res.csr_bytes() / B
    print(f"wpc {wpc or net.layout(B,400)['waves_per_clip']}: LIF {ms:.2f} ms for {B} clips = {B/ms*1e3:.0f} clips/s; streamed {per_clip*B/ms/1e6:.1f} GB/s ({per_clip*B/ms/1e6/8000*100:.2f}% of 8 TB/s); "
          f"spikes/out-neuron/clip {float(f[:, :cfg['n_out']].mean()):.2f}", flush=True)
if ncheck <= 0:
    sys.exit(0)
rn = r[:ncheck].cpu().numpy()
t0 = time.time(); ref = cport.lif_run_batch(res, rn, bench.FEATURE_SET, n_threads=min(ncheck, os.cpu_count()))
print(f"oracle {ncheck} clip(s) in {time.time()-t0:.1f}s; bit-exact: {np.array_equal(ref, f[:ncheck].cpu().numpy())}", flush=True)
