"""Diagnostic: per-phase cycle shares of the LIF kernel (needs a -DLSM_STAMP=1 build via LSM_HIP_LIB)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lsm_speech_classifier_amd import _lib, reservoir, snn, synth, frontend
import bench
wpc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
lib = _lib.load()
fe = frontend.SpikeFrontEnd(128, "gammatone")
audio = torch.from_numpy(bench.make_audio("speech_like", B, 1234)).cuda()
r = fe.encode(audio)
wc = bench.w_critico(200, 2.0, 2, r)
p = reservoir.SimulationParams(num_neurons=1000, num_output_neurons=400, small_world_graph_k=200, mean_weight=wc * 0.6)
net = snn.SNN(p, n_channels=128)
if os.environ.get("LSM_KERNEL"):
    net.set_kernel(os.environ["LSM_KERNEL"])
out = (ctypes.c_ulonglong * 8)()
ptr = ctypes.cast(out, ctypes.c_void_p)
net.run_batch(r, bench.FEATURE_SET, waves_per_clip=wpc); torch.cuda.synchronize()
lib.lsm_debug_lif_stamps(ptr, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
f, sm, _ = net.run_batch(r, bench.FEATURE_SET, waves_per_clip=wpc, want_spike_matrix=False)
e1.record(); torch.cuda.synchronize()
print("kernel wall ms (events):", round(e0.elapsed_time(e1), 4))
lib.lsm_debug_lif_stamps(ptr, 1)
v = np.array(list(out), dtype=np.float64) / B / 400
names = (["scan+list", "seg+syn loads", "input drive", "rmw chain", "update", "barrier", "-", "-"]
         if os.environ.get("LSM_KERNEL") == "sparse" else
         ["list read", "row loads issued", "load wait + adds", "update + ballots", "list write", "barrier", "-", "-"])
raw = np.array(list(out), dtype=np.float64)
print(f"loop: {raw[6]/B:.0f} shader cycles, {raw[7]/B*10:.0f} ns per clip -> clock {raw[6]/raw[7]*100:.0f} MHz; loop time {raw[7]/B/100:.1f} us")
print(f"wpc {wpc} B {B}: cycles per step (wave 0):", {n: round(x, 1) for n, x in zip(names, v) if n != "-"}, "total", round(v.sum(), 1),
      "out-neuron spikes/step", round(float(f[:, :400].sum()) / B / 400, 2))
