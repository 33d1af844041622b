"""Round 3: random reservoirs / batch sizes through lsm_reservoir_run_ordered against the plain launch (every case) and the C oracle
(every clip of every third case): features, statistics, spike matrices.  Batches are larger than the GPU's compute-unit count, so the
order is in force; kernels rotate through dense / ring / sparse where the reservoir has them."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lsm_speech_classifier_amd import reservoir as R, snn, _lib
from oracle import cport, ref_numpy as O

KEYS = ['spike_counts', 'spike_variances', 'mean_spike_times', 'first_spike_times', 'last_spike_times', 'mean_isi',
        'isi_variances', 'burst_counts']
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
cus = torch.cuda.get_device_properties(0).multi_processor_count
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
ok = 0
t_start = time.time()
for case in range(n_cases):
    n = int(rs.choice([200, 333, 700, 1000, 1300, 1536, 2048, 3072]))
    k = max(4, int(n * rs.uniform(0.05, 0.2)) // 2 * 2)
    c = int(rs.choice([8, 24, 40, 128]))
    t = int(rs.choice([17, 40, 64, 100]))
    b = cus + int(rs.randint(1, 500))
    dens = rs.uniform(0.0, 0.5, size=b)
    dens[rs.randint(0, b, size=3)] = 0.0
    rasters = (rs.random_sample((b, c, t)) < dens[:, None, None]).astype(np.uint8)
    wc = O.w_critico(k, 2.0, 2, rasters[:64])
    res = R.build_reservoir(R.SimulationParams(num_neurons=n, num_output_neurons=max(1, n // int(rs.choice([2, 3, 5]))),
                                               small_world_graph_k=k, mean_weight=wc * rs.uniform(0.6, 1.6)), c)
    net = snn.SNN(None, reservoir=res)
    kernel = ["dense", "ring", "sparse"][case % 3]
    try:
        net.set_kernel(kernel)
        net.plan(b, t, 0)
    except _lib.LsmHipError:
        kernel = "auto"
        net.set_kernel("auto")
    dev = torch.from_numpy(rasters).cuda()
    sa = torch.empty((b, 2), dtype=torch.int32, device="cuda")
    sb = torch.empty((b, 2), dtype=torch.int32, device="cuda")
    fa, ma, _ = net.run_batch(dev, KEYS, want_spike_matrix=True, stats_out=sa, longest_first=False)
    fb, mb, _ = net.run_batch(dev, KEYS, want_spike_matrix=True, stats_out=sb, longest_first=True)
    torch.cuda.synchronize()
    assert torch.equal(fa, fb) and torch.equal(ma, mb) and torch.equal(sa, sb), (case, n, k, c, t, b, kernel)
    if case % 3 == 0:
        ref = cport.lif_run_batch(res, rasters, KEYS, n_threads=16)
        assert np.array_equal(fb.cpu().numpy(), ref), (case, "oracle")
    ok += 1
    print(f"case {case}: N={n} k={k} C={c} T={t} B={b} kernel {net.kernel_in_use()} ok "
          f"(spikes per clip {int(sb[:, 1].min())}..{int(sb[:, 1].max())})", flush=True)
print(f"{ok} of {n_cases} cases equal ({time.time() - t_start:.0f} s)")
