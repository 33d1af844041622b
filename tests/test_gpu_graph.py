"""include/lsm_hip.h promises that the launch functions neither allocate nor synchronise and, after a kernel's
first use, make no runtime call but the launch (hipFuncSetAttribute used to run on every
launch).  Proof: front end + reservoir are captured into a hipGraph and replayed on new inputs, bit-exact
against the eager path; two threads launch the same kernels with different LDS sizes at once."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ['spike_counts', 'spike_variances', 'mean_spike_times', 'mean_isi', 'isi_variances']


def _setup(n, k, n_out, filters):
    import torch
    from lsm_speech_classifier_amd import frontend, reservoir as R, snn, synth
    from oracle import ref_numpy as O
    fe = frontend.SpikeFrontEnd(filters, "gammatone")
    a0 = synth.class_chirps(np.arange(16) % 12, seed=3)
    a1 = synth.class_chirps((np.arange(16) + 5) % 12, seed=8)
    r0 = fe.encode(a0).cpu().numpy()
    p = R.SimulationParams(num_neurons=n, num_output_neurons=n_out, small_world_graph_k=k,
                           mean_weight=O.w_critico(k, 2.0, 2, r0) * 0.6)
    net = snn.SNN(None, reservoir=R.build_reservoir(p, filters))
    return torch, fe, net, torch.from_numpy(a0).cuda(), torch.from_numpy(a1).cuda()


@pytest.mark.parametrize("n,k,n_out,filters,kernel", [(1000, 200, 400, 128, "dense"), (1000, 200, 400, 128, "sparse"),
                                                      (4000, 800, 1600, 128, "ring")])
def test_hot_path_captures_into_a_hip_graph(n, k, n_out, filters, kernel):
    torch, fe, net, a0, a1 = _setup(n, k, n_out, filters)
    net.set_kernel(kernel)
    assert net.kernel_in_use() == kernel
    eager0, _, _ = net.run_batch(fe.encode(a0), KEYS)           # first use of every kernel (attributes are set here)
    eager1, _, _ = net.run_batch(fe.encode(a1), KEYS)
    torch.cuda.synchronize()
    static_in = a0.clone()
    stats = torch.zeros((16, 2), dtype=torch.int32, device="cuda")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):                                   # capture: any allocation/sync/attribute call would fail it
        feats, _, _ = net.run_batch(fe.encode(static_in), KEYS, stats_out=stats)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(feats, eager0)
    static_in.copy_(a1)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(feats, eager1) and not torch.equal(eager0, eager1)
    assert int(stats[:, 1].sum()) >= int(feats[:, :n_out].sum())
    for _ in range(3):                                          # replays are idempotent
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(feats, eager1)


def test_two_threads_launch_with_different_lds_sizes():
    """The dynamic-LDS limit is raised once per kernel to the CU's 160 KB; launches of the same kernel with
    different LDS sizes from two threads on two streams cannot lower each other's limit."""
    import torch
    from lsm_speech_classifier_amd import reservoir as R, snn, synth
    r_small = synth.bernoulli_raster(8, 256, 100, 0.1, seed=1)      # T=100  -> small input-bit image
    r_big = synth.bernoulli_raster(8, 256, 2200, 0.1, seed=2)       # T=2200 x 256 channels -> LDS image > 64 KB
    p = R.SimulationParams(num_neurons=900, num_output_neurons=900, small_world_graph_k=60, mean_weight=0.02)
    net = snn.SNN(None, reservoir=R.build_reservoir(p, 256))
    assert net.layout(8, 2200)["lds_bytes"] > 64 * 1024 > net.layout(8, 100)["lds_bytes"]
    want = {}
    for name, r in (("small", r_small), ("big", r_big)):
        want[name] = net.run_batch(r, KEYS)[0].cpu().numpy()
    errors = []

    def work(name, r):
        try:
            st = torch.cuda.Stream()
            dev = torch.from_numpy(r).cuda()
            with torch.cuda.stream(st):
                for _ in range(20):
                    f, _, _ = net.run_batch(dev, KEYS)
                st.synchronize()
            np.testing.assert_array_equal(f.cpu().numpy(), want[name])
        except Exception as exc:                                    # noqa: BLE001
            errors.append((name, repr(exc)))

    ts = [threading.Thread(target=work, args=a) for a in (("small", r_small), ("big", r_big))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
