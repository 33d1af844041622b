"""lsm_reservoir_run_ordered: the clips of a batch started longest first (include/lsm_hip.h).

The order is a launch schedule, never a result: every clip is still what
/root/reference/extract_lsm_features.py:78-87 computes for it on its own (reset, set_input_spike_times, simulate,
extract_features_from_spikes), at its own row.  Checked here: the ordered launch equals the plain one and the C
oracle bit for bit on all three reservoir kernels, with clips of very different activity, ties and silent clips;
the spike counts and the order the library derives are the ones NumPy derives; argument errors.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KEYS = ['spike_counts', 'spike_variances', 'mean_spike_times', 'mean_isi', 'isi_variances']


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from lsm_speech_classifier_amd import _lib
    _lib.require_gpu()
    return torch


def _uneven_rasters(n, c, t, seed):
    """Clips whose input density runs from silent to 0.45, in shuffled order, with exact ties."""
    rs = np.random.RandomState(seed)
    dens = np.linspace(0.0, 0.45, n)
    rs.shuffle(dens)
    r = (rs.random_sample((n, c, t)) < dens[:, None, None]).astype(np.uint8)
    r[n // 3] = r[n // 3 + 1]                  # a tie: equal spike counts, different clips
    r[5] = 0                                   # a silent clip
    return r


def _net(n_neurons, k, c, rasters, mult=1.2):
    from lsm_speech_classifier_amd import reservoir as R, snn
    from oracle import ref_numpy as O
    wc = O.w_critico(k, 2.0, 2, rasters)
    res = R.build_reservoir(R.SimulationParams(num_neurons=n_neurons, num_output_neurons=n_neurons // 2,
                                               small_world_graph_k=k, mean_weight=wc * mult), c)
    return res, snn.SNN(None, reservoir=res)


@pytest.mark.parametrize("kernel", ["dense", "ring", "sparse"])
def test_ordered_launch_equals_plain_launch_and_oracle(torch_cuda, oracle_c, kernel):
    torch = torch_cuda
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    n, c, t = cus + 37, 24, 60                                     # more clips than compute units: the order applies
    rasters = _uneven_rasters(n, c, t, seed=5)
    res, net = _net(1024, 200, c, rasters)
    net.set_kernel(kernel)
    dev = torch.from_numpy(rasters).cuda()
    st_a = torch.empty((n, 2), dtype=torch.int32, device="cuda")
    st_b = torch.empty((n, 2), dtype=torch.int32, device="cuda")
    f_plain, sm_plain, vt_plain = net.run_batch(dev, KEYS, want_spike_matrix=True, want_v_trace=True, stats_out=st_a,
                                                longest_first=False)
    f_ord, sm_ord, vt_ord = net.run_batch(dev, KEYS, want_spike_matrix=True, want_v_trace=True, stats_out=st_b,
                                          longest_first=True)
    assert torch.equal(f_plain, f_ord) and torch.equal(sm_plain, sm_ord) and torch.equal(st_a, st_b)
    assert torch.equal(vt_plain.view(torch.int32), vt_ord.view(torch.int32))
    f_default, _, _ = net.run_batch(dev, KEYS)                       # default: ordered, since n > compute units
    assert torch.equal(f_default, f_plain)
    ref = oracle_c.lif_run_batch(res, rasters, KEYS, n_threads=8)
    np.testing.assert_array_equal(f_ord.cpu().numpy(), ref)
    tot = st_b[:, 1].cpu().numpy()
    assert ref.sum() > 0 and tot.min() == 0 and tot.max() >= 2 * max(1, int(np.median(tot)))     # uneven clips indeed


def test_keys_and_order_are_numpys(torch_cuda):
    """The workspace after the call: input spike count of every clip, then the start order -- descending count, ties
    by clip index (a stable sort), ranked inside windows of 4096 clips."""
    torch = torch_cuda
    from lsm_speech_classifier_amd import _lib
    lib = _lib.load()
    n, c, t = 4096 + 700, 8, 37                                    # two ranking windows; 296 bytes per clip (not a multiple of 16)
    rs = np.random.RandomState(3)
    rasters = (rs.random_sample((n, c, t)) < rs.random_sample((n, 1, 1)) * 0.5).astype(np.uint8)
    rasters[17] = 0
    rasters[40] = rasters[41]
    res, net = _net(256, 40, c, rasters[:64])
    dev = torch.from_numpy(rasters).cuda()
    need = lib.lsm_reservoir_order_workspace(n)
    assert need == 8 * n and lib.lsm_reservoir_order_workspace(0) == 0
    ws = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
    feats = torch.empty((n, net.num_output_neurons), dtype=torch.float32, device="cuda")
    kid = np.array([0], dtype=np.int32)
    _lib.check(lib.lsm_reservoir_run_ordered(net._handle, dev.data_ptr(), n, t, kid.ctypes.data, 1, feats.data_ptr(),
                                             None, None, None, 0, ws.data_ptr(), need,
                                             torch.cuda.current_stream().cuda_stream), "run_ordered")
    torch.cuda.synchronize()
    w = ws.cpu().numpy()
    keys = rasters.reshape(n, -1).sum(axis=1).astype(np.int32)
    np.testing.assert_array_equal(w[:n], keys)
    want = np.concatenate([first + np.argsort(-keys[first:first + 4096].astype(np.int64), kind="stable")
                           for first in range(0, n, 4096)]).astype(np.int32)
    np.testing.assert_array_equal(w[n:], want)
    plain, _, _ = net.run_batch(dev, ['spike_counts'], longest_first=False)
    assert torch.equal(plain, feats)


def test_small_batches_are_launched_as_they_are_and_errors(torch_cuda):
    torch = torch_cuda
    from lsm_speech_classifier_amd import _lib
    lib = _lib.load()
    c, t = 8, 40
    rasters = _uneven_rasters(12, c, t, seed=9)
    res, net = _net(256, 40, c, rasters)
    dev = torch.from_numpy(rasters).cuda()
    ws = torch.full((24,), -7, dtype=torch.int32, device="cuda")
    feats = torch.empty((12, net.num_output_neurons), dtype=torch.float32, device="cuda")
    kid = np.array([0], dtype=np.int32)
    args = (net._handle, dev.data_ptr(), 12, t, kid.ctypes.data, 1, feats.data_ptr(), None, None, None, 0)
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.lsm_reservoir_run_ordered(*args, ws.data_ptr(), 96, stream), "run_ordered")
    torch.cuda.synchronize()
    assert (ws.cpu().numpy() == -7).all()                            # one round: no ranking launches, workspace untouched
    plain, _, _ = net.run_batch(dev, ['spike_counts'], longest_first=False)
    assert torch.equal(plain, feats)
    assert lib.lsm_reservoir_run_ordered(*args, ws.data_ptr(), 95, stream) != 0
    assert b"workspace" in lib.lsm_last_error()
    assert lib.lsm_reservoir_run_ordered(*args, None, 96, stream) != 0
    assert b"workspace" in lib.lsm_last_error()
    # an empty batch is fine without a workspace, like lsm_reservoir_run
    assert lib.lsm_reservoir_run_ordered(net._handle, None, 0, t, kid.ctypes.data, 1, None, None, None, None, 0,
                                         None, 0, stream) == 0


def test_ordered_launch_captures_into_a_hip_graph(torch_cuda):
    """Counting, ranking and the LIF launch are three launches on the caller's stream and nothing else (no allocation,
    no synchronisation): captured once, replayed on another batch, equal to the eager launches."""
    torch = torch_cuda
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    n, c, t = cus + 11, 16, 48
    r0, r1 = _uneven_rasters(n, c, t, seed=21), _uneven_rasters(n, c, t, seed=22)
    res, net = _net(512, 100, c, r0)
    d0, d1 = torch.from_numpy(r0).cuda(), torch.from_numpy(r1).cuda()
    e0, _, _ = net.run_batch(d0, KEYS, longest_first=True)           # first use of every kernel
    e1, _, _ = net.run_batch(d1, KEYS, longest_first=False)
    torch.cuda.synchronize()
    static_in = d0.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        feats, _, _ = net.run_batch(static_in, KEYS, longest_first=True)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(feats, e0)
    static_in.copy_(d1)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(feats, e1) and not torch.equal(e0, e1)


@pytest.mark.parametrize("n,k,want_wpc", [(3072, 614, 4), (1536, 100, 2)])
def test_ring_rows_with_three_quads_per_wave(torch_cuda, oracle_c, n, k, want_wpc):
    """Reservoirs whose quad count is a multiple of three (N = 3072: 12 quads) had no strided ring layout:
    with three quads per wave they do (4 waves x 3 quads; the window's <= 4 quads fall into different waves).  Bit-exact
    against the oracle, spike matrix and membrane trace included, and against the contiguous layouts and the dense rows."""
    torch = torch_cuda
    from lsm_speech_classifier_amd import reservoir as R, snn, synth
    from oracle import ref_numpy as O
    c, t, b = 32, 50, 6
    rasters = synth.bernoulli_raster(b, c, t, 0.25, seed=n)
    wc = O.w_critico(k, 2.0, 2, rasters)
    res = R.build_reservoir(R.SimulationParams(num_neurons=n, num_output_neurons=n // 3, small_world_graph_k=k,
                                               mean_weight=wc * 1.3), c)
    net = snn.SNN(None, reservoir=res)
    net.set_kernel("ring-quads")           # (round 5: "ring" would take the reservoir's pair-block layout, csrc/lif_pair.h)
    plan = net.plan(b, t, 0)
    assert plan["kernel"] == "ring" and plan["waves_per_clip"] == want_wpc and plan["slots_per_lane"] == 12
    feats, sm, vt = net.run_batch(rasters, KEYS, want_spike_matrix=True, want_v_trace=True)
    for i in range(b):
        f_ref, sm_ref, vt_ref = oracle_c.lif_run(res, rasters[i], KEYS, want_trace=True)
        np.testing.assert_array_equal(feats[i].cpu().numpy(), f_ref)
        np.testing.assert_array_equal(sm[i].cpu().numpy(), sm_ref)
        np.testing.assert_array_equal(vt[i].cpu().numpy().view(np.uint32), vt_ref.view(np.uint32))
    assert sm.sum() > 0
    net.set_kernel("ring-contiguous")
    f_cont, _, _ = net.run_batch(rasters, KEYS)
    net.set_kernel("dense")
    f_dense, _, _ = net.run_batch(rasters, KEYS)
    assert torch.equal(feats, f_cont) and torch.equal(feats, f_dense)
