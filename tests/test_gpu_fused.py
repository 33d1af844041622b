"""The one-launch gammatone front end (`lsm_gammatone_spikes_f64`: filterbank -> dB -> floor/normalise -> resize
-> hysteresis encoder) against the C oracle, the reference-generated golden fixtures' semantics and the two split
entry points.  Reference: /root/reference/create_dataset.py:49-104 (per-clip body of the loop at :143-157)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THR = [0.70, 0.80, 0.90, 0.95]
GAP = 0.1


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from lsm_speech_classifier_amd import _lib
    _lib.require_gpu()
    return torch


def _oracle_raster(oracle_c, audio, n_filters, nwin=400, hop=160, ncols=98, thr=THR, gap=GAP, time_bins=100):
    from oracle import ref_numpy as O
    coefs = O.gammatone_coefs(16000, n_filters, 50)
    out = []
    for a in audio:
        spec = oracle_c.gammatone_spec(a, coefs, nwin, hop, ncols)
        out.append(oracle_c.encode_hysteresis(oracle_c.normalise_resize(oracle_c.gammatone_db(spec), time_bins),
                                              thr, gap))
    return np.stack(out)


@pytest.mark.parametrize("n_filters", [2, 40, 64, 65, 128, 200, 256])
def test_fused_front_end_matches_oracle_and_split_path(torch_cuda, oracle_c, n_filters):
    """One and two chains per lane, one / two waves per clip, ragged last channel group, silent and flat clips,
    a batch that does not fill its last workgroup."""
    from lsm_speech_classifier_amd import frontend, synth
    audio = np.concatenate([synth.class_chirps([0, 3, 7, 11, 5], seed=21), synth.white_noise(3, seed=5)])
    audio[2] = 0.0                                            # silent clip: flat spectrogram -> zeros (:64-65)
    audio[6] = 0.25                                           # constant clip
    fe = frontend.SpikeFrontEnd(n_filters, "gammatone")
    fused = fe.encode(audio, fused=True)
    split = fe.encode(audio, fused=False)
    assert fused.shape == (len(audio), n_filters, 400) and fused.dtype == torch_cuda.uint8
    assert torch_cuda.equal(fused, split)
    assert torch_cuda.equal(fused, fe.encode(audio, fused=True, low_latency=True))   # one chain per lane
    got = fused.cpu().numpy()
    assert not got[2].any()
    np.testing.assert_array_equal(got, _oracle_raster(oracle_c, audio, n_filters))
    assert got.sum() > 0


def test_fused_redundancy_and_other_threshold_tables(torch_cuda, oracle_c):
    from lsm_speech_classifier_amd import frontend, synth
    audio = synth.class_chirps([1, 4, 9], seed=3)
    for red, thr, gap in ((3, THR, GAP), (2, [0.5], 0.05), (1, [0.3, 0.6, 0.65, 0.7, 0.8, 0.9, 0.95, 0.99], 0.02)):
        fe = frontend.SpikeFrontEnd(96, "gammatone", redundancy=red, thresholds=thr, gap=gap)
        got = fe.encode(audio, fused=True)
        assert torch_cuda.equal(got, fe.encode(audio, fused=False))
        ref = np.repeat(_oracle_raster(oracle_c, audio, 96, thr=thr, gap=gap), red, axis=1)
        np.testing.assert_array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("n_samples,time_bins", [(12000, 100), (16000, 100), (24000, 100), (48000, 100), (13000, 100),
                                                 (8000, 50), (12345, 77), (16000, 98), (16000, 40)])
def test_fused_other_clip_lengths_and_window_overlaps(torch_cuda, oracle_c, n_samples, time_bins):
    """1-4 overlapping windows, hops that are not multiples of 8, a column count equal to the bin count (no
    resize) -- the cases of test_gpu_fuzz.py's split-path test, through the fused entry point."""
    from lsm_speech_classifier_amd import frontend, synth
    rng = np.random.RandomState(n_samples + time_bins)
    base = synth.class_chirps([2, 6, 10], seed=8)
    audio = np.ascontiguousarray(np.resize(base, (3, n_samples)).astype(np.float32))
    audio += 0.01 * rng.randn(3, n_samples).astype(np.float32)
    fe = frontend.SpikeFrontEnd(72, "gammatone", n_samples=n_samples, time_bins=time_bins)
    assert 1 <= (fe.nwin + fe.hop - 1) // fe.hop <= 4
    got = fe.encode(audio, fused=True)
    assert torch_cuda.equal(got, fe.encode(audio, fused=False))
    ref = _oracle_raster(oracle_c, audio, 72, nwin=fe.nwin, hop=fe.hop, ncols=fe.ncols, time_bins=time_bins)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)


def test_fused_with_channel_dependent_first_section_gain(torch_cuda, oracle_c):
    """coef_flags bit 2 (one A0/B0 for every channel: the product b0*x is shared by the two channels of a lane) is a
    property the HOST verifies; a table without it takes the unshared kernel and still equals the oracle on that table."""
    from lsm_speech_classifier_amd import frontend, synth
    audio = synth.class_chirps([3, 8], seed=12)
    fe = frontend.SpikeFrontEnd(128, "gammatone")
    assert fe.coef_flags == 7
    want = fe.encode(audio, fused=True)
    tab = fe.coefs.cpu().numpy().copy()
    tab[:, 0] *= 1.0 + 1e-3 * np.arange(128)                 # channel-dependent A0
    fe.coefs = torch_cuda.from_numpy(tab).cuda()
    fe.coef_flags = frontend.coef_flags(tab)
    assert fe.coef_flags == 3
    got = fe.encode(audio, fused=True)
    assert torch_cuda.equal(got, fe.encode(audio, fused=False)) and not torch_cuda.equal(got, want)
    ref = np.stack([oracle_c.encode_hysteresis(oracle_c.normalise_resize(oracle_c.gammatone_db(
        oracle_c.gammatone_spec(a, tab, 400, 160, 98))), THR, GAP) for a in audio])
    np.testing.assert_array_equal(got.cpu().numpy(), ref)
    fe.coef_flags = 7                                        # lying about the table must change the result: the bit is used
    assert not torch_cuda.equal(fe.encode(audio, fused=True), got)


def test_fused_large_batch_equals_small_batches_and_split_path(torch_cuda):
    """cfg2's launch shape (256 clips x 128 filters: one wave per clip, CU-exclusive placement) and a launch
    that needs several workgroups per CU; batch position must not matter."""
    from lsm_speech_classifier_amd import frontend, synth
    audio = synth.class_chirps(np.arange(1100) % 12, seed=77)
    fe = frontend.SpikeFrontEnd(128, "gammatone")
    dev = torch_cuda.from_numpy(audio).cuda()
    big = fe.encode(dev, fused=True)
    assert torch_cuda.equal(big, fe.encode(dev, fused=False))
    assert torch_cuda.equal(big[:256], fe.encode(dev[:256], fused=True))
    assert torch_cuda.equal(big[1000:1003], fe.encode(dev[1000:1003], fused=True))


def test_fused_argument_errors(torch_cuda):
    import ctypes as C
    from lsm_speech_classifier_amd import _lib, frontend
    lib = _lib.load()
    fe = frontend.SpikeFrontEnd(128, "gammatone")
    audio = torch_cuda.zeros((4, 16000), dtype=torch_cuda.float32, device="cuda")
    raster = torch_cuda.empty((4, 128, 400), dtype=torch_cuda.uint8, device="cuda")
    need = lib.lsm_gammatone_spikes_workspace(4, 128, 98)
    assert need == 4 * 98 * 128 * 8
    ws = torch_cuda.empty((need // 8,), dtype=torch_cuda.float64, device="cuda")
    on, off = frontend.threshold_tables(THR, GAP, np.float64)
    p = lambda t: C.c_void_p(t.data_ptr())
    h = lambda a: C.c_void_p(a.ctypes.data)
    call = lambda **kw: lib.lsm_gammatone_spikes_f64(
        p(audio), kw.get("B", 4), 16000, p(fe.coefs), 128, 400, 160, 98, 100, h(on), h(off), kw.get("n_thr", 4),
        kw.get("red", 1), kw.get("raster", p(raster)), p(ws), kw.get("ws_bytes", need), fe.coef_flags,
        kw.get("flags", 0), None)
    assert call() == 0
    want = raster.clone()
    assert call(flags=1) == 0 and torch_cuda.equal(raster, want)      # the low-latency layout: same raster
    assert call(flags=2) == 0 and torch_cuda.equal(raster, want)      # no LDS reservation: same raster
    assert call(flags=3) == 0 and torch_cuda.equal(raster, want)
    assert call(flags=4) == -1
    assert call(B=0) == 0
    assert call(ws_bytes=need - 8) == -1 and b"workspace" in lib.lsm_last_error()
    assert call(n_thr=9) == -1
    assert call(red=0) == -1
    assert call(raster=None) == -1
    torch_cuda.cuda.synchronize()


@pytest.mark.parametrize("n_filters", [640, 1024])
def test_low_latency_flag_is_only_a_hint_above_512_filters(torch_cuda, oracle_c, n_filters):
    """513..1024 filters need more than 8 one-chain waves per clip; the low-latency flag then keeps the
    two-chain layout instead of failing, directly and as HotPath's first step after a synchronisation."""
    from lsm_speech_classifier_amd import frontend, reservoir, snn, synth
    from lsm_speech_classifier_amd.pipeline import HotPath
    audio = synth.class_chirps([0, 5], seed=13)
    fe = frontend.SpikeFrontEnd(n_filters, "gammatone")
    plain = fe.encode(audio, fused=True)
    assert torch_cuda.equal(plain, fe.encode(audio, fused=True, low_latency=True))
    np.testing.assert_array_equal(plain[:1].cpu().numpy(), _oracle_raster(oracle_c, audio[:1], n_filters))
    params = reservoir.SimulationParams(num_neurons=256, num_output_neurons=32, small_world_graph_k=16,
                                        mean_weight=0.02)
    net = snn.SNN(params, n_channels=n_filters)
    hp = HotPath(fe, net, None)                      # default topology: its first front end meets an idle GPU
    dev_audio = torch_cuda.from_numpy(audio).cuda()
    feats, st = hp.submit(dev_audio)
    hp.synchronize()
    ref, _, _ = net.run_batch(plain)
    torch_cuda.cuda.synchronize()
    assert torch_cuda.equal(feats, ref)


def test_negative_hysteresis_gap_clears_an_active_latch_like_the_reference(torch_cuda, oracle_c):
    """with off > on (a negative gap) an active latch can see S and R together; the reference takes
    rising and falling from the latch before the update (create_dataset.py:90-95), so the latch is cleared."""
    from lsm_speech_classifier_amd import frontend, synth
    from oracle import ref_numpy as O
    rng = np.random.RandomState(4)
    audio = np.concatenate([synth.class_chirps([1, 8, 10], seed=2), synth.white_noise(2, seed=9)])
    for gap in (-0.05, -0.2):
        thr = sorted(rng.uniform(0.3, 0.95, size=4).tolist())
        fe = frontend.SpikeFrontEnd(96, "gammatone", thresholds=thr, gap=gap)
        got = fe.encode(audio, fused=True)
        assert torch_cuda.equal(got, fe.encode(audio, fused=False))
        assert torch_cuda.equal(got, fe.encode(audio, fused=True, low_latency=True))
        ref = _oracle_raster(oracle_c, audio, 96, thr=thr, gap=gap)
        np.testing.assert_array_equal(got.cpu().numpy(), ref)
        # and the literal NumPy restatement of the reference's loop agrees with the C port on this case
        coefs = O.gammatone_coefs(16000, 96, 50)
        norm = oracle_c.normalise_resize(oracle_c.gammatone_db(oracle_c.gammatone_spec(audio[0], coefs, 400, 160, 98)), 100)
        np.testing.assert_array_equal(O.encode_hysteresis(norm, thr, gap), ref[0])


def test_split_route_refuses_caller_owned_buffers(torch_cuda, monkeypatch):
    """raster_out / workspace belong to the fused launch; the split route (here forced) must not drop them
    silently -- pipeline.HotPath relies on `will_fuse()` to decide who owns the raster buffer."""
    from lsm_speech_classifier_amd import frontend, reservoir, snn, synth
    from lsm_speech_classifier_amd.pipeline import HotPath
    audio = torch_cuda.from_numpy(synth.class_chirps([0, 5, 7], seed=1)).cuda()
    fe = frontend.SpikeFrontEnd(64, "gammatone")
    want = fe.encode(audio)
    buf = torch_cuda.empty((3, 64, 400), dtype=torch_cuda.uint8, device="cuda")
    monkeypatch.setenv("LSM_FRONTEND_SPLIT", "1")
    assert not fe.will_fuse()
    with pytest.raises(ValueError):
        fe.encode(audio, raster_out=buf)
    # the pipeline on the split route: rasters come from the allocator and are handed over with record_stream
    params = reservoir.SimulationParams(num_neurons=256, num_output_neurons=32, small_world_graph_k=16,
                                        mean_weight=0.02)
    net = snn.SNN(params, n_channels=64)
    hp = HotPath(fe, net, None)
    rows = [hp.submit(audio)[0] for _ in range(8)]
    hp.synchronize()
    assert not hp._rasters                              # no pipeline-owned raster ring was touched
    ref, _, _ = net.run_batch(want)
    torch_cuda.cuda.synchronize()
    for r in rows:
        assert torch_cuda.equal(r, ref)
