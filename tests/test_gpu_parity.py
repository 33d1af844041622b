"""HIP path vs the CPU oracle on identical seeded inputs (run on the MI355X box: -m gpu).

Bars: bit-exact for every integer/byte result (rasters, spike matrices, counts) and — because the
kernels repeat the oracle's float operation order with contraction off — also for the float
results (gammatone spectrogram, membrane traces, features).  log10 is the one exception: dB
values are compared at 1e-12 absolute (values are O(100)), and the rasters derived from them
must still be identical.
"""
import os

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


def _mixed_audio(n, seed):
    from lsm_speech_classifier_amd import synth
    a = synth.class_chirps(list(range(n)), seed=seed)
    a[n // 2:] = synth.white_noise(n - n // 2, seed=seed + 1)
    return a


# ----------------------------------------------------------------------------- front end ----
@pytest.mark.parametrize("n_filters", [2, 40, 128])
def test_gammatone_frontend_matches_oracle(torch_cuda, oracle_c, n_filters):
    from lsm_speech_classifier_amd import frontend
    from oracle import ref_numpy as O
    audio = _mixed_audio(4, seed=11 + n_filters)
    fe = frontend.SpikeFrontEnd(n_filters, "gammatone")
    assert (fe.nwin, fe.hop, fe.ncols) == (400, 160, 98)
    coefs = O.gammatone_coefs(16000, n_filters, 50)
    np.testing.assert_array_equal(frontend.gammatone_filter_table(16000, n_filters, 50), coefs)
    db, spec = fe.spectrogram_db(audio, want_spec=True)
    raster, norm = fe.spikes_from_db(db, want_norm=True)
    spec, db, norm, raster = (t.cpu().numpy() for t in (spec, db, norm, raster))
    for b in range(len(audio)):
        s_ref = oracle_c.gammatone_spec(audio[b], coefs, 400, 160, 98)
        np.testing.assert_array_equal(spec[b], s_ref)                       # bit-exact float64
        np.testing.assert_allclose(db[b], 20 * np.log10(s_ref + 1e-9), rtol=0, atol=1e-12)
        n_ref = oracle_c.normalise_resize(oracle_c.gammatone_db(s_ref))
        np.testing.assert_allclose(norm[b], n_ref, rtol=0, atol=1e-13)
        r_ref = oracle_c.encode_hysteresis(n_ref, THR, GAP)
        np.testing.assert_array_equal(raster[b], r_ref)
        assert raster[b].dtype == np.uint8 and raster[b].shape == (n_filters, 400)


@pytest.mark.parametrize("n_filters,n_clips", [(40, 3), (40, 5), (130, 3), (64, 1)])
def test_gammatone_partial_workgroups(torch_cuda, oracle_c, n_filters, n_clips):
    """Wave counts that do not fill the last 4-wave workgroup, and a channel count whose last wave has
    only two live lanes: every (clip, channel) still equals the oracle bit for bit."""
    from lsm_speech_classifier_amd import frontend
    from oracle import ref_numpy as O
    audio = _mixed_audio(n_clips, seed=3 * n_filters + n_clips)
    fe = frontend.SpikeFrontEnd(n_filters, "gammatone")
    coefs = O.gammatone_coefs(16000, n_filters, 50)
    _, spec = fe.spectrogram_db(audio, want_spec=True)
    spec = spec.cpu().numpy()
    assert spec.shape == (n_clips, n_filters, 98)
    for b in range(n_clips):
        np.testing.assert_array_equal(spec[b], oracle_c.gammatone_spec(audio[b], coefs, 400, 160, 98))


def test_gammatone_special_sample_values(torch_cuda, oracle_c):
    """Exact zeros of both signs, float32 denormals (smallest, largest, in between), the smallest and largest
    normal magnitudes and ordinary values as audio samples must give the oracle's bits (added with an experiment
    that converted the wave-uniform sample to float64 with scalar integer instructions -- bit-exact, but 1.32 ->
    2.25 ms per launch, DESIGN.md section 9 -- and kept: padded clips are full of exact zeros)."""
    from lsm_speech_classifier_amd import frontend
    from oracle import ref_numpy as O
    rs = np.random.RandomState(321)
    special = np.array([0.0, -0.0, 1e-45, -1e-45, 1.1754942e-38, -1.1754942e-38, 3e-42, 7e-40, 1.17549435e-38,
                        -1.17549435e-38, 3.0e38, -3.0e38, 1.0, -1.0, 0.5, 1.5, 0.1, 123456.789, 2.0 ** -126,
                        2.0 ** -127, 2.0 ** -149, 2.0 ** 127], dtype=np.float32)
    audio = (rs.randn(3, 16000) * 0.1).astype(np.float32)
    audio[0, rs.randint(0, 16000, size=4000)] = special[rs.randint(0, len(special), size=4000)]
    audio[1, :] = special[rs.randint(0, 12, size=16000)]                 # zeros and denormals only
    audio[2, 2000:9000] = 0.0                                            # a padded clip
    bits = audio.view(np.uint32)
    assert ((bits >> 23) & 0xFF == 0).sum() > 1000                       # zeros and denormals are really there
    fe = frontend.SpikeFrontEnd(64, "gammatone")
    coefs = O.gammatone_coefs(16000, 64, 50)
    _, spec = fe.spectrogram_db(audio, want_spec=True)
    spec = spec.cpu().numpy()
    for b in range(3):
        np.testing.assert_array_equal(spec[b], oracle_c.gammatone_spec(audio[b], coefs, 400, 160, 98))


def test_gammatone_large_launch_equals_small_launches(torch_cuda):
    """More workgroups than CUs (no CU-exclusive LDS reservation) must give the same bits as the small
    launches that carry the reservation: 600 clips tiled from 5 distinct ones."""
    import torch
    from lsm_speech_classifier_amd import frontend
    base = _mixed_audio(5, seed=77)
    fe = frontend.SpikeFrontEnd(128, "gammatone")
    db_small, spec_small = fe.spectrogram_db(base, want_spec=True)
    reps = 120
    big = np.tile(base, (reps, 1))
    db_big, spec_big = fe.spectrogram_db(big, want_spec=True)
    assert spec_big.shape == (5 * reps, 128, 98)
    assert torch.equal(spec_big.view(reps, 5, 128, 98), spec_small.unsqueeze(0).expand(reps, -1, -1, -1))
    assert torch.equal(db_big.view(reps, 5, 128, 98), db_small.unsqueeze(0).expand(reps, -1, -1, -1))
    raster_big = fe.encode(big)
    raster_small = fe.encode(base)
    assert torch.equal(raster_big.view(reps, 5, 128, 400), raster_small.unsqueeze(0).expand(reps, -1, -1, -1))


def test_frontend_against_numpy_scipy_restatement(torch_cuda):
    """Same path against the literal NumPy/SciPy restatement (scipy.signal.lfilter inside)."""
    from lsm_speech_classifier_amd import frontend
    from oracle import ref_numpy as O
    audio = _mixed_audio(2, seed=5)
    fe = frontend.SpikeFrontEnd(64, "gammatone")
    db, spec = fe.spectrogram_db(audio, want_spec=True)
    raster, _ = fe.spikes_from_db(db)
    for b in range(2):
        g = O.gtgram(audio[b], 16000, 0.025, 0.01, 64, 50)
        np.testing.assert_array_equal(spec[b].cpu().numpy(), g)
        r = O.encode_hysteresis(O.normalise_resize(O.gammatone_db(g)), THR, GAP)
        np.testing.assert_array_equal(raster[b].cpu().numpy(), r)


@pytest.mark.parametrize("gap", [0.05, 0.0, -0.2])
@pytest.mark.parametrize("n_filters,ncols,time_bins,n_thr,redundancy", [(5, 98, 100, 4, 1), (70, 101, 100, 4, 2), (3, 130, 130, 3, 1),
                                                                         (130, 33, 70, 8, 1), (2, 64, 31, 1, 3)])
def test_spikes_from_db_over_shapes_gaps_and_thresholds(torch_cuda, gap, n_filters, ncols, time_bins, n_thr, redundancy):
    """lsm_spec_to_spikes_* (csrc/spikes_body.h: comparison bits by ballot, the latch of 32 time bins as one addition) against
    the restated reference: row groups past 64 filters, rows of more and of less than 64 bins, no resize, up to 8 thresholds,
    redundancy, and a NEGATIVE gap -- an off-bound above its on-bound, where a value between the two flips the latch
    (create_dataset.py:88-96 takes rising and falling from the latch before the update)."""
    import torch
    from lsm_speech_classifier_amd import frontend
    from oracle import ref_numpy as O
    rng = np.random.default_rng(100 * n_filters + ncols + n_thr)
    thr = sorted(rng.uniform(0.15, 0.95, n_thr).tolist(), reverse=True)
    db = (rng.standard_normal((3, n_filters, ncols)).cumsum(axis=2) * 4.0 - 30.0)
    db[2, :, ncols // 2:] = db[2, :, ncols // 2:][:, ::-1] * 0.5            # more threshold crossings
    fe = frontend.SpikeFrontEnd(n_filters, "gammatone", redundancy=redundancy, thresholds=thr, gap=gap, time_bins=time_bins)
    raster, norm = fe.spikes_from_db(torch.from_numpy(db).cuda(), want_norm=True)
    raster, norm = raster.cpu().numpy(), norm.cpu().numpy()
    assert raster.shape == (3, n_filters * redundancy, time_bins * n_thr)
    for b in range(3):
        n_ref = O.normalise_resize(np.maximum(db[b], db[b].max() - 80.0), time_bins)      # create_dataset.py:60-78
        np.testing.assert_allclose(norm[b], n_ref, rtol=0, atol=1e-12)
        r_ref = np.repeat(O.encode_hysteresis(norm[b], thr, gap), redundancy, axis=0)     # create_dataset.py:101-104
        np.testing.assert_array_equal(raster[b], r_ref)


def test_encoder_matches_reference_golden(torch_cuda, golden_dir):
    from lsm_speech_classifier_amd import frontend
    g = np.load(os.path.join(golden_dir, "encoder.npz"))
    names = sorted(k[:-3] for k in g.files if k.endswith("_in") and not k.startswith("gap005"))
    for n in names:
        out = frontend.convert_spectrogram_to_spikes_hysteresis(g[n + "_in"], list(g["thresholds"]),
                                                                float(g["gap"]))
        np.testing.assert_array_equal(out, g[n + "_out"], err_msg=n)
    out = frontend.convert_spectrogram_to_spikes_hysteresis(g["gap005_float64_in"], [0.5, 0.9, 0.3])
    np.testing.assert_array_equal(out, g["gap005_float64_out"])


def test_postfilter_matches_reference_golden(torch_cuda, golden_dir):
    """create_dataset.py:59-78 fixtures: normalise + zoom + crop on the GPU."""
    import torch
    from lsm_speech_classifier_amd import frontend
    g = np.load(os.path.join(golden_dir, "postfilter.npz"))
    for n in ("gt_a", "gt_b", "gt_c", "gt_flat"):
        spec = g[n + "_in"]
        fe = frontend.SpikeFrontEnd(spec.shape[0], "gammatone")
        db = torch.from_numpy(20 * np.log10(spec + 1e-9))[None].cuda()
        raster, norm = fe.spikes_from_db(db, want_norm=True)
        ref = g[n + "_out"].astype(np.float64)
        np.testing.assert_allclose(norm[0].cpu().numpy(), ref, rtol=0, atol=1e-13, err_msg=n)
        from oracle import ref_numpy as O
        np.testing.assert_array_equal(raster[0].cpu().numpy(), O.encode_hysteresis(ref, THR, GAP))
    for n in ("mel_a", "mel_b"):
        db = g[n + "_in"]
        fe = frontend.SpikeFrontEnd.__new__(frontend.SpikeFrontEnd)       # shape-only instance
        fe.__dict__.update(lib=frontend._lib.load(), device=torch.device("cuda"),
                           n_filters=db.shape[0], filterbank="mel", redundancy=1, thresholds=THR,
                           gap=GAP, time_bins=100)
        raster, norm = fe.spikes_from_db(torch.from_numpy(db)[None].cuda(), want_norm=True)
        np.testing.assert_array_equal(norm[0].cpu().numpy(), g[n + "_out"], err_msg=n)   # float32 bit-exact


def test_postfilter_propagates_nan_like_the_reference(torch_cuda, golden_dir):
    """a NaN (or an Inf) among a clip's dB values.  The reference's `spec_db.max()` / `.min()` /
    `np.maximum` (create_dataset.py:59-67) propagate NaN: the whole clip normalises to NaN and its raster is all zeros
    (tests/golden/postfilter_nonfinite.npz holds the reference's own outputs).  The kernels' comparisons used to skip the
    NaN and normalise the rest of the clip."""
    import torch
    from lsm_speech_classifier_amd import frontend
    g = np.load(os.path.join(golden_dir, "postfilter_nonfinite.npz"))
    with np.errstate(all="ignore"):
        for n in ("gt_nan", "gt_nan_first", "gt_nan_last", "gt_inf"):
            spec = g[n + "_in"]
            fe = frontend.SpikeFrontEnd(spec.shape[0], "gammatone")
            db = torch.from_numpy(20 * np.log10(spec + 1e-9))[None].cuda()
            raster, norm = fe.spikes_from_db(db, want_norm=True)
            np.testing.assert_array_equal(norm[0].cpu().numpy(), g[n + "_out"], err_msg=n)         # NaN everywhere
            np.testing.assert_array_equal(raster[0].cpu().numpy(), g[n + "_raster"], err_msg=n)    # no spike
        for n in ("mel_nan", "mel_nan_first"):
            db = g[n + "_in"]
            fe = frontend.SpikeFrontEnd.__new__(frontend.SpikeFrontEnd)       # shape-only instance
            fe.__dict__.update(lib=frontend._lib.load(), device=torch.device("cuda"),
                               n_filters=db.shape[0], filterbank="mel", redundancy=1, thresholds=THR,
                               gap=GAP, time_bins=100)
            raster, norm = fe.spikes_from_db(torch.from_numpy(db)[None].cuda(), want_norm=True)
            np.testing.assert_array_equal(norm[0].cpu().numpy(), g[n + "_out"], err_msg=n)
            np.testing.assert_array_equal(raster[0].cpu().numpy(), g[n + "_raster"], err_msg=n)


@pytest.mark.parametrize("filterbank", ["gammatone", "mel"])
def test_audio_with_nan_or_inf_samples_gives_the_reference_raster(torch_cuda, oracle_c, filterbank):
    """A NaN or +-Inf SAMPLE: the filterbank turns it into NaN columns, the reference's max / min / maximum spread them
    over the clip, and the clip's raster is all zeros (create_dataset.py:49-67 through oracle/ref_numpy.py, whose NumPy
    calls are the reference's); the neighbouring clips of the batch are untouched.  Every route: one launch, split
    launches, both fused layouts."""
    from lsm_speech_classifier_amd import frontend
    from oracle import ref_numpy as O
    audio = _mixed_audio(5, seed=11)
    audio[1, 7000] = np.nan
    audio[2, 0] = np.inf
    audio[3, 15999] = -np.inf
    F = 32
    fe = frontend.SpikeFrontEnd(F, filterbank)
    with np.errstate(all="ignore"):
        if filterbank == "gammatone":
            coefs = O.gammatone_coefs(16000, F, 50)
            ref = np.stack([O.encode_hysteresis(O.normalise_resize(O.gammatone_db(
                O.gtgram(a, 16000, 0.025, 0.01, F, 50))), THR, GAP) for a in audio])
            ref_c = np.stack([oracle_c.encode_hysteresis(oracle_c.normalise_resize(oracle_c.gammatone_db(
                oracle_c.gammatone_spec(a, coefs, fe.nwin, fe.hop, fe.ncols))), THR, GAP) for a in audio])
            np.testing.assert_array_equal(ref_c, ref)
        else:
            ref = np.stack([O.encode_hysteresis(O.normalise_resize(O.mel_db(a, F)), THR, GAP) for a in audio])
    assert not ref[1].any() and not ref[2].any() and ref[0].any() and ref[4].any()
    routes = [dict(fused=False), dict(fused=True)]
    if filterbank == "gammatone":
        routes.append(dict(fused=True, low_latency=True))
    for kw in routes:
        got = fe.encode(audio, **kw).cpu().numpy()
        np.testing.assert_array_equal(got, ref, err_msg=str(kw))


def test_redundancy_rows(torch_cuda):
    from lsm_speech_classifier_amd import frontend
    audio = _mixed_audio(2, seed=3)
    r1 = frontend.SpikeFrontEnd(16, "gammatone", redundancy=1).encode(audio).cpu().numpy()
    r3 = frontend.SpikeFrontEnd(16, "gammatone", redundancy=3).encode(audio).cpu().numpy()
    np.testing.assert_array_equal(r3, np.repeat(r1, 3, axis=1))


def test_flat_and_silent_audio(torch_cuda):
    from lsm_speech_classifier_amd import frontend
    out = frontend.audio_to_spectrogram(np.zeros(16000, dtype=np.float32), 8, "gammatone")
    assert out.shape == (8, 100) and out.dtype == np.float32 and not out.any()
    fe = frontend.SpikeFrontEnd(8, "gammatone")
    assert int(fe.encode(np.zeros((1, 16000), dtype=np.float32)).sum()) == 0


# ----------------------------------------------------------------------------- reservoir ----
def _reservoir(n, k, n_out, c, rasters, **kw):
    from lsm_speech_classifier_amd import reservoir as R
    from oracle import ref_numpy as O
    wc = O.w_critico(k, 2.0, kw.get("refractory_period", 2), rasters)
    p = R.SimulationParams(num_neurons=n, num_output_neurons=n_out, small_world_graph_k=k,
                           mean_weight=wc * kw.pop("multiplier", 0.6), **kw)
    return R.build_reservoir(p, c)


def _kernels(net):
    """The reservoir kernels this reservoir offers: dense rows, sparse CSC, and ring rows when it is ring-like
    and has at least a few quads of 256 neurons."""
    from lsm_speech_classifier_amd import _lib
    out = ["dense", "sparse"]
    try:
        net.set_kernel("ring")
        # "ring": what the library prefers (pair blocks where the reservoir has them); "ring-quads" / "ring-contiguous":
        # both quad ownerships of csrc/lif_ring.h (the second: contiguous layouts only)
        out[1:1] = ["ring", "ring-quads", "ring-contiguous"]
    except _lib.LsmHipError:
        pass
    net.set_kernel("auto")
    return out


def _check_against_oracle(net, rasters, oracle_c, wpc, keys=None):
    from lsm_speech_classifier_amd import _lib
    import torch
    stats = torch.full((len(rasters), 2), -1, dtype=torch.int32, device="cuda")
    try:
        feats, sm, vt = net.run_batch(rasters, keys, want_spike_matrix=True, want_v_trace=True,
                                      waves_per_clip=wpc, stats_out=stats)
    except _lib.LsmHipError as e:
        if "no ring-row layout" in str(e):                   # the ring kernel has 1-3 layouts per reservoir
            return 1
        raise
    feats, sm, vt = feats.cpu().numpy(), sm.cpu().numpy(), vt.cpu().numpy()
    total = 0
    for b in range(len(rasters)):
        f_ref, sm_ref, vt_ref = oracle_c.lif_run(net.reservoir, rasters[b], keys, want_trace=True)
        np.testing.assert_array_equal(sm[b], sm_ref, err_msg=f"spike matrix clip {b} wpc {wpc}")
        np.testing.assert_array_equal(vt[b], vt_ref, err_msg=f"membrane trace clip {b} wpc {wpc}")
        np.testing.assert_array_equal(feats[b], f_ref, err_msg=f"features clip {b} wpc {wpc}")
        per = sm_ref.sum(axis=0)                                   # the in-kernel health statistics (§8f-3)
        assert stats[b].tolist() == [int(np.count_nonzero(per)), int(per.sum())], f"stats clip {b} wpc {wpc}"
        total += int(sm_ref.sum())
    return total


@pytest.mark.parametrize("n,k,n_out,c,wpcs", [
    (200, 40, 80, 32, (1, 2, 4)),
    (500, 100, 200, 40, (1, 2, 4, 8)),
    (1000, 200, 400, 128, (1, 2, 4, 8, 16)),
    # input-drive modes of the dense kernel: channel masks (C <= 128 and <= 4 neurons per lane) with a
    # partly filled last mask word / fewer than 4 words, and the entry-list modes for C > 128
    (256, 40, 100, 100, (1, 2, 4)),
    (192, 30, 64, 33, (1, 2)),
    (300, 60, 120, 200, (2, 4, 8)),
    # ring rows: 6 whole quads / a partly filled last quad with C > 128 / 9 quads, 4 neurons of the 10th
    (1536, 150, 600, 64, (2, 4, 8)),
    (2000, 400, 800, 200, (2, 4, 8)),
    (2308, 300, 700, 48, (4, 8, 16)),
])
def test_reservoir_matches_oracle_all_layouts(torch_cuda, oracle_c, n, k, n_out, c, wpcs):
    from lsm_speech_classifier_amd import snn, synth
    rasters = synth.bernoulli_raster(3, c, 400, 0.2, seed=n)
    rasters[2] = synth.bernoulli_raster(1, c, 400, 0.05, seed=n + 1)[0]
    res = _reservoir(n, k, n_out, c, rasters)
    net = snn.SNN(None, reservoir=res)
    kernels = _kernels(net)
    assert ("ring" in kernels) == (n >= 1000)            # small-world reservoirs are ring-like; ring rows need a few quads
    for kernel in kernels:                               # every kernel, every layout
        net.set_kernel(kernel)
        for wpc in wpcs:
            total = _check_against_oracle(net, rasters, oracle_c, wpc)
            assert total > 0, "test input must make the reservoir spike"


def test_reservoir_edge_cases(torch_cuda, oracle_c):
    from lsm_speech_classifier_amd import snn, synth
    c = 24
    base = synth.bernoulli_raster(2, c, 96, 0.3, seed=9)
    cases = {
        "silent": np.zeros((1, c, 96), dtype=np.uint8),
        "saturated": np.ones((1, c, 96), dtype=np.uint8),
        "odd_T": synth.bernoulli_raster(2, c, 37, 0.4, seed=10),          # T % 4 != 0 path
        "byte_values": (base * 7).astype(np.uint8),                        # any non-zero byte = spike
        "single_step": synth.bernoulli_raster(1, c, 1, 0.9, seed=12),
    }
    for mult, refr, div in ((0.6, 2, None), (3.0, 0, None), (1.5, 5, 4.0)):
        res = _reservoir(130, 20, 130, c, base, multiplier=mult, refractory_period=refr,
                         leak_variance_divisor=div)
        net = snn.SNN(None, reservoir=res)
        for kernel in _kernels(net):
            net.set_kernel(kernel)
            for name, r in cases.items():
                for wpc in (1, 2):
                    _check_against_oracle(net, r, oracle_c, wpc)
        net.set_kernel("auto")
    silent_feats, sm, _ = net.run_batch(cases["silent"], want_spike_matrix=True)
    assert int(sm.sum()) == 0 and not silent_feats.cpu().numpy().any()
    # empty batch is a no-op
    feats, _, _ = net.run_batch(np.zeros((0, c, 96), dtype=np.uint8))
    assert feats.shape[0] == 0


def test_feature_sets_and_snn_protocol(torch_cuda, oracle_c, golden_dir):
    """Key subsets/order (FEATURE_SETS, extract_lsm_features.py:19-28) and the single-clip
    reset/set_input/simulate/extract protocol (extract_lsm_features.py:79-87)."""
    from lsm_speech_classifier_amd import snn, synth
    from oracle import ref_numpy as O
    g = np.load(os.path.join(golden_dir, "constants.npz"))
    c = 32
    rasters = synth.bernoulli_raster(2, c, 400, 0.25, seed=21)
    res = _reservoir(256, 50, 100, c, rasters)
    net = snn.SNN(None, reservoir=res)
    for name in g["feature_set_names"]:
        keys = [str(k) for k in g[f"feature_set_{name}"]]
        _check_against_oracle(net, rasters, oracle_c, 0, keys)
    net.reset()
    net.set_input_spike_times(rasters[0])
    net.simulate()
    d = net.extract_features_from_spikes()
    sm_ref, _ = O.lif_run(res, rasters[0])
    np.testing.assert_array_equal(net.spike_matrix, sm_ref)
    ref = O.spike_features(sm_ref, res.out_idx, res.burst_isi_max)
    for kname in O.FEATURE_KEYS:
        np.testing.assert_array_equal(np.isnan(d[kname]), np.isnan(ref[kname]), err_msg=kname)
        np.testing.assert_allclose(np.nan_to_num(d[kname]), np.nan_to_num(ref[kname]), rtol=1e-6,
                                   atol=0, err_msg=kname)
    assert net.num_neurons == 256


def test_full_size_properties(torch_cuda, oracle_c):
    """BASELINE configs[1] shape (F=128, N=1000, N_out=400, B=256): size-independent properties
    plus a spot check of a few clips against the oracle."""
    import torch
    from lsm_speech_classifier_amd import snn, synth
    B, c, T = 256, 128, 400
    rasters = synth.bernoulli_raster(B, c, T, 0.2, seed=1234)
    rasters[7] = 0                                       # a silent clip
    rasters[100] = rasters[3]                            # duplicates must give identical rows
    res = _reservoir(1000, 200, 400, c, rasters)
    net = snn.SNN(None, reservoir=res)
    dev = torch.from_numpy(rasters).cuda()
    feats, _, _ = net.run_batch(dev, waves_per_clip=0)
    f = feats.cpu().numpy()
    counts = f[:, :400]
    assert counts.max() <= -(-T // (res.refractory_period + 1))          # <= ceil(T/(R+1))
    assert np.all(counts == np.round(counts)) and counts.min() >= 0
    assert not f[7].any()                                                # zero input => zero spikes
    np.testing.assert_array_equal(f[100], f[3])
    perm = np.random.RandomState(0).permutation(B)                       # batch order independence
    f2, _, _ = net.run_batch(dev[torch.from_numpy(perm).cuda()], waves_per_clip=0)
    np.testing.assert_array_equal(f2.cpu().numpy(), f[perm])
    kernels = _kernels(net)
    assert "ring" in kernels
    for kernel in kernels:                                               # kernels and layouts agree
        net.set_kernel(kernel)
        # (ring rows at N=1000: quads on 2 or 4 waves, pair blocks -- 8 blocks of 128 -- on 4 or 8)
        for wpc in ((2, 4) if kernel in ("ring-quads", "ring-contiguous") else (4,) if kernel == "ring" else (1, 4, 16)):
            fw, _, _ = net.run_batch(dev, waves_per_clip=wpc)
            np.testing.assert_array_equal(fw.cpu().numpy(), f)
    net.set_kernel("auto")
    ref = oracle_c.lif_run_batch(res, rasters[:6], n_threads=6)
    np.testing.assert_array_equal(f[:6], ref)


@pytest.mark.parametrize("n,k,n_out,c,clips", [(4000, 800, 1600, 128, 2), (8000, 1600, 3200, 256, 1)])
def test_large_reservoirs_match_oracle(torch_cuda, oracle_c, n, k, n_out, c, clips):
    """BASELINE.json configs[3]/[4] shapes (W no longer fits L2): parity spot check on a few clips,
    default layout and one explicit layout."""
    from lsm_speech_classifier_amd import snn, synth
    rasters = synth.bernoulli_raster(clips, c, 400, 0.25, seed=n)
    res = _reservoir(n, k, n_out, c, rasters)
    net = snn.SNN(None, reservoir=res)
    keys = ["spike_counts", "spike_variances", "mean_spike_times", "mean_isi", "isi_variances"]
    ref = oracle_c.lif_run_batch(res, rasters, keys, n_threads=clips)
    assert ref[:, :n_out].sum() > 0
    assert "ring" in _kernels(net)
    for kernel in ("dense", "ring", "ring-contiguous", "sparse"):   # 64 / 262 MB dense tables, ring rows, CSC scatter
        net.set_kernel(kernel)
        for wpc in (0, 8):
            feats, _, _ = net.run_batch(rasters, keys, waves_per_clip=wpc)
            np.testing.assert_array_equal(feats.cpu().numpy(), ref, err_msg=f"{kernel} wpc {wpc}")


def test_oracle_on_a_reservoir_it_did_not_build(torch_cuda, oracle_c):
    """every reservoir the oracle ever saw came from the product's own build_reservoir.  Here the
    wiring is drawn by the test itself -- a random directed graph (no ring, asymmetric, negative and positive
    weights, a few neurons without inputs), random input map with channels feeding one neuron twice-removed,
    random output subset, heterogeneous leak -- and handed to BOTH sides as plain arrays."""
    from lsm_speech_classifier_amd import reservoir as R, snn, synth
    rs = np.random.RandomState(2024)
    n, c, t, n_out, fan = 333, 19, 160, 77, 5
    dens = rs.rand(n, n) < 0.06
    np.fill_diagonal(dens, False)
    dens[:, [5, 100]] = False                                  # two neurons without outgoing synapses
    dens[[7, 200], :] = False                                  # two without incoming ones
    post, pre = np.nonzero(dens)                               # CSR by postsynaptic row, presynaptic ascending
    w = (rs.randn(len(post)) * 0.08 + 0.05).astype(np.float32)
    csr_ptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(post, minlength=n), out=csr_ptr[1:])
    order = np.lexsort((post, pre))
    csc_ptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(pre, minlength=n), out=csc_ptr[1:])
    in_tgt = np.stack([np.sort(rs.choice(n, fan, replace=False)) for _ in range(c)]).astype(np.int32)
    flat_c, flat_i = np.repeat(np.arange(c, dtype=np.int32), fan), in_tgt.reshape(-1)
    o2 = np.lexsort((flat_c, flat_i))
    in_ptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(flat_i, minlength=n), out=in_ptr[1:])
    res = R.Reservoir(
        num_neurons=n, n_channels=c, seed=-1, theta=np.float32(1.0), refractory_period=3, w_in=np.float32(0.3),
        burst_isi_max=4, csr_ptr=csr_ptr, csr_pre=pre.astype(np.int32), csr_w=w,
        csc_ptr=csc_ptr, csc_post=post[order].astype(np.int32), csc_w=np.ascontiguousarray(w[order]),
        leak=np.clip(rs.normal(0.05, 0.03, n), 0, 1).astype(np.float32), in_fanout=fan, in_tgt=in_tgt,
        in_ptr=in_ptr, in_chan=flat_c[o2].astype(np.int32), out_idx=np.sort(rs.choice(n, n_out, replace=False)).astype(np.int32))
    net = snn.SNN(None, reservoir=res)
    rasters = synth.bernoulli_raster(4, c, t, 0.25, seed=5)
    assert "ring" not in _kernels(net)                         # nothing ring-like about this graph
    total = 0
    for kernel in ("dense", "sparse"):
        net.set_kernel(kernel)
        for wpc in (0, 1, 2, 4):
            total += _check_against_oracle(net, rasters, oracle_c, wpc)
    assert total > 2000                                        # it does spike (negative weights included)
    from oracle import ref_numpy as O                          # and the two oracle restatements agree on it too
    sm_np, vt_np = O.lif_run(res, rasters[0], want_trace=True)
    _, sm_c, vt_c = oracle_c.lif_run(res, rasters[0], want_trace=True)
    np.testing.assert_array_equal(sm_np, sm_c)
    np.testing.assert_array_equal(vt_np, vt_c)


@pytest.mark.parametrize("divisor,channels", [(None, 128), (5.0, 128), (None, 200), (None, 40)])
def test_ring_rows_count_inputs_from_masks_or_entries(torch_cuda, oracle_c, divisor, channels):
    """Round 4: with one leak coefficient for every neuron (the reference's default, extract_lsm_features.py:174 passes
    leak_variance_divisor=None) and at most 128 channels the ring kernel keeps per-neuron channel masks in registers and
    counts the input drive in the update (input modes 12 / 13); heterogeneous leaks need those registers and more than 128
    channels need more mask words: both keep the packed input-map entries (modes 10 / 11).  Same results either way."""
    from lsm_speech_classifier_amd import snn, synth
    n, k, t = 4096, 300, 90
    rasters = synth.bernoulli_raster(2, channels, t, 0.25, seed=channels)
    kw = {} if divisor is None else {"leak_variance_divisor": divisor}
    res = _reservoir(n, k, n // 3, channels, rasters, **kw)
    net = snn.SNN(None, reservoir=res)
    net.set_kernel("ring")
    mode = net.plan(2, t, 8)["input_mode"]
    # (round 5: 14 / 15 = the masks in the pair-block kernel, csrc/lif_pair.h, which also holds a leak coefficient per neuron)
    assert mode in ((12, 13, 14, 15) if channels <= 128 else (10, 11)), mode
    net.set_kernel("ring-quads")
    assert net.plan(2, t, 8)["input_mode"] in ((12, 13) if divisor is None and channels <= 128 else (10, 11))
    assert _check_against_oracle(net, rasters, oracle_c, 8) > 1
    net.set_kernel("ring")
    ran = 0
    for wpc in (0, 8, 16):
        ran += _check_against_oracle(net, rasters, oracle_c, wpc) > 1
    assert ran >= 2


@pytest.mark.parametrize("hub_channels", [33, 32, 1])
def test_input_maps_with_and_without_a_mask_colouring(torch_cuda, oracle_c, hub_channels):
    """Round 4: the dense kernel counts a neuron's active inputs with ONE popcount when the library finds bit positions
    for the channels such that the channels feeding one neuron differ mod 32 (INMODE 3, csrc/reservoir.hip
    colour_input_channels).  A neuron fed by 33 channels admits no such assignment: the library must fall back to
    the four-popcount form (INMODE 2) -- same results either way; 32 channels onto one neuron is the tightest map
    that can be coloured; 1 is an ordinary map.  Reference semantics: SPEC.md 3 (input term w_in * count)."""
    import copy
    from lsm_speech_classifier_amd import snn, synth
    n, k, c, t, fan = 1000, 60, 96, 150, 5
    rasters = synth.bernoulli_raster(3, c, t, 0.3, seed=hub_channels)
    res = copy.copy(_reservoir(n, k, 300, c, rasters))
    rs = np.random.RandomState(hub_channels)
    in_tgt = np.empty((c, fan), dtype=np.int32)
    for ch in range(c):
        others = rs.choice(np.arange(1, n), fan - 1, replace=False)
        first = 0 if ch < hub_channels else int(rs.randint(1, n))     # the first `hub_channels` channels all feed neuron 0
        while first in others:
            first = int(rs.randint(1, n))
        in_tgt[ch] = np.sort(np.append(others, first))
    flat_c, flat_i = np.repeat(np.arange(c, dtype=np.int32), fan), in_tgt.reshape(-1)
    o2 = np.lexsort((flat_c, flat_i))
    in_ptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(flat_i, minlength=n), out=in_ptr[1:])
    res.in_fanout, res.in_tgt, res.in_ptr, res.in_chan = fan, in_tgt, in_ptr, flat_c[o2].astype(np.int32)
    assert np.bincount(flat_i, minlength=n)[0] >= hub_channels
    net = snn.SNN(None, reservoir=res)
    net.set_kernel("dense")
    # one popcount (mode 3) when a colouring exists, four (mode 2) when 33 channels meet in one neuron
    assert net.plan(3, t, 4)["input_mode"] == (2 if hub_channels > 32 else 3)
    total = 0
    for kernel in ("dense", "sparse"):
        net.set_kernel(kernel)
        for wpc in (0, 4, 8):
            total += _check_against_oracle(net, rasters, oracle_c, wpc)
    assert total > 500
    # the pair-block ring kernel holds the same masks: coloured positions (input mode 15) or, for the 33-channel hub, natural
    # ones and four popcounts (14)
    net.set_kernel("ring-pairs")
    for wpc in (4, 8):
        assert net.plan(3, t, wpc)["input_mode"] == (14 if hub_channels > 32 else 15)
        assert _check_against_oracle(net, rasters, oracle_c, wpc) > 1


@pytest.mark.parametrize("n,k,c,t", [(1024, 120, 40, 120), (2048, 300, 96, 100), (4096, 300, 200, 80),
                                     (8192, 400, 256, 60), (6144, 500, 64, 60)])
def test_ring_rows_at_quad_multiples(torch_cuda, oracle_c, n, k, c, t):
    """Sizes whose quad count is a multiple of the wave counts (strided quad ownership exists, except at 6144 =
    24 quads) and N = 8192, the largest reservoir: no padding neurons at all, windows that wrap exactly at a quad
    boundary, input maps beyond 128 channels.  Both ring ownerships, every layout the reservoir offers."""
    from lsm_speech_classifier_amd import snn, synth
    rasters = synth.bernoulli_raster(2, c, t, 0.25, seed=n + k)
    res = _reservoir(n, k, n // 3, c, rasters)
    net = snn.SNN(None, reservoir=res)
    assert "ring" in _kernels(net)
    ran = 0
    for kernel in [k for k in _kernels(net) if k != "sparse"]:
        net.set_kernel(kernel)
        for wpc in (0, 2, 4, 8, 16):
            if kernel == "dense" and wpc not in (0, 16):
                continue
            ran += _check_against_oracle(net, rasters, oracle_c, wpc) > 1
    assert ran >= 4
