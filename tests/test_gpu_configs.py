"""BASELINE.json configs[3] and configs[4] at their FULL sizes on the MI355X (cfg5 was never run
under -m gpu): the 256-filter gammatone front end against the oracle, and the N=4000 / N=8000 reservoirs at
batch 1024 / 4096 through size-independent properties plus >= 8 clips against the C oracle, on every kernel.
Reference call sites: /root/reference/create_dataset.py:49-60, extract_lsm_features.py:76-89."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THR = [0.70, 0.80, 0.90, 0.95]
GAP = 0.1
KEYS = ['spike_counts', 'spike_variances', 'mean_spike_times', 'mean_isi', 'isi_variances']    # 'original'


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from lsm_speech_classifier_amd import _lib
    _lib.require_gpu()
    return torch


def test_cfg5_frontend_256_filters_matches_oracle(torch_cuda, oracle_c):
    """256 gammatone filters (4 channel groups per clip), white-noise and speech-like clips: spectrogram
    bit-exact, dB at 1e-12, normalised spectrogram at 1e-13, raster identical."""
    from lsm_speech_classifier_amd import frontend, synth
    from oracle import ref_numpy as O
    audio = np.concatenate([synth.white_noise(3, seed=1234), synth.class_chirps([0, 5, 11], seed=9)])
    fe = frontend.SpikeFrontEnd(256, "gammatone")
    assert (fe.nwin, fe.hop, fe.ncols, fe.n_channels, fe.n_steps) == (400, 160, 98, 256, 400)
    coefs = O.gammatone_coefs(16000, 256, 50)
    db, spec = fe.spectrogram_db(audio, want_spec=True)
    raster, norm = fe.spikes_from_db(db, want_norm=True)
    spec, db, norm, raster = (t.cpu().numpy() for t in (spec, db, norm, raster))
    assert raster.shape == (6, 256, 400) and raster.dtype == np.uint8
    for b in range(len(audio)):
        s_ref = oracle_c.gammatone_spec(audio[b], coefs, 400, 160, 98)
        np.testing.assert_array_equal(spec[b], s_ref)
        np.testing.assert_allclose(db[b], 20 * np.log10(s_ref + 1e-9), rtol=0, atol=1e-12)
        n_ref = oracle_c.normalise_resize(oracle_c.gammatone_db(s_ref))
        np.testing.assert_allclose(norm[b], n_ref, rtol=0, atol=1e-13)
        np.testing.assert_array_equal(raster[b], oracle_c.encode_hysteresis(n_ref, THR, GAP))
    # the batched C front end of the oracle (bench.py's all-cores baseline) is the same function
    np.testing.assert_array_equal(
        oracle_c.gammatone_frontend_batch(audio, coefs, 400, 160, 98, THR, GAP, n_threads=6), raster)


def _full_config(torch, oracle_c, n_filters, audio_kind, N, k, n_out, B, n_oracle):
    from lsm_speech_classifier_amd import frontend, reservoir as R, snn, synth
    from oracle import ref_numpy as O
    fe = frontend.SpikeFrontEnd(n_filters, "gammatone")
    audio = (synth.white_noise(B, seed=1234) if audio_kind == "white_noise"
             else synth.class_chirps(np.arange(B) % 12, seed=1234))
    audio[7] = 0.0                                           # a silent clip -> no input spikes
    audio[100] = audio[3]                                    # duplicates must give identical rows
    rasters = fe.encode(torch.from_numpy(audio).cuda())
    del audio
    assert rasters.shape == (B, n_filters, 400) and int(rasters[7].sum()) == 0
    head = rasters[:500].cpu().numpy()
    wc = O.w_critico(k, 2.0, 2, head)
    p = R.SimulationParams(num_neurons=N, num_output_neurons=n_out, small_world_graph_k=k, mean_weight=wc * 0.6)
    res = R.build_reservoir(p, n_filters)
    net = snn.SNN(None, reservoir=res)
    assert net.kernel_in_use() == "ring"                     # what `auto` picks at these sizes
    stats = torch.empty((B, 2), dtype=torch.int32, device="cuda")
    feats, _, _ = net.run_batch(rasters, KEYS, stats_out=stats)
    f = feats.cpu().numpy()
    counts = f[:, :n_out]
    assert counts.max() <= -(-400 // (res.refractory_period + 1)) and counts.min() >= 0
    assert np.all(counts == np.round(counts)) and counts.sum() > 0
    assert not f[7].any() and stats[7].tolist() == [0, 0]    # zero input => zero spikes
    np.testing.assert_array_equal(f[100], f[3])
    st = stats.cpu().numpy()
    assert np.all(st[:, 0] <= N) and np.all(st[:, 1] >= counts.sum(axis=1))   # all neurons >= output neurons
    perm = torch.from_numpy(np.random.RandomState(0).permutation(B)).cuda()    # batch order independence
    f2, _, _ = net.run_batch(rasters[perm], KEYS)
    assert torch.equal(f2, feats[perm])
    del f2
    others = ["dense", "sparse"]
    if N == 4000:                                            # cfg4 runs on pair blocks (csrc/lif_pair.h): the quad form
        others = ["ring-quads"] + others                     # (csrc/lif_ring.h) must agree on the whole batch too
    for kernel in others:                                    # the other kernels agree on the whole batch
        net.set_kernel(kernel)
        st2 = torch.empty_like(stats)
        fk, _, _ = net.run_batch(rasters, KEYS, stats_out=st2)
        assert torch.equal(fk, feats), kernel
        assert torch.equal(st2, stats), kernel
        del fk
    net.set_kernel("auto")
    # >= 8 clips (the silent one, the duplicate pair and the busiest clip among them) against the C oracle
    busiest = int(st[:, 1].argmax())
    idx = sorted(set([0, 1, 2, 3, 7, 100, B - 1, busiest] + list(range(4, 4 + max(0, n_oracle - 8)))))[:max(8, n_oracle)]
    sub = rasters[torch.tensor(idx).cuda()].cpu().numpy()
    ref = oracle_c.lif_run_batch(res, sub, KEYS, n_threads=min(len(idx), os.cpu_count() or 1))
    np.testing.assert_array_equal(f[idx], ref)
    # ... and their in-kernel statistics against the oracle's spike matrices
    for j in (busiest, 0):
        _, sm, _ = oracle_c.lif_run(res, rasters[j].cpu().numpy(), KEYS)
        per = sm.sum(axis=0)
        assert st[j].tolist() == [int(np.count_nonzero(per)), int(per.sum())]


def test_cfg4_full_batch(torch_cuda, oracle_c):
    """configs[3]: 128 filters, N=4000, k=800, batch 1024 (speech-like clips)."""
    _full_config(torch_cuda, oracle_c, 128, "speech_like", 4000, 800, 1600, 1024, 8)


def test_cfg5_full_batch(torch_cuda, oracle_c):
    """configs[4]: white noise, 256 filters, N=8000, k=1600, batch 4096 (the HBM-roofline stress config)."""
    _full_config(torch_cuda, oracle_c, 256, "white_noise", 8000, 1600, 3200, 4096, 8)
