"""Seeded random configurations of the reservoir path on the GPU against the C oracle: odd sizes (N not a
multiple of 64, C not a multiple of 32, T not a multiple of 4), every feature set, heterogeneous leak,
refractory 0-5, quiet and saturated drive, both kernels, the chosen layout and a forced one.  Everything
(spike matrix, float32 membrane trace, features) must be bit-identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from lsm_speech_classifier_amd import _lib
    _lib.require_gpu()
    return torch

ALL_KEYS = ['spike_counts', 'spike_variances', 'mean_spike_times', 'first_spike_times',
            'last_spike_times', 'mean_isi', 'isi_variances', 'burst_counts']


def _cases(n_cases, seed, n_lo=40, n_hi=700, k_hi=60):
    rng = np.random.RandomState(seed)
    for i in range(n_cases):
        n = int(rng.randint(n_lo, n_hi))
        k = int(2 * rng.randint(2, max(3, min(n // 4, k_hi))))
        c = int(rng.randint(1, 200))
        t = int(rng.choice([1, 7, 37, 100, 255, 400]))
        yield dict(
            n=n, k=k, c=c, t=t, n_out=int(rng.randint(1, n + 1)),
            density=float(rng.choice([0.0, 0.02, 0.2, 0.6, 1.0])),
            mult=float(rng.choice([0.3, 0.6, 1.5, 4.0])),
            refr=int(rng.randint(0, 6)),
            div=(None if rng.rand() < 0.5 else float(rng.choice([2.0, 5.0, 20.0]))),
            keys=[ALL_KEYS[j] for j in sorted(rng.choice(8, size=rng.randint(1, 9), replace=False))],
            wpc=int(rng.choice([1, 2, 4, 8, 16])), seed=1000 * seed + i)


@pytest.mark.parametrize("seed,n_cases,n_lo,n_hi,k_hi", [(1, 16, 40, 700, 60), (2, 16, 40, 700, 60),
                                                       (3, 16, 40, 700, 60),
                                                       # sizes at which the ring-row kernel exists (>= 3 quads)
                                                       (4, 8, 700, 2700, 160), (5, 8, 700, 2700, 160)])
def test_random_configurations_match_the_oracle(torch_cuda, oracle_c, seed, n_cases, n_lo, n_hi, k_hi):
    from lsm_speech_classifier_amd import _lib, reservoir as R, snn, synth
    from oracle import ref_numpy as O
    spikes = 0
    ring_runs = 0
    for case in _cases(n_cases, seed, n_lo, n_hi, k_hi):
        rasters = synth.bernoulli_raster(3, case["c"], case["t"], case["density"], seed=case["seed"])
        rasters[1] = (rasters[1] * 201).astype(np.uint8)                # any non-zero byte is a spike
        wc = O.w_critico(case["k"], 2.0, case["refr"], rasters)
        p = R.SimulationParams(num_neurons=case["n"], num_output_neurons=case["n_out"],
                               small_world_graph_k=case["k"], mean_weight=wc * case["mult"],
                               refractory_period=case["refr"], leak_variance_divisor=case["div"])
        res = R.build_reservoir(p, case["c"])
        net = snn.SNN(None, reservoir=res)
        kernels = ["dense", "sparse"]
        try:
            net.set_kernel("ring")
            kernels += ["ring", "ring-quads", "ring-contiguous"]
        except _lib.LsmHipError:
            pass                                       # not ring-like enough, or too small, for ring rows
        for kernel in kernels:
            net.set_kernel(kernel)
            for wpc in (0, case["wpc"]):
                try:
                    f, sm, vt = net.run_batch(rasters, case["keys"], want_spike_matrix=True,
                                              want_v_trace=True, waves_per_clip=wpc)
                except _lib.LsmHipError as e:          # a forced layout this reservoir does not have
                    assert "layout" in str(e) and wpc != 0, (case, str(e))
                    continue
                ring_runs += kernel == "ring-contiguous"
                f, sm, vt = f.cpu().numpy(), sm.cpu().numpy(), vt.cpu().numpy()
                for b in range(len(rasters)):
                    f_ref, sm_ref, vt_ref = oracle_c.lif_run(res, rasters[b], case["keys"], want_trace=True)
                    msg = f"{case} kernel {kernel} wpc {wpc} clip {b}"
                    np.testing.assert_array_equal(sm[b], sm_ref, err_msg=msg)
                    np.testing.assert_array_equal(vt[b], vt_ref, err_msg=msg)
                    np.testing.assert_array_equal(f[b], f_ref, err_msg=msg)
                    spikes += int(sm_ref.sum())
    assert spikes > (10000 if n_hi <= 700 else 3000)       # the cases do exercise spiking networks
    assert n_lo < 700 or ring_runs >= n_cases              # the large cases do run the ring-row kernel


@pytest.mark.parametrize("n_samples,nw", [(12000, 4), (16000, 3), (24000, 2), (48000, 1), (13000, 4)])
def test_front_end_other_clip_lengths_and_window_overlaps(torch_cuda, oracle_c, n_samples, nw):
    """The reference computes hop_time from the clip length (create_dataset.py:50), so other lengths change
    how many analysis windows overlap: 4, 3, 2 or 1 window kernels, hops that are not multiples of 8 (13000
    samples: hop 130), odd filter counts, several thresholds and redundancy -- all against the C oracle."""
    from lsm_speech_classifier_amd import frontend, synth
    from oracle import ref_numpy as O
    rng = np.random.RandomState(n_samples)
    F = int(rng.choice([2, 33, 64, 65, 130]))
    thr = sorted(float(x) for x in rng.choice(np.arange(0.05, 0.99, 0.05), size=rng.randint(1, 6), replace=False))
    gap = float(rng.choice([0.02, 0.05, 0.1]))
    red = int(rng.randint(1, 4))
    fe = frontend.SpikeFrontEnd(F, "gammatone", redundancy=red, thresholds=thr, gap=gap, n_samples=n_samples)
    assert (fe.nwin + fe.hop - 1) // fe.hop == nw
    t = np.arange(n_samples) / 16000.0
    audio = np.stack([
        (0.3 * np.sin(2 * np.pi * (200 + 900 * t) * t) + 0.02 * rng.standard_normal(n_samples)),
        0.1 * rng.standard_normal(n_samples),
        np.zeros(n_samples)]).astype(np.float32)
    coefs = O.gammatone_coefs(16000, F, 50)
    db, spec = fe.spectrogram_db(audio, want_spec=True)
    raster, norm = fe.spikes_from_db(db, want_norm=True)
    spec, norm, raster = spec.cpu().numpy(), norm.cpu().numpy(), raster.cpu().numpy()
    assert raster.shape == (3, F * red, 100 * len(thr))
    for b in range(3):
        s_ref = oracle_c.gammatone_spec(audio[b], coefs, fe.nwin, fe.hop, fe.ncols)
        np.testing.assert_array_equal(spec[b], s_ref)
        n_ref = oracle_c.normalise_resize(oracle_c.gammatone_db(s_ref))
        np.testing.assert_allclose(norm[b], n_ref, rtol=0, atol=1e-13)
        r_own = oracle_c.encode_hysteresis(norm[b], thr, gap)              # exact on the GPU's own input
        np.testing.assert_array_equal(raster[b], np.repeat(r_own, red, axis=0))
    assert not raster[2].any()                                             # silence stays silent


def test_reservoir_size_limit_is_enforced_at_create_time(torch_cuda, oracle_c):
    """8192 neurons is the limit (the per-clip LDS image of a larger reservoir fits no CU): 8192 works and is
    bit-exact, 8193 is refused with a message when the reservoir is created, not at run time."""
    from lsm_speech_classifier_amd import _lib, reservoir as R, snn, synth
    from oracle import ref_numpy as O
    c, t, k = 20, 40, 24
    rasters = synth.bernoulli_raster(2, c, t, 0.3, seed=4)
    wc = O.w_critico(k, 2.0, 2, rasters)
    big = R.build_reservoir(R.SimulationParams(num_neurons=8193, num_output_neurons=100, small_world_graph_k=k,
                                               mean_weight=wc * 2.0), c)
    with pytest.raises(_lib.LsmHipError, match="8192"):
        snn.SNN(None, reservoir=big)
    res = R.build_reservoir(R.SimulationParams(num_neurons=8192, num_output_neurons=300, small_world_graph_k=k,
                                               mean_weight=wc * 2.0), c)
    net = snn.SNN(None, reservoir=res)
    keys = ["spike_counts", "mean_isi", "burst_counts"]
    for kernel in ("dense", "sparse"):
        net.set_kernel(kernel)
        f, sm, vt = net.run_batch(rasters, keys, want_spike_matrix=True, want_v_trace=True)
        for b in range(2):
            f_ref, sm_ref, vt_ref = oracle_c.lif_run(res, rasters[b], keys, want_trace=True)
            np.testing.assert_array_equal(sm[b].cpu().numpy(), sm_ref)
            np.testing.assert_array_equal(vt[b].cpu().numpy(), vt_ref)
            np.testing.assert_array_equal(f[b].cpu().numpy(), f_ref)
            assert sm_ref.any()
