"""Bit-packed rasters on the GPU: pack/unpack kernels against the oracle's shift-based restatement,
packed File 1 written by create_dataset, and identical features from packed and unpacked input."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(4, 16, 400), (3, 5, 403), (1, 1, 1), (2, 3, 8), (5, 7, 9), (0, 4, 400)])
def test_pack_unpack_match_oracle(shape):
    import torch
    from lsm_speech_classifier_amd import frontend
    from oracle import ref_numpy as O
    rng = np.random.default_rng(sum(shape))
    x = (rng.random(shape) < 0.25).astype(np.uint8)
    if x.size:
        x.flat[0] = 200                               # any non-zero byte is a spike
    d = torch.from_numpy(x).cuda()
    p = frontend.pack_raster(d)
    assert p.shape == shape[:-1] + ((shape[-1] + 7) // 8,)
    assert np.array_equal(p.cpu().numpy(), O.pack_bits(x))
    u = frontend.unpack_raster(p, shape[-1])
    assert np.array_equal(u.cpu().numpy(), (x != 0).astype(np.uint8))


def test_large_round_trip_and_density():
    """Full-size batch: unpack(pack(x)) == (x != 0) and the set-bit count equals the spike count."""
    import torch
    from lsm_speech_classifier_amd import frontend, synth
    x = torch.from_numpy(synth.bernoulli_raster(2048, 128, 400, 0.2, seed=5)).cuda()
    p = frontend.pack_raster(x)
    assert p.shape == (2048, 128, 50)
    assert torch.equal(frontend.unpack_raster(p, 400), (x != 0).to(torch.uint8))
    bits = sum(int(((p >> k) & 1).sum(dtype=torch.int64)) for k in range(8))
    assert bits == int(x.count_nonzero())


def test_wrong_arguments_fail_loudly():
    import torch
    from lsm_speech_classifier_amd import frontend
    with pytest.raises(ValueError):
        frontend.pack_raster(torch.zeros((2, 8), dtype=torch.float32, device="cuda"))
    with pytest.raises(ValueError):
        frontend.unpack_raster(torch.zeros((2, 5), dtype=torch.uint8, device="cuda"), 400)
    with pytest.raises(ValueError):
        frontend.pack_raster(torch.zeros((2, 8), dtype=torch.uint8))


def test_packed_dataset_gives_identical_features(tmp_path, monkeypatch):
    import torch
    import create_dataset as cd
    import extract_lsm_features as ex
    from lsm_speech_classifier_amd import reservoir as R, snn, spikefile
    monkeypatch.chdir(tmp_path)
    words = ["yes", "no", "up"]
    cd.create_dataset(32, "gammatone", commands=words, synthetic_per_class=6, output_file="dense.npz")
    cd.create_dataset(32, "gammatone", commands=words, synthetic_per_class=6, output_file="packed.npz",
                      packed=True)
    Xd, yd = ex.load_spike_dataset("dense.npz")
    Xp, yp = ex.load_spike_dataset("packed.npz")
    assert Xd.shape == (18, 32, 400) and np.array_equal(Xd, Xp) and np.array_equal(yd, yp)
    P, T, _ = spikefile.load_packed("packed.npz")
    assert T == 400 and P.shape == (18, 32, 50)

    params = R.SimulationParams(num_neurons=256, small_world_graph_k=32, num_output_neurons=64,
                                mean_weight=0.02)
    net = snn.SNN(params, n_channels=32)
    keys = ["spike_counts", "mean_isi", "burst_counts"]
    f_dense, _, _ = net.run_batch(Xd, keys)
    f_packed, _, _ = net.run_batch(P, keys, packed_time_steps=T)
    assert torch.equal(f_dense, f_packed)
    assert float(f_dense.abs().sum()) > 0
