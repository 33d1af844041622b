"""File 1 schemas (CPU only): the reference's uint8 layout and the bit-packed variant must load to the
same arrays, and the packing convention is checked against the oracle's shift-based restatement."""
import numpy as np
import pytest

from lsm_speech_classifier_amd import spikefile
from oracle import ref_numpy as O


@pytest.mark.parametrize("T", [1, 7, 8, 9, 400, 403])
def test_pack_convention_matches_oracle(T):
    rng = np.random.default_rng(T)
    x = (rng.random((3, 5, T)) < 0.3).astype(np.uint8)
    x[0, 0, 0] = 7                                   # any non-zero byte is a spike
    p = spikefile.pack_host(x)
    assert p.dtype == np.uint8 and p.shape == (3, 5, (T + 7) // 8)
    assert np.array_equal(p, O.pack_bits(x))
    assert np.array_equal(spikefile.unpack_host(p, T), (x != 0).astype(np.uint8))
    assert np.array_equal(O.unpack_bits(p, T), (x != 0).astype(np.uint8))
    if T % 8:                                         # unused high bits of the last byte are zero
        assert int(p[..., -1].max()) < (1 << (T % 8))


def test_known_vector():
    x = np.zeros((1, 1, 12), dtype=np.uint8)
    x[0, 0, [0, 3, 8, 11]] = 1
    assert spikefile.pack_host(x).tolist() == [[[0b00001001, 0b00001001]]]


def test_both_schemas_load_to_the_reference_arrays(tmp_path):
    rng = np.random.default_rng(0)
    X = (rng.random((6, 4, 400)) < 0.2).astype(np.uint8)
    y = np.arange(6, dtype=np.int32) % 3
    dense, packed = tmp_path / "dense.npz", tmp_path / "packed.npz"
    spikefile.save(dense, X_spikes=X, y_labels=y)
    spikefile.save(packed, packed=spikefile.pack_host(X), time_steps=400, y_labels=y)
    with np.load(dense) as d:                         # the reference's reader sees its own schema
        assert set(d.files) == {"X_spikes", "y_labels"} and d["X_spikes"].dtype == np.uint8
        assert d["y_labels"].dtype == np.int32
    with np.load(packed) as d:
        assert set(d.files) == {"X_spikes_packed", "time_steps", "y_labels"}
        assert d["X_spikes_packed"].shape == (6, 4, 50)
    for f in (dense, packed):
        Xl, yl = spikefile.load(f)
        assert Xl.dtype == np.uint8 and np.array_equal(Xl, X) and np.array_equal(yl, y)
        P, T, yp = spikefile.load_packed(f)
        assert T == 400 and np.array_equal(P, O.pack_bits(X)) and np.array_equal(yp, y)
    assert packed.stat().st_size < dense.stat().st_size * 1.05   # never meaningfully larger once compressed


def test_extract_loader_accepts_both(tmp_path, capsys):
    import extract_lsm_features as ex
    X = (np.random.default_rng(1).random((4, 3, 16)) < 0.5).astype(np.uint8)
    y = np.zeros(4, dtype=np.int32)
    spikefile.save(tmp_path / "a.npz", X_spikes=X, y_labels=y)
    spikefile.save(tmp_path / "b.npz", packed=spikefile.pack_host(X), time_steps=16, y_labels=y)
    for name in ("a.npz", "b.npz"):
        Xl, yl = ex.load_spike_dataset(str(tmp_path / name))
        assert np.array_equal(Xl, X) and np.array_equal(yl, y)
    assert "Loaded 4 samples" in capsys.readouterr().out


def test_bad_shapes_are_rejected(tmp_path):
    with pytest.raises(ValueError):
        spikefile.save(tmp_path / "x.npz", packed=np.zeros((2, 3, 5), np.uint8), time_steps=400,
                       y_labels=np.zeros(2))
    with pytest.raises(ValueError):
        spikefile.save(tmp_path / "x.npz", packed=np.zeros((2, 3, 50), np.uint8), y_labels=np.zeros(2))
    with pytest.raises(ValueError):
        spikefile.unpack_host(np.zeros((2, 3, 5), np.uint8), 400)
    with pytest.raises(ValueError):
        spikefile.save(tmp_path / "x.npz", X_spikes=np.zeros((2, 3, 5), np.uint8), y_labels=np.zeros(3))
