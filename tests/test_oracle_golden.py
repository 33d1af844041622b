"""The oracle against every golden vector taken from the reference's in-tree functions
(tests/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import ref_numpy as O

THR = [0.70, 0.80, 0.90, 0.95]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _encoder_cases(golden_dir):
    g = _load(golden_dir, "encoder.npz")
    names = sorted(k[:-3] for k in g.files if k.endswith("_in") and not k.startswith("gap005"))
    return g, names


def test_encoder_numpy_matches_reference(golden_dir):
    g, names = _encoder_cases(golden_dir)
    assert len(names) >= 20
    for n in names:
        out = O.encode_hysteresis(g[n + "_in"], list(g["thresholds"]), float(g["gap"]))
        assert out.dtype == np.uint8
        np.testing.assert_array_equal(out, g[n + "_out"], err_msg=n)
    out = O.encode_hysteresis(g["gap005_float64_in"], [0.5, 0.9, 0.3], 0.05)
    np.testing.assert_array_equal(out, g["gap005_float64_out"])


def test_encoder_c_matches_reference(golden_dir, oracle_c):
    g, names = _encoder_cases(golden_dir)
    for n in names:
        out = oracle_c.encode_hysteresis(g[n + "_in"], list(g["thresholds"]), float(g["gap"]))
        np.testing.assert_array_equal(out, g[n + "_out"], err_msg=n)
    out = oracle_c.encode_hysteresis(g["gap005_float64_in"], [0.5, 0.9, 0.3], 0.05)
    np.testing.assert_array_equal(out, g["gap005_float64_out"])


def test_redundancy(golden_dir):
    g = _load(golden_dir, "redundancy.npz")
    np.testing.assert_array_equal(O.pure_redundancy(g["x"], 1), g["r1"])
    np.testing.assert_array_equal(O.pure_redundancy(g["x"], 3), g["r3"])


@pytest.mark.parametrize("impl", ["numpy", "c"])
def test_postfilter_matches_reference(golden_dir, oracle_c, impl):
    """create_dataset.py:59-78.  log10 is the only operation that is not correctly rounded in
    every libm, so the gammatone cases allow 1e-13 absolute on values in [0, 1]; the mel cases
    (no log inside the tested span) must be bit-exact."""
    g = _load(golden_dir, "postfilter.npz")
    for n in ("gt_a", "gt_b", "gt_c", "gt_flat"):
        spec = g[n + "_in"]
        if impl == "numpy":
            out = O.normalise_resize(O.gammatone_db(spec))
        else:
            out = oracle_c.normalise_resize(oracle_c.gammatone_db(spec))
        ref = g[n + "_out"]
        assert out.shape == ref.shape == (spec.shape[0], 100)
        np.testing.assert_allclose(out, ref, rtol=0, atol=1e-13, err_msg=n)
        enc_ref = O.encode_hysteresis(ref.astype(np.float64), THR, 0.1)
        enc_out = O.encode_hysteresis(out.astype(np.float64), THR, 0.1)
        np.testing.assert_array_equal(enc_out, enc_ref, err_msg=n)
    for n in ("mel_a", "mel_b"):
        db = g[n + "_in"]
        out = O.normalise_resize(db) if impl == "numpy" else oracle_c.normalise_resize(db)
        assert out.dtype == np.float32
        np.testing.assert_array_equal(out, g[n + "_out"], err_msg=n)


def test_zoom_restatement_matches_scipy():
    from scipy.ndimage import zoom
    rs = np.random.RandomState(5)
    for n_in in (98, 101, 99, 63, 150, 200):
        for dt in (np.float64, np.float32):
            x = rs.rand(7, n_in).astype(dt)
            z = zoom(x, (1, 100 / n_in), order=1)
            assert z.shape == (7, 100)
            np.testing.assert_array_equal(O.zoom_linear(x, 100), z)


def test_w_critico(golden_dir):
    g = _load(golden_dir, "w_critico.npz")
    for n in ("dense", "sparse", "many", "kzero", "empty"):
        k, theta, ref = g[n + "_params"]
        w = O.w_critico(int(k), float(theta), int(ref), g[n + "_in"])
        assert w == float(g[n + "_out"]), n


def test_constants(golden_dir):
    g = _load(golden_dir, "constants.npz")
    assert list(g["feature_set_all"]) == O.FEATURE_KEYS
    assert list(g["reservoir"]) == [1000, 400, 0.01, 2, 2.0, 0.1, 200]
    assert list(g["frontend"]) == [16000, 1.0, 100, 0.1, 1000, 1]
