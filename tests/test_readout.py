"""Device readouts (PyTorch) vs scikit-learn on the same arrays: every test runs on CPU tensors (here) and,
marked `gpu`, on ROCm tensors on the MI355X box (/root/reference/train_classifier.py:36-45,
extract_lsm_features.py:199-201) -- same bars on both."""
import numpy as np
import pytest
import torch

from lsm_speech_classifier_amd import readout

DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _n(t):
    return t.detach().cpu().numpy()


def _data(n=600, d=80, k=6, seed=0):
    rs = np.random.RandomState(seed)
    centers = rs.randn(k, d) * 1.2
    y = rs.randint(0, k, size=n).astype(np.int32)
    X = centers[y] + rs.randn(n, d) * 2.0
    X[:, 5] = 3.0                                         # a constant column
    X *= rs.uniform(0.1, 50.0, size=d)                    # very different scales
    return X[: n * 3 // 4], y[: n * 3 // 4], X[n * 3 // 4:], y[n * 3 // 4:]


@pytest.mark.parametrize("device", DEVICES)
def test_standard_scaler_matches_sklearn(device):
    from sklearn.preprocessing import StandardScaler
    Xtr, _, Xte, _ = _data()
    sk = StandardScaler().fit(Xtr)
    ours = readout.StandardScaler().fit(_t(Xtr, device))
    assert ours.mean_.device.type == device
    np.testing.assert_allclose(_n(ours.mean_), sk.mean_, rtol=1e-12)
    np.testing.assert_allclose(_n(ours.scale_), sk.scale_, rtol=1e-10)
    np.testing.assert_allclose(_n(ours.transform(_t(Xte, device))), sk.transform(Xte), rtol=1e-9, atol=1e-12)
    f32 = ours.transform(_t(Xte.astype(np.float32), device))
    assert f32.dtype == torch.float32


@pytest.mark.parametrize("device", DEVICES)
def test_ridge_readout_matches_sklearn(device):
    from sklearn.linear_model import RidgeClassifier
    from sklearn.preprocessing import StandardScaler
    Xtr, ytr, Xte, yte = _data()
    sc = StandardScaler().fit(Xtr)
    A, B = sc.transform(Xtr), sc.transform(Xte)
    for alpha in (1.0, 100.0):
        sk = RidgeClassifier(alpha=alpha).fit(A, ytr)
        ours = readout.RidgeReadout(alpha).fit(_t(A, device), _t(ytr, device))
        np.testing.assert_allclose(_n(ours.coef_), sk.coef_, rtol=1e-7, atol=1e-9)
        np.testing.assert_array_equal(_n(ours.predict(_t(B, device))), sk.predict(B))
    # more features than samples: dual form
    small = readout.RidgeReadout(1.0).fit(_t(A[:40], device), _t(ytr[:40], device))
    sk = RidgeClassifier(alpha=1.0).fit(A[:40], ytr[:40])
    np.testing.assert_array_equal(_n(small.predict(_t(B, device))), sk.predict(B))


@pytest.mark.parametrize("device", DEVICES)
def test_logistic_readout_matches_sklearn(device):
    from sklearn.linear_model import LogisticRegression
    from sklearn.preprocessing import StandardScaler
    Xtr, ytr, Xte, yte = _data()
    sc = StandardScaler().fit(Xtr)
    A, B = sc.transform(Xtr), sc.transform(Xte)
    sk = LogisticRegression(random_state=42, max_iter=1000).fit(A, ytr)
    ours = readout.LogisticReadout(C=1.0, max_iter=1000).fit(_t(A, device), _t(ytr, device))
    agree = (_n(ours.predict(_t(B, device))) == sk.predict(B)).mean()
    assert agree >= 0.99
    # both minimise the same objective; scikit-learn stops at its own tolerance, so compare the
    # objective values (ours must not be worse) and the coefficients loosely
    def objective(W, b):
        z = A @ W.T + b
        z = z - z.max(axis=1, keepdims=True)
        logp = z - np.log(np.exp(z).sum(axis=1, keepdims=True))
        idx = np.searchsorted(sk.classes_, ytr)
        return -logp[np.arange(len(ytr)), idx].sum() + 0.5 * (W * W).sum()
    assert objective(_n(ours.coef_), _n(ours.intercept_)) <= objective(sk.coef_, sk.intercept_) * (1 + 1e-6)
    np.testing.assert_allclose(_n(ours.coef_), sk.coef_, rtol=0, atol=0.05 * np.abs(sk.coef_).max())
    acc_ours = (_n(ours.predict(_t(B, device))) == yte).mean()
    acc_sk = (sk.predict(B) == yte).mean()
    assert abs(acc_ours - acc_sk) <= 0.01
