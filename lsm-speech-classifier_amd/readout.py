"""Linear readouts on the device: StandardScaler, multinomial logistic regression
and a ridge classifier in PyTorch (ROCm on the GPU box, CPU tensors in the tests).

They mirror what the reference does on the host with scikit-learn
(/root/reference/extract_lsm_features.py:199-201 StandardScaler; /root/reference/train_classifier.py:36-45
LogisticRegression(multinomial, max_iter=1000)), plus the ridge readout that BASELINE.json configs[3]
names and the reference lacks.  BASELINE.json's north star keeps the readout in PyTorch on purpose: this
is plumbing around the hot path, not a hand-written kernel.  Parity target = scikit-learn on the same
arrays (tests/test_readout.py): identical predictions for ridge, >= 99 % agreement for logistic.
"""
from __future__ import annotations

import torch


class StandardScaler:
    """sklearn.preprocessing.StandardScaler semantics: population variance, zero-variance columns
    are left unscaled, statistics in float64."""

    def fit(self, X: torch.Tensor):
        X64 = X.to(torch.float64)
        self.mean_ = X64.mean(dim=0)
        var = ((X64 - self.mean_) ** 2).mean(dim=0)
        self.scale_ = torch.where(var > 0, var.sqrt(), torch.ones_like(var))
        # sklearn treats variances at rounding-noise level as zero as well
        eps = torch.finfo(torch.float64).eps
        noise = var <= (X64.shape[0] * eps * var.new_tensor(1.0) * self.mean_.abs() ** 2 * 10)
        self.scale_ = torch.where(noise, torch.ones_like(var), self.scale_)
        return self

    def transform(self, X: torch.Tensor) -> torch.Tensor:
        return ((X.to(torch.float64) - self.mean_) / self.scale_).to(X.dtype if X.dtype.is_floating_point
                                                                     else torch.float64)

    def fit_transform(self, X: torch.Tensor) -> torch.Tensor:
        return self.fit(X).transform(X)


class RidgeReadout:
    """One-vs-rest ridge regression on {-1, +1} targets with an unpenalised intercept
    (sklearn.linear_model.RidgeClassifier): closed form in float64."""

    def __init__(self, alpha: float = 1.0):
        self.alpha = float(alpha)

    def fit(self, X: torch.Tensor, y: torch.Tensor):
        X64 = X.to(torch.float64)
        self.classes_ = torch.unique(y)
        Y = (y[:, None] == self.classes_[None, :]).to(torch.float64) * 2.0 - 1.0
        xm, ym = X64.mean(dim=0), Y.mean(dim=0)
        Xc, Yc = X64 - xm, Y - ym
        n, d = Xc.shape
        if d <= n:
            A = Xc.T @ Xc
            A.diagonal().add_(self.alpha)
            W = torch.linalg.solve(A, Xc.T @ Yc)
        else:                                            # dual form when features outnumber samples
            K = Xc @ Xc.T
            K.diagonal().add_(self.alpha)
            W = Xc.T @ torch.linalg.solve(K, Yc)
        self.coef_ = W.T.contiguous()
        self.intercept_ = ym - xm @ W
        return self

    def decision_function(self, X: torch.Tensor) -> torch.Tensor:
        return X.to(torch.float64) @ self.coef_.T + self.intercept_

    def predict(self, X: torch.Tensor) -> torch.Tensor:
        return self.classes_[self.decision_function(X).argmax(dim=1)]


class LogisticReadout:
    """Multinomial logistic regression, L2 penalty 1/(2C) ||W||^2 on the coefficients (not on the
    intercept), full-batch L-BFGS in float64: the objective scikit-learn's lbfgs solver minimises."""

    def __init__(self, C: float = 1.0, max_iter: int = 1000, tol: float = 1e-6):
        self.C, self.max_iter, self.tol = float(C), int(max_iter), float(tol)

    def fit(self, X: torch.Tensor, y: torch.Tensor):
        X64 = X.to(torch.float64)
        self.classes_ = torch.unique(y)
        k, d = len(self.classes_), X64.shape[1]
        target = (y[:, None] == self.classes_[None, :]).to(torch.float64).argmax(dim=1)
        W = torch.zeros((k, d), dtype=torch.float64, device=X.device, requires_grad=True)
        b = torch.zeros(k, dtype=torch.float64, device=X.device, requires_grad=True)
        opt = torch.optim.LBFGS([W, b], lr=1.0, max_iter=self.max_iter, tolerance_grad=self.tol,
                                tolerance_change=1e-12, history_size=10, line_search_fn="strong_wolfe")

        def closure():
            opt.zero_grad()
            logits = X64 @ W.T + b
            loss = torch.nn.functional.cross_entropy(logits, target, reduction="sum") \
                + 0.5 / self.C * (W * W).sum()
            loss.backward()
            return loss

        opt.step(closure)
        self.coef_, self.intercept_ = W.detach(), b.detach()
        return self

    def decision_function(self, X: torch.Tensor) -> torch.Tensor:
        return X.to(torch.float64) @ self.coef_.T + self.intercept_

    def predict(self, X: torch.Tensor) -> torch.Tensor:
        return self.classes_[self.decision_function(X).argmax(dim=1)]
