"""Stage 3 of the pipeline: LSM features (File 2) -> linear readout, accuracy and report.

Drop-in for the reference script of the same name (/root/reference/train_classifier.py): by default
the readout is scikit-learn's multinomial logistic regression on the host, exactly as there.
``LSM_READOUT=torch-logistic`` or ``torch-ridge`` (or the ``readout=`` argument) runs the PyTorch
readouts of ``lsm_speech_classifier_amd.readout`` on the GPU instead.
"""
import inspect
import os
from pathlib import Path

import numpy as np

CLASS_NAMES = ["yes", "no", "up", "visual", "backward", "stop", "bird", "cat", "nine", "eight",
               "zero", "follow"]
FEATURE_FILE = "lsm_features_larger.npz"


def _readout():
    from sklearn.linear_model import LogisticRegression
    kw = dict(random_state=42, max_iter=1000)
    # `multi_class` is deprecated from scikit-learn 1.5 (multinomial is what lbfgs does anyway)
    # and gone in 1.8: pass it only where it is still a silent, accepted argument.
    import sklearn
    major, minor = (int(p) for p in sklearn.__version__.split(".")[:2])
    if "multi_class" in inspect.signature(LogisticRegression).parameters and (major, minor) < (1, 5):
        kw["multi_class"] = "multinomial"
    return LogisticRegression(**kw)


class _TorchReadout:
    """Adapter giving the PyTorch readouts scikit-learn's fit/predict on NumPy arrays."""

    def __init__(self, kind: str):
        import torch
        from lsm_speech_classifier_amd import readout
        self.torch = torch
        self.dev = "cuda" if torch.cuda.is_available() else "cpu"
        self.model = readout.RidgeReadout(1.0) if kind == "torch-ridge" else readout.LogisticReadout(1.0, 1000)

    def fit(self, X, y):
        t = self.torch
        self.model.fit(t.from_numpy(np.asarray(X)).to(self.dev), t.from_numpy(np.asarray(y)).to(self.dev))
        return self

    def predict(self, X):
        return self.model.predict(self.torch.from_numpy(np.asarray(X)).to(self.dev)).cpu().numpy()


READOUTS = ("sklearn", "torch-ridge", "torch-logistic")


def report_results(y_train, y_test, y_pred, class_names=None):
    """The reference's final report (train_classifier.py:47-52): accuracy + per-class table.  Shared by the file
    route below and by extract_lsm_features.main_from_audio(readout=...), whose readout ran on the device."""
    from sklearn.metrics import accuracy_score, classification_report
    class_names = list(CLASS_NAMES if class_names is None else class_names)
    accuracy = accuracy_score(y_test, y_pred)
    present = sorted(set(np.unique(y_train)) | set(np.unique(y_test)))
    names = [class_names[i] if i < len(class_names) else f"class_{i}" for i in present]
    report = classification_report(y_test, y_pred, labels=present, target_names=names, zero_division=0)
    print("\n--- Final Results ---")
    print(f"Test Accuracy: {accuracy * 100:.2f}%\n")
    print("Classification Report:")
    print(report)
    return accuracy


def train_and_evaluate_classifier(readout=None, class_names=None):
    """The reference's function (no arguments there).  `readout`: one of READOUTS (default: LSM_READOUT, else the
    reference's scikit-learn logistic regression); `class_names`: report names when the dataset was built with
    another class list than the reference's 12 words (train_classifier.py:8-20 hard-codes them)."""
    readout = readout or os.environ.get("LSM_READOUT", "sklearn")
    if readout not in READOUTS:
        raise ValueError(f"readout must be one of {READOUTS}, got {readout!r}")
    if not Path(FEATURE_FILE).exists():
        print("Error: Dataset file not found. Please run 'extract_lsm_features.py' first.")
        return
    with np.load(FEATURE_FILE, allow_pickle=True) as data:
        X_train, y_train = data['X_train_features'], data['y_train']
        X_test, y_test = data['X_test_features'], data['y_test']
    print(f"Loaded {len(X_train)} training and {len(X_test)} test samples.")

    print("Training the Logistic Regression classifier..." if readout != "torch-ridge"
          else "Training the ridge classifier...")
    clf = _readout() if readout == "sklearn" else _TorchReadout(readout)
    clf.fit(X_train, y_train)
    print("Training complete.")

    print("Evaluating performance on the test set...")
    y_pred = clf.predict(X_test)
    return report_results(y_train, y_test, y_pred, class_names)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="Train and evaluate the linear readout on the LSM features.")
    ap.add_argument("--readout", type=str, default=None, choices=READOUTS,
                    help="default: LSM_READOUT, else scikit-learn logistic regression (the reference's readout)")
    ap.add_argument("--commands", type=str, default=None, help="Comma-separated class names for the report.")
    ap.add_argument("--commands-file", type=str, default=None, help="File with one class name per line.")
    a = ap.parse_args()
    names = None
    if a.commands_file:
        with open(a.commands_file) as fh:
            names = [ln.strip() for ln in fh if ln.strip() and not ln.lstrip().startswith("#")]
    elif a.commands:
        names = [w.strip() for w in a.commands.split(",") if w.strip()]
    train_and_evaluate_classifier(readout=a.readout, class_names=names)
