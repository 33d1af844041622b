"""Generate tests/golden/split.npz: the index permutation of the reference's train/test split.

    python tests/make_golden_split.py

/root/reference/extract_lsm_features.py:160-162 splits with
``train_test_split(X, y, test_size=0.2, random_state=42, stratify=y)``; the first <= 500 TRAINING clips then set
w_critico (:40), i.e. the reservoir's weights.  Which clips those are depends on scikit-learn's stratified
shuffle, so the permutation is pinned here for the class layouts the pipeline uses: a change
of scikit-learn that reorders it would silently change every feature.  Needs scikit-learn only (1.7.2 in the
build container); the fixture holds index arrays, nothing else.
"""
import os

import numpy as np
from sklearn.model_selection import train_test_split

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "split.npz")
# name: (classes, clips per class) -- labels in dataset order (class-major, create_dataset.py:121-162)
LAYOUTS = {
    "ref12x1000": (12, 1000),      # the reference's corpus: 12 words, MAX_SAMPLES_PER_CLASS = 1000
    "cfg1_4x200": (4, 200),        # BASELINE configs[0]
    "cfg4_35x100": (35, 100),      # BASELINE configs[3] shape, capped
    "tests_3x16": (3, 16),         # tests/test_gpu_hotpath.py
    "tests_35x5": (35, 5),         # tests/test_gpu_sharded.py (cfg4-shaped)
}


def split(n_classes, per_class):
    y = np.repeat(np.arange(n_classes, dtype=np.int32), per_class)
    idx_train, idx_test, y_train, y_test = train_test_split(np.arange(len(y)), y, test_size=0.2, random_state=42,
                                                            stratify=y)
    return idx_train.astype(np.int32), idx_test.astype(np.int32)


if __name__ == "__main__":
    import sklearn
    out = {"sklearn_version": np.array(sklearn.__version__)}
    for name, (c, p) in LAYOUTS.items():
        tr, te = split(c, p)
        out[name + "_train"], out[name + "_test"] = tr, te
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: (v.shape if v.ndim else str(v)) for k, v in out.items()})
