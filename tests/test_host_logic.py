"""CPU-only checks of the host side: C-ABI exports, parameter tables, reservoir builder, sharding
arithmetic, script surface, loud failure without a GPU."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """The built liblsm_hip.so must load here (no GPU needed) and export every function that
    include/lsm_hip.h declares -- no compute call is made.  (_lib.load() itself rebuilds a library that is absent or was
    built from other sources than the tree's.)"""
    from lsm_speech_classifier_amd import _lib, build
    header = open(os.path.join(ROOT, "include", "lsm_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:const\s+char\s*\*|int|long)\s*\*?\s*(lsm_[a-z0-9_]+)\s*\(", header, re.M))
    assert len(declared) >= 12
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in lsm_hip.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    assert lib.lsm_version() == build.version_number() >= 100
    # the library says which sources it was built from, and that is this tree
    assert lib.lsm_build_id().decode() == "LSM_BUILD_ID=" + build.source_id() and build.built_id() == build.source_id()


def test_a_library_built_from_other_sources_is_refused(tmp_path):
    """A stale binary must not pass for a build of the sources beside it: a copy of the library whose embedded build id
    is altered is refused by the loader (auto-rebuild off), with both ids in the message."""
    from lsm_speech_classifier_amd import build
    blob = open(build.lib_path(), "rb").read()
    mark = b"LSM_BUILD_ID=" + build.source_id().encode()
    assert blob.count(mark) >= 1
    bad = blob.replace(mark, b"LSM_BUILD_ID=" + b"0" * 24)
    pkg = tmp_path / "pkg"
    import shutil
    shutil.copytree(os.path.join(ROOT, "lsm-speech-classifier_amd"), pkg, ignore=shutil.ignore_patterns("build", "__pycache__"))
    (pkg / "liblsm_hip.so").write_bytes(bad)
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")         # part of what the build id hashes
    code = ("import importlib.util, sys, os\n"
            "os.environ['LSM_NO_AUTO_BUILD'] = '1'\n"
            "spec = importlib.util.spec_from_file_location('lsm_speech_classifier_amd', %r, submodule_search_locations=[%r])\n"
            "m = importlib.util.module_from_spec(spec); sys.modules['lsm_speech_classifier_amd'] = m; spec.loader.exec_module(m)\n"
            "from lsm_speech_classifier_amd import _lib\n"
            "try:\n    _lib.load()\nexcept _lib.LsmHipError as e:\n    print('REFUSED', 'other sources' in str(e), '0' * 24 in str(e))\n"
            ) % (str(pkg / "__init__.py"), str(pkg))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "REFUSED True True" in out.stdout, out.stdout + out.stderr


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from lsm_speech_classifier_amd import _lib, frontend, reservoir, snn
    with pytest.raises(_lib.LsmHipError):
        frontend.SpikeFrontEnd(8, "gammatone")
    with pytest.raises(_lib.LsmHipError):
        snn.SNN(reservoir.SimulationParams(num_neurons=64, small_world_graph_k=8), n_channels=4)
    with pytest.raises(_lib.LsmHipError):
        frontend.convert_spectrogram_to_spikes_hysteresis(np.zeros((2, 5)), [0.5], 0.1)


def test_one_gammatone_filter_is_refused_with_the_reason():
    """SPEC.md 1.1: with one channel the reference's window sums take NumPy's pairwise order, which the GPU filterbank does
    not restate: refused before anything touches the GPU (no GPU needed to see the message)."""
    from lsm_speech_classifier_amd import frontend
    with pytest.raises(ValueError, match="n_filters >= 2"):
        frontend.SpikeFrontEnd(1, "gammatone")
    with pytest.raises(ValueError, match="filterbank must be"):
        frontend.SpikeFrontEnd(8, "bark")


def test_missing_library_is_an_error_not_a_fallback():
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['LSM_HIP_LIB'] = '/nonexistent/liblsm_hip.so';"
            "from lsm_speech_classifier_amd import _lib\n"
            "try:\n    _lib.load()\nexcept _lib.LsmHipError as e:\n    print('RAISED', 'no CPU fallback' in str(e).lower() or 'not built' in str(e))\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "RAISED True" in out.stdout, out.stdout + out.stderr


def test_product_never_imports_the_oracle():
    offenders = []
    files = [os.path.join(ROOT, f) for f in ("create_dataset.py", "extract_lsm_features.py",
                                             "train_classifier.py", "main.py")]
    pkg = os.path.join(ROOT, "lsm-speech-classifier_amd")
    for d, _, names in os.walk(pkg):
        files += [os.path.join(d, n) for n in names if n.endswith((".py", ".hip", ".h"))]
    for f in files:
        txt = open(f).read()
        if re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M) or "liblsm_oracle" in txt or "oracle/" in txt:
            offenders.append(f)
    assert not offenders, offenders


def test_gammatone_table_matches_oracle_restatement():
    from lsm_speech_classifier_amd import frontend
    from oracle import ref_numpy as O
    for F in (2, 40, 128, 256):
        tab = frontend.gammatone_filter_table(16000, F, 50)
        np.testing.assert_array_equal(tab, O.gammatone_coefs(16000, F, 50))
        assert frontend.coef_flags(tab) == 7              # A2 == 0, divisible gains, one A0/B0 for every channel
        assert np.all(np.diff(tab[:, 7]) != 0)
    assert frontend.gtgram_strides(16000, 0.025, 0.01, 16000) == (400, 160, 98)
    bad = frontend.gammatone_filter_table(16000, 4, 50)
    bad[0, 5] = 1e-3
    assert frontend.coef_flags(bad) & 1 == 0
    other = frontend.gammatone_filter_table(16000, 4, 50)
    other[2, 0] *= 1.0 + 2.0 ** -40                          # one channel with its own first-section gain
    assert frontend.coef_flags(other) == 3


def test_threshold_tables_follow_the_reference_rounding():
    from lsm_speech_classifier_amd import frontend
    on, off = frontend.threshold_tables([0.70, 0.80, 0.90, 0.95], 0.1, np.float64)
    assert list(on) == [0.95, 0.9, 0.8, 0.7]
    assert off[2] == 0.8 - 0.1 and off[2] != 0.7            # Python-float subtraction, not 0.7
    on32, off32 = frontend.threshold_tables([0.70, 0.80, 0.90, 0.95], 0.1, np.float32)
    assert on32.dtype == np.float32 and off32[0] == np.float32(0.95 - 0.1)


@pytest.mark.parametrize("n,k,c", [(64, 8, 5), (200, 40, 32), (1000, 200, 128)])
def test_reservoir_builder_invariants(n, k, c):
    from lsm_speech_classifier_amd import reservoir as R
    p = R.SimulationParams(num_neurons=n, num_output_neurons=max(1, int(0.4 * n)),
                           small_world_graph_k=k, mean_weight=0.01)
    a, b = R.build_reservoir(p, c), R.build_reservoir(p, c)
    for key in ("csr_ptr", "csr_pre", "csr_w", "csc_ptr", "csc_post", "csc_w", "leak", "in_tgt", "out_idx"):
        np.testing.assert_array_equal(getattr(a, key), getattr(b, key))       # deterministic
    assert a.nnz == n * k                                   # rewiring keeps the edge count
    dense = np.zeros((n, n), dtype=np.float32)
    post = np.repeat(np.arange(n), np.diff(a.csr_ptr))
    dense[post, a.csr_pre] = a.csr_w
    assert not dense.diagonal().any()                       # no self loops
    assert np.array_equal(dense != 0, (dense != 0).T)       # every edge has both directions
    for i in range(n):
        row = a.csr_pre[a.csr_ptr[i]:a.csr_ptr[i + 1]]
        assert np.all(np.diff(row) > 0)
    pre = np.repeat(np.arange(n), np.diff(a.csc_ptr))       # CSC is the same matrix
    dense2 = np.zeros_like(dense)
    dense2[a.csc_post, pre] = a.csc_w
    np.testing.assert_array_equal(dense, dense2)
    assert np.all(np.diff(a.out_idx) > 0) and len(a.out_idx) == p.num_output_neurons
    assert a.in_tgt.shape == (c, R.input_fanout(n, c))
    assert all(len(set(r)) == len(r) for r in a.in_tgt)
    assert a.w_in == np.float32(R.W_IN_SCALE * 2.0) and a.csr_w.dtype == np.float32
    assert np.all(a.leak == np.float32(0.01))
    c2 = R.build_reservoir(p, c, seed=7)
    assert not np.array_equal(c2.csr_pre, a.csr_pre) or not np.array_equal(c2.csr_w, a.csr_w)


def test_reservoir_heterogeneous_leak_and_fanout():
    from lsm_speech_classifier_amd import reservoir as R
    p = R.SimulationParams(num_neurons=300, small_world_graph_k=20, mean_weight=0.01,
                           leak_variance_divisor=4.0)
    r = R.build_reservoir(p, 16)
    assert r.leak.std() > 0 and r.leak.min() >= 0 and r.leak.max() <= 1
    assert R.input_fanout(1000, 128) == 8 and R.input_fanout(500, 40) == 13 and R.input_fanout(8000, 256) == 31
    assert R.input_fanout(10, 100) == 1
    with pytest.raises(ValueError):
        R.build_reservoir(R.SimulationParams(num_neurons=10, small_world_graph_k=10), 4)


def test_shard_bounds_cover_all_clips_in_order():
    from lsm_speech_classifier_amd import dist
    for n in (0, 1, 7, 256, 9600, 9601):
        for world in (1, 2, 3, 8):
            b = dist.shard_bounds(n, world)
            assert len(b) == world and b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert all(hi - lo <= -(-n // world) for lo, hi in b)


def test_script_surface_matches_the_reference(golden_dir):
    sys.path.insert(0, ROOT)
    import create_dataset as cd
    import extract_lsm_features as ex
    import main as pipeline
    import train_classifier as tc
    g = np.load(os.path.join(golden_dir, "constants.npz"))
    assert list(g["feature_set_names"]) == list(ex.FEATURE_SETS)
    for name in ex.FEATURE_SETS:
        assert [str(k) for k in g[f"feature_set_{name}"]] == ex.FEATURE_SETS[name]
    assert [ex.NUM_NEURONS, ex.NUM_OUTPUT_NEURONS, ex.LEAK_COEFFICIENT, ex.REFRACTORY_PERIOD,
            ex.MEMBRANE_THRESHOLD, ex.SMALL_WORLD_P, ex.SMALL_WORLD_K] == list(g["reservoir"])
    assert [cd.SAMPLE_RATE, cd.DURATION, cd.TIME_BINS, cd.HYSTERESIS_GAP, cd.MAX_SAMPLES_PER_CLASS,
            cd.REDUNDANCY_FACTOR] == list(g["frontend"])
    assert cd.SPIKE_THRESHOLDS == [0.70, 0.80, 0.90, 0.95]
    for fn in (cd.load_audio_file, cd.audio_to_spectrogram, cd.convert_spectrogram_to_spikes_hysteresis,
               cd.create_pure_redundancy, cd.create_dataset, ex.calculate_theoretical_w_critico,
               ex.load_spike_dataset, ex.extract_all_features, ex.run_network_diagnostics, ex.main,
               tc.train_and_evaluate_classifier, pipeline.run_pipeline):
        assert callable(fn)
    import inspect
    assert list(inspect.signature(cd.create_dataset).parameters)[:2] == ["n_filters", "filterbank"]
    sig = inspect.signature(ex.main).parameters
    assert list(sig)[:3] == ["feature_set", "multiplier", "leak_variance_divisor"]
    # what the reference hard-codes is keyword-only here, and None means "the reference's constant"
    assert all(p.kind is p.KEYWORD_ONLY and p.default is None for p in list(sig.values())[3:])
    assert ex.reservoir_shape() == (1000, 400, 200) == (ex.NUM_NEURONS, ex.NUM_OUTPUT_NEURONS, ex.SMALL_WORLD_K)
    assert list(inspect.signature(ex.extract_all_features).parameters) == ["lsm", "spike_data", "feature_keys", "desc"]
    assert inspect.signature(cd.convert_spectrogram_to_spikes_hysteresis).parameters["hysteresis_gap"].default == 0.05
    np.testing.assert_array_equal(cd.create_pure_redundancy(np.eye(2, dtype=np.uint8), 2),
                                  np.repeat(np.eye(2, dtype=np.uint8), 2, axis=0))


def test_w_critico_matches_reference_golden(golden_dir, capsys):
    import types
    import extract_lsm_features as ex
    g = np.load(os.path.join(golden_dir, "w_critico.npz"))
    for n in ("dense", "sparse", "many", "kzero", "empty"):
        k, theta, ref = g[n + "_params"]
        p = types.SimpleNamespace(small_world_graph_k=int(k) if k == int(k) else k,
                                  membrane_threshold=float(theta), refractory_period=int(ref))
        assert ex.calculate_theoretical_w_critico(p, g[n + "_in"]) == float(g[n + "_out"]), n
    capsys.readouterr()


def test_a_stale_library_is_refused_with_a_rebuild_message(monkeypatch):
    """lsm_version() is kept in step with the package (0.4.0 = 400) and checked when the library loads."""
    import lsm_speech_classifier_amd as pkg
    from lsm_speech_classifier_amd import _lib
    lib = _lib.load()
    major, minor, patch = (int(x) for x in pkg.__version__.split("."))
    assert lib.lsm_version() == _lib.ABI_VERSION == major * 10000 + minor * 100 + patch
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(_lib.LsmHipError, match="rebuild the extension"):
        _lib.load()


def test_train_test_split_permutation_is_the_pinned_one(golden_dir):
    """extract_lsm_features.py:160-162 splits with train_test_split(test_size=.2, random_state=42,
    stratify=y) and the first <= 500 training clips set w_critico (:40) -- the permutation is pinned for the class
    layouts in use (tests/make_golden_split.py), so another scikit-learn cannot reorder it unnoticed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import make_golden_split as mk
    g = np.load(os.path.join(golden_dir, "split.npz"))
    for name, (c, p) in mk.LAYOUTS.items():
        tr, te = mk.split(c, p)
        np.testing.assert_array_equal(tr, g[name + "_train"], err_msg=name)
        np.testing.assert_array_equal(te, g[name + "_test"], err_msg=name)
        assert len(tr) + len(te) == c * p and not set(tr) & set(te)
        # stratified: every class keeps 80 % of its clips (to rounding) in the training part
        per_class = np.bincount(np.repeat(np.arange(c), p)[tr], minlength=c)
        assert per_class.min() >= int(0.8 * p) - 1 and per_class.max() <= int(np.ceil(0.8 * p)) + 1
    # the w_critico head of the reference's corpus: 500 clips spread over all 12 classes, not the first class only
    head = g["ref12x1000_train"][:500]
    assert len(np.unique(head // 1000)) == 12


def test_main_leaves_the_process_group_when_file_1_is_missing(tmp_path, monkeypatch, capsys):
    """extract_lsm_features.main() on a missing File 1 returns through dist.finish()."""
    import extract_lsm_features as ex
    from lsm_speech_classifier_amd import dist as lsm_dist
    monkeypatch.chdir(tmp_path)
    calls = []
    monkeypatch.setattr(lsm_dist, "init", lambda *a, **k: (0, 0, 1))
    monkeypatch.setattr(lsm_dist, "finish", lambda: calls.append("finish"))
    assert ex.main("original", 0.6) is None
    assert calls == ["finish"] and "Dataset not found" in capsys.readouterr().out
    # only rank 0 prints the set-up lines
    with ex._rank0_only(1):
        print("never shown")
    with ex._rank0_only(0):
        print("shown")
    out = capsys.readouterr().out
    assert "never shown" not in out and "shown" in out


def test_missing_files_print_and_return(tmp_path, monkeypatch, capsys):
    import extract_lsm_features as ex
    import train_classifier as tc
    monkeypatch.chdir(tmp_path)
    assert ex.load_spike_dataset() == (None, None)
    assert tc.train_and_evaluate_classifier() is None
    out = capsys.readouterr().out
    assert "Dataset not found" in out and "Dataset file not found" in out


def test_load_audio_file_and_readout(tmp_path, monkeypatch, capsys):
    from scipy.io import wavfile
    import create_dataset as cd
    import train_classifier as tc
    rs = np.random.RandomState(0)
    wav = (rs.randn(8000) * 3000).astype(np.int16)
    wavfile.write(tmp_path / "a.wav", 16000, wav)
    a = cd.load_audio_file(tmp_path / "a.wav")
    assert a.shape == (16000,) and a.dtype == np.float32 and not a[8000:].any()
    np.testing.assert_allclose(a[:8000], wav / 32768.0, atol=1e-7)
    wavfile.write(tmp_path / "b.wav", 8000, np.stack([wav, wav], axis=1))          # stereo, 8 kHz
    assert cd.load_audio_file(tmp_path / "b.wav").shape == (16000,)
    assert cd.load_audio_file(tmp_path / "missing.wav") is None
    # readout on a separable toy feature file with File-2's schema
    monkeypatch.chdir(tmp_path)
    y = np.repeat(np.arange(3), 40).astype(np.int32)
    X = rs.randn(120, 10) + y[:, None] * 3.0
    np.savez_compressed("lsm_features_larger.npz", X_train_features=X[::2], y_train=y[::2],
                        X_test_features=X[1::2], y_test=y[1::2], feature_set="original",
                        leak_variance_divisor=None)
    acc = tc.train_and_evaluate_classifier()
    assert acc > 0.9
    capsys.readouterr()


def test_train_classifier_torch_readouts(tmp_path, monkeypatch, capsys):
    import train_classifier as tc
    rs = np.random.RandomState(1)
    monkeypatch.chdir(tmp_path)
    y = np.repeat(np.arange(4), 30).astype(np.int32)
    X = rs.randn(120, 12) + (rs.randn(4, 12) * 3.0)[y]      # class means in general position
    np.savez_compressed("lsm_features_larger.npz", X_train_features=X[::2], y_train=y[::2],
                        X_test_features=X[1::2], y_test=y[1::2], feature_set="original",
                        leak_variance_divisor=None)
    base = tc.train_and_evaluate_classifier()
    for kind in ("torch-logistic", "torch-ridge"):
        acc = tc.train_and_evaluate_classifier(readout=kind)
        assert acc > 0.9 and abs(acc - base) <= 0.1
    capsys.readouterr()
