"""VERDICT r2 #2 and #3 on the GPU box: (a) the WHOLE path shards in the product -- `create_dataset.py` and the
in-memory route under a launcher give the same File 1 / File 2, bit for bit, as one process (ranks share the box's
one GPU, rows travel through gloo: LSM_SHARE_GPU / LSM_DIST_BACKEND); (b) BASELINE configs[0]- and configs[3]-shaped
runs are expressible through main.py's flags and their File 2 equals the oracle-derived arrays.
Reference loops: /root/reference/create_dataset.py:143, extract_lsm_features.py:78; constants :10-16 / :15,108-120."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT",
                                                            "LSM_SYNTHETIC_PER_CLASS")}
    env.update(extra)
    return env


def _run_ranks(world, argv, cwd):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = _clean_env(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                         MASTER_PORT=str(port), LSM_SHARE_GPU="1", LSM_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, cwd=cwd, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    return outs


def _wav_corpus(root, words, per_class):
    """PCM corpus with one unreadable file and one missing folder (skip-on-error must survive sharding)."""
    from scipy.io import wavfile
    from lsm_speech_classifier_amd import synth
    for c, w in enumerate(words):
        if w == "missing":
            continue
        (root / w).mkdir(parents=True)
        audio = synth.class_chirps([c] * per_class, seed=50 + c)
        for i, a in enumerate(audio):
            wavfile.write(str(root / w / f"{i:03d}.wav"), 16000, np.round(a * 32767).clip(-32768, 32767).astype(np.int16))
    (root / words[0] / "001.wav").write_bytes(b"not a wav file")


@pytest.mark.parametrize("world", [2, 3])
def test_create_dataset_under_a_launcher_writes_the_same_file_1(tmp_path, world):
    import create_dataset as cd
    words = ["yes", "no", "missing", "up"]
    corpus = tmp_path / "corpus"
    _wav_corpus(corpus, words, 5)
    single, multi = tmp_path / "single", tmp_path / "multi"
    single.mkdir(); multi.mkdir()
    base = ["--n-filters", "64", "--commands", ",".join(words), "--dataset-root", str(corpus), "--max-per-class", "4"]
    for packed in ([], ["--packed"]):
        argv = [os.path.join(ROOT, "create_dataset.py")] + base + packed
        out = subprocess.run([sys.executable] + argv, env=_clean_env(), cwd=single, capture_output=True, text=True,
                             timeout=600)
        assert out.returncode == 0 and "Saved to" in out.stdout, out.stdout + out.stderr
        outs = _run_ranks(world, argv, multi)
        assert sum("Saved to" in o for o in outs) == 1                       # rank 0 alone writes
        with np.load(single / cd.OUTPUT_FILE) as a, np.load(multi / cd.OUTPUT_FILE) as b:
            assert sorted(a.files) == sorted(b.files)
            for k in a.files:
                np.testing.assert_array_equal(a[k], b[k], err_msg=k)
            n = len(a["y_labels"])
        assert n == 3 * 4 - 1                                                # three folders x cap 4, one unreadable
        os.remove(single / cd.OUTPUT_FILE); os.remove(multi / cd.OUTPUT_FILE)
    # the synthetic corpus shards the same way (5 clips over `world` ranks: ragged, and at 3 ranks for 2 clips empty)
    for per_class in (5, 1):
        argv = [os.path.join(ROOT, "create_dataset.py"), "--n-filters", "40", "--filterbank", "mel", "--commands",
                "a,b", "--synthetic-per-class", str(per_class)]
        subprocess.run([sys.executable] + argv, env=_clean_env(), cwd=single, check=True, capture_output=True, timeout=600)
        _run_ranks(world, argv, multi)
        with np.load(single / cd.OUTPUT_FILE) as a, np.load(multi / cd.OUTPUT_FILE) as b:
            np.testing.assert_array_equal(a["X_spikes"], b["X_spikes"])
            np.testing.assert_array_equal(a["y_labels"], b["y_labels"])
            assert a["X_spikes"].shape == (2 * per_class, 40, 400) and a["X_spikes"].any()


def test_in_memory_route_with_nproc_2_writes_the_same_file_2(tmp_path):
    """main.py --in-memory --nproc 2 (torch.distributed.run, one rank per GPU; here both on cuda:0): File 2 equals
    the single-process in-memory run, which tests/test_gpu_hotpath.py ties to the npz route."""
    import extract_lsm_features as ex
    args = ["--in-memory", "--n-filters", "64", "--commands", "yes,no,up", "--synthetic-per-class", "11",
            "--num-neurons", "600", "--seed", "5"]
    files = {}
    for name, extra, env in (("single", [], _clean_env()),
                             ("multi", ["--nproc", "2"], _clean_env(LSM_SHARE_GPU="1", LSM_DIST_BACKEND="gloo",
                                                                    LSM_MASTER_PORT=str(_free_port())))):
        d = tmp_path / name
        d.mkdir()
        out = subprocess.run([sys.executable, os.path.join(ROOT, "main.py")] + args + extra, env=env, cwd=d,
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "Test Accuracy" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
        assert not (d / "speech_spike_dataset_pure_redundancy.npz").exists()
        with np.load(d / ex.FEATURE_FILE, allow_pickle=True) as f:
            files[name] = {k: f[k] for k in f.files}
    assert files["single"]["X_train_features"].shape == (26, 5 * 240)          # N = 600 -> 240 output neurons
    for k in ("X_train_features", "X_test_features", "y_train", "y_test"):
        np.testing.assert_array_equal(files["single"][k], files["multi"][k], err_msg=k)


def test_main_py_nproc_2_file_route_writes_the_same_files(tmp_path):
    """`main.py --nproc 2` (the documented multi-GPU command): stages 1 and 2 each as two ranks through
    torch.distributed.run, the readout as one process; File 1 and File 2 equal the single-process run's."""
    import create_dataset as cd
    import extract_lsm_features as ex
    args = ["--n-filters", "64", "--commands", "yes,no,up,down", "--synthetic-per-class", "9", "--num-neurons", "700",
            "--feature-set", "rate"]
    out = {}
    for name, extra, env in (("single", [], _clean_env()),
                             ("multi", ["--nproc", "2"], _clean_env(LSM_SHARE_GPU="1", LSM_DIST_BACKEND="gloo",
                                                                    LSM_MASTER_PORT=str(_free_port())))):
        d = tmp_path / name
        d.mkdir()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py")] + args + extra, env=env, cwd=d,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "Test Accuracy" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
        assert r.stdout.count("Saved to") == 1 and r.stdout.count("Extraction complete") == 1     # rank 0 alone writes
        with np.load(d / cd.OUTPUT_FILE) as f1, np.load(d / ex.FEATURE_FILE, allow_pickle=True) as f2:
            out[name] = ({k: f1[k] for k in f1.files}, {k: f2[k] for k in ("X_train_features", "X_test_features", "y_train", "y_test")})
    for part in (0, 1):
        for k in out["single"][part]:
            np.testing.assert_array_equal(out["single"][part][k], out["multi"][part][k], err_msg=k)
    assert out["single"][0]["X_spikes"].shape == (36, 64, 400) and out["single"][1]["X_train_features"].shape == (28, 3 * 280)


def _oracle_file_2(oracle_c, X, y, n, n_out, k, n_channels, multiplier=0.6):
    from lsm_speech_classifier_amd import reservoir as R
    from oracle import ref_numpy as O
    from sklearn.model_selection import train_test_split
    from sklearn.preprocessing import StandardScaler
    import extract_lsm_features as ex
    X_train, X_test, y_train, y_test = train_test_split(X, y, test_size=0.2, random_state=42, stratify=y)
    p = R.SimulationParams(num_neurons=n, num_output_neurons=n_out, small_world_graph_k=k)
    p.mean_weight = O.w_critico(k, 2.0, 2, X_train) * multiplier
    res = R.build_reservoir(p, n_channels)
    keys = ex.FEATURE_SETS["original"]
    threads = min(16, os.cpu_count() or 1)
    f_train = oracle_c.lif_run_batch(res, X_train, keys, n_threads=threads)
    f_test = oracle_c.lif_run_batch(res, X_test, keys, n_threads=threads)
    sc = StandardScaler()
    return sc.fit_transform(f_train), sc.transform(f_test), y_train, y_test, f_train


def test_main_py_runs_a_cfg1_shaped_pipeline(tmp_path, oracle_c):
    """BASELINE configs[0]: 4 classes, 40 mel filters, 500-neuron reservoir -- through main.py's flags alone."""
    from oracle import ref_numpy as O
    out = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--n-filters", "40", "--filterbank", "mel",
                          "--num-neurons", "500", "--commands", "yes,no,up,down", "--synthetic-per-class", "15"],
                         env=_clean_env(), cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "Test Accuracy" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    assert "down" in out.stdout                                              # the report names the classes given
    with np.load(tmp_path / "speech_spike_dataset_pure_redundancy.npz") as d:
        X, y = d["X_spikes"], d["y_labels"]
    assert X.shape == (60, 40, 400) and list(np.unique(y)) == [0, 1, 2, 3]
    from lsm_speech_classifier_amd import synth
    audio = synth.class_chirps(np.repeat(np.arange(4), 15), seed=1234)
    for b in (0, 31, 59):                                                    # stage 1 against the NumPy mel oracle
        ref = O.encode_hysteresis(O.normalise_resize(O.mel_db(audio[b], 40)), [0.70, 0.80, 0.90, 0.95], 0.1)
        np.testing.assert_array_equal(X[b], ref)
    Xtr, Xte, ytr, yte, f_train = _oracle_file_2(oracle_c, X, y, 500, 200, 100, 40)
    with np.load(tmp_path / "lsm_features_larger.npz", allow_pickle=True) as d:
        assert d["X_train_features"].shape == (48, 5 * 200)
        np.testing.assert_array_equal(d["X_train_features"], Xtr)
        np.testing.assert_array_equal(d["X_test_features"], Xte)
        np.testing.assert_array_equal(d["y_train"], ytr)
        np.testing.assert_array_equal(d["y_test"], yte)
    assert f_train[:, :200].sum() > 0


def test_main_py_runs_a_cfg4_shaped_pipeline(tmp_path, oracle_c):
    """BASELINE configs[3]: 35 classes, 128 gammatone filters, 4000-neuron reservoir (ring-row kernel), ridge readout."""
    from oracle import ref_numpy as O
    words = tmp_path / "words.txt"
    words.write_text("\n".join(f"word{i:02d}" for i in range(35)) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--num-neurons", "4000", "--commands-file",
                          str(words), "--synthetic-per-class", "5", "--readout", "torch-ridge"],
                         env=_clean_env(), cwd=tmp_path, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and "Test Accuracy" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    assert "ridge" in out.stdout and "word34" in out.stdout
    with np.load(tmp_path / "speech_spike_dataset_pure_redundancy.npz") as d:
        X, y = d["X_spikes"], d["y_labels"]
    assert X.shape == (175, 128, 400) and len(np.unique(y)) == 35
    Xtr, Xte, ytr, yte, f_train = _oracle_file_2(oracle_c, X, y, 4000, 1600, 800, 128)
    with np.load(tmp_path / "lsm_features_larger.npz", allow_pickle=True) as d:
        assert d["X_train_features"].shape == (140, 5 * 1600)
        np.testing.assert_array_equal(d["X_train_features"], Xtr)
        np.testing.assert_array_equal(d["X_test_features"], Xte)
        np.testing.assert_array_equal(d["y_train"], ytr)
    assert f_train[:, :1600].sum() > 0
