"""the constants the reference hard-codes (/root/reference/extract_lsm_features.py:10-16,
create_dataset.py:15,108-120, train_classifier.py:8-20) are flags of the drop-in scripts with unchanged defaults,
and main.py forwards them to the stages.  CPU only: nothing here starts a stage for real."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_reservoir_shape_defaults_follow_the_reference_constants():
    import extract_lsm_features as ex
    assert ex.reservoir_shape() == (1000, 400, 200)
    assert ex.reservoir_shape(500) == (500, 200, 100)                  # BASELINE configs[0]
    assert ex.reservoir_shape(4000) == (4000, 1600, 800)               # configs[3]
    assert ex.reservoir_shape(8000) == (8000, 3200, 1600)              # configs[4]
    assert ex.reservoir_shape(1000, 250, 60) == (1000, 250, 60)
    assert ex.reservoir_shape(None, None, 50) == (1000, 400, 50)
    with pytest.raises(ValueError):
        ex.reservoir_shape(100, 101)
    p = ex._simulation_params(None, None, 500, None, None, 7)
    assert (p.num_neurons, p.num_output_neurons, p.small_world_graph_k, p.seed) == (500, 200, 100, 7)
    q = ex._simulation_params(None, 4.0, None, None, None, None)
    assert (q.num_neurons, q.num_output_neurons, q.small_world_graph_k, q.seed) == (1000, 400, 200, 42)
    assert q.leak_variance_divisor == 4.0 and q.membrane_threshold == 2.0 and q.refractory_period == 2


@pytest.mark.parametrize("script,flags", [
    ("create_dataset.py", ["--commands", "--commands-file", "--dataset-root", "--max-per-class", "--n-filters",
                           "--filterbank", "--synthetic-per-class", "--packed"]),
    ("extract_lsm_features.py", ["--num-neurons", "--num-output-neurons", "--small-world-k", "--seed",
                                 "--feature-set", "--multiplier", "--leak-variance-divisor"]),
    ("train_classifier.py", ["--readout", "--commands", "--commands-file"]),
    ("main.py", ["--n-filters", "--filterbank", "--feature-set", "--multiplier", "--in-memory", "--commands",
                 "--commands-file", "--dataset-root", "--max-per-class", "--num-neurons", "--num-output-neurons",
                 "--small-world-k", "--seed", "--readout", "--nproc"]),
])
def test_every_script_offers_the_flags(script, flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, script), "--help"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr
    for f in flags:
        assert f in out.stdout, f"{script} lacks {f}"


def test_main_forwards_the_constants_to_the_stages(monkeypatch, tmp_path):
    import main as pipeline
    calls = []
    monkeypatch.setattr(pipeline.subprocess, "call", lambda cmd, **kw: calls.append(list(cmd)) or 0)
    monkeypatch.delenv("LSM_SYNTHETIC_PER_CLASS", raising=False)
    # the reference's call: nothing but its own four flags reaches the stages
    pipeline.run_pipeline(128, "gammatone", "original", 0.6)
    assert [c[1:] for c in calls] == [
        [os.path.join(ROOT, "create_dataset.py"), "--n-filters", "128", "--filterbank", "gammatone"],
        [os.path.join(ROOT, "extract_lsm_features.py"), "--feature-set", "original", "--multiplier", "0.6"],
        [os.path.join(ROOT, "train_classifier.py")]]
    calls.clear()
    # BASELINE configs[3]: 35 classes, 4000 neurons, ridge readout, two ranks
    words = tmp_path / "words.txt"
    words.write_text("\n".join(f"w{i}" for i in range(35)))
    pipeline.run_pipeline(128, "gammatone", "original", 0.6, commands_file=str(words), dataset_root="corpus",
                          max_per_class=50, num_neurons=4000, seed=3, readout="torch-ridge", nproc=2)
    s1, s2, s3 = calls
    assert s1[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in s1
    tail = s1[s1.index(os.path.join(ROOT, "create_dataset.py")):]
    assert tail[1:] == ["--n-filters", "128", "--filterbank", "gammatone", "--commands-file", str(words),
                        "--dataset-root", "corpus", "--max-per-class", "50"]
    tail = s2[s2.index(os.path.join(ROOT, "extract_lsm_features.py")):]
    assert tail[1:] == ["--feature-set", "original", "--multiplier", "0.6", "--num-neurons", "4000", "--seed", "3"]
    assert s3[1:] == [os.path.join(ROOT, "train_classifier.py"), "--readout", "torch-ridge",
                      "--commands-file", str(words)]                  # the readout stays one process
    with pytest.raises(TypeError):
        pipeline.run_pipeline(128, "gammatone", "original", 0.6, no_such_flag=1)


def test_commands_parsing(tmp_path):
    import argparse
    import create_dataset as cd
    f = tmp_path / "c.txt"
    f.write_text("# speech commands\nyes\n\n no \nup\n")
    assert cd.read_commands_file(f) == ["yes", "no", "up"]
    ns = argparse.Namespace(commands="a, b,,c", commands_file=None)
    assert cd.commands_from_args(ns) == ["a", "b", "c"]
    assert cd.commands_from_args(argparse.Namespace(commands=None, commands_file=str(f))) == ["yes", "no", "up"]
    assert cd.commands_from_args(argparse.Namespace(commands=None, commands_file=None)) is None


def test_file_listing_order_and_cap(tmp_path, capsys):
    """create_dataset.py:130-143: classes in list order, sorted file names, per-class cap, missing folders skipped."""
    import create_dataset as cd
    for word, names in (("b", ["3.wav", "1.wav", "2.wav", "x.txt"]), ("a", ["9.wav"]), ("empty", [])):
        (tmp_path / word).mkdir()
        for n in names:
            (tmp_path / word / n).write_bytes(b"")
    listing = cd._list_files(["b", "missing", "a", "empty"], tmp_path, 2)
    assert [(p.parent.name, p.name, lab) for p, lab in listing] == [("b", "1.wav", 0), ("b", "2.wav", 0), ("a", "9.wav", 2)]
    out = capsys.readouterr().out
    assert "Directory not found" in out and "No files found for 'empty'" in out
    assert cd._list_files(["b"], tmp_path, 2, verbose=False) == listing[:2] and capsys.readouterr().out == ""
