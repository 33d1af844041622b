"""§8(f)-1 on the GPU box: class directories of PCM .wav files through create_dataset() -> X_spikes, against
the oracle on independently decoded audio.  Mirrors /root/reference/create_dataset.py:22-36 (load, mono,
16 kHz, pad/trim to one second, errors -> skipped) and :121-162 (class walk, sorted glob, per-class cap,
missing folders, label = position in the class list)."""
import os

import numpy as np
import pytest
from scipy.io import wavfile

pytestmark = pytest.mark.gpu

THR = [0.70, 0.80, 0.90, 0.95]
GAP = 0.1


def _tone(rate, seconds, f0, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(int(rate * seconds)) / rate
    x = 0.4 * np.sin(2 * np.pi * (f0 + 300 * t) * t) + 0.02 * rng.standard_normal(len(t))
    return x


def test_wav_corpus_through_create_dataset(tmp_path, monkeypatch, oracle_c, capsys):
    import create_dataset as cd
    from oracle import ref_numpy as O
    root = tmp_path / "speech_commands_v0.02"
    words = ["yes", "no", "up", "ghost"]                       # "ghost" has no folder: warned about and skipped
    decoded = {}                                               # (word, file) -> float32 (16000,)

    def put(word, name, rate, pcm):                            # pcm: int16 (n,) or (n, 2)
        (root / word).mkdir(parents=True, exist_ok=True)
        wavfile.write(str(root / word / name), rate, pcm)
        x = pcm.astype(np.float32) / 32768.0
        if x.ndim == 2:
            x = x.mean(axis=1)
        if rate != 16000:
            from scipy.signal import resample_poly
            x = resample_poly(x, 16000 // np.gcd(rate, 16000), rate // np.gcd(rate, 16000)).astype(np.float32)
        x = x[:16000]
        decoded[(word, name)] = np.ascontiguousarray(np.pad(x, (0, 16000 - len(x))), dtype=np.float32)

    i16 = lambda x: np.clip(np.round(x * 32767), -32768, 32767).astype(np.int16)
    put("yes", "b_exact.wav", 16000, i16(_tone(16000, 1.0, 300, 1)))
    put("yes", "a_short.wav", 16000, i16(_tone(16000, 0.6, 500, 2)))                 # padded with zeros
    put("yes", "c_long.wav", 16000, i16(_tone(16000, 1.4, 700, 3)))                  # trimmed
    put("yes", "d_beyond_cap.wav", 16000, i16(_tone(16000, 1.0, 900, 4)))            # 4th of a cap of 3: not read
    put("no", "a_stereo_8k.wav", 8000, np.stack([i16(_tone(8000, 1.0, 400, 5)), i16(_tone(8000, 1.0, 650, 6))], axis=1))
    (root / "no" / "b_broken.wav").write_bytes(b"RIFF\x00\x00this is not a wav file")  # unreadable: skipped
    put("no", "c_ok.wav", 16000, i16(_tone(16000, 1.0, 1200, 7)))
    put("up", "only.wav", 16000, i16(_tone(16000, 1.0, 2000, 8)))
    (root / "up" / "notes.txt").write_text("not audio")                              # not matched by *.wav

    monkeypatch.chdir(tmp_path)
    cd.create_dataset(64, "gammatone", commands=words, dataset_root=root, max_per_class=3)
    out = capsys.readouterr().out
    assert "Error loading" in out and "b_broken.wav" in out and "Directory not found" in out and "ghost" in out
    with np.load(cd.OUTPUT_FILE) as d:
        X, y = d["X_spikes"], d["y_labels"]
    # sorted glob, cap 3, broken file skipped, label = index in the class list
    expect = [("yes", "a_short.wav", 0), ("yes", "b_exact.wav", 0), ("yes", "c_long.wav", 0),
              ("no", "a_stereo_8k.wav", 1), ("no", "c_ok.wav", 1), ("up", "only.wav", 2)]
    assert X.shape == (len(expect), 64, 400) and X.dtype == np.uint8 and y.dtype == np.int32
    assert y.tolist() == [lab for _, _, lab in expect]
    coefs = O.gammatone_coefs(16000, 64, 50)
    for row, (word, name, _) in enumerate(expect):
        a = decoded[(word, name)]
        got = cd.load_audio_file(root / word / name)
        np.testing.assert_array_equal(got, a, err_msg=name)                          # the decoder itself
        ref = oracle_c.encode_hysteresis(oracle_c.normalise_resize(oracle_c.gammatone_db(
            oracle_c.gammatone_spec(a, coefs, 400, 160, 98))), THR, GAP)
        np.testing.assert_array_equal(X[row], ref, err_msg=name)
    assert X.sum() > 0 and cd.load_audio_file(root / "no" / "b_broken.wav") is None
    # an empty corpus prints the reference's error and writes nothing
    os.remove(cd.OUTPUT_FILE)
    cd.create_dataset(64, "gammatone", commands=["ghost"], dataset_root=root)
    assert "No audio files were successfully processed" in capsys.readouterr().out and not os.path.exists(cd.OUTPUT_FILE)
