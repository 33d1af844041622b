"""Stage 1 of the pipeline: wav clips -> spike-train dataset (File 1).

Drop-in for the reference script of the same name (same CLI flags, function names, argument
meaning and ``speech_spike_dataset_pure_redundancy.npz`` schema; /root/reference/create_dataset.py),
with the filterbank, normalise/resize and hysteresis encoder running as HIP kernels on an MI355X
(``lsm_speech_classifier_amd.frontend``), batched over clips instead of one clip per Python
iteration.  There is no CPU fallback for those stages.
"""
import argparse
import os
import warnings
from pathlib import Path

import numpy as np

SAMPLE_RATE = 16000
DURATION = 1.0
TIME_BINS = 100
SPIKE_THRESHOLDS = [0.70, 0.80, 0.90, 0.95]
HYSTERESIS_GAP = 0.1
MAX_SAMPLES_PER_CLASS = 1000
REDUNDANCY_FACTOR = 1
COMMANDS = ["yes", "no", "up", "visual", "backward", "stop", "bird", "cat", "nine", "eight",
            "zero", "follow"]
DATASET_ROOT = Path("speech_commands_v0.02")
OUTPUT_FILE = "speech_spike_dataset_pure_redundancy.npz"
ENCODE_BATCH = 2048          # clips per GPU launch

np.random.seed(42)


def _frontend():
    from lsm_speech_classifier_amd import frontend
    return frontend


def load_audio_file(filepath: Path):
    """One wav file as mono float32 at 16 kHz, padded/trimmed to exactly one second; None (with a
    message) when the file cannot be read.  PCM wav only (scipy.io.wavfile); other rates are
    resampled polyphase."""
    from scipy.io import wavfile
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rate, data = wavfile.read(str(filepath))
        if data.dtype.kind == "i":
            data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
        elif data.dtype.kind == "u":
            data = (data.astype(np.float32) - 128.0) / 128.0
        else:
            data = data.astype(np.float32)
        if data.ndim == 2:
            data = data.mean(axis=1)
        if rate != SAMPLE_RATE:
            from math import gcd
            from scipy.signal import resample_poly
            g = gcd(int(rate), SAMPLE_RATE)
            data = resample_poly(data, SAMPLE_RATE // g, int(rate) // g).astype(np.float32)
        want = int(SAMPLE_RATE * DURATION)
        data = data[:want]
        if len(data) < want:
            data = np.pad(data, (0, want - len(data)))
        return np.ascontiguousarray(data, dtype=np.float32)
    except Exception as exc:
        print(f"Error loading {filepath}: {exc}")
        return None


def audio_to_spectrogram(audio: np.ndarray, n_filters: int, filterbank: str) -> np.ndarray:
    """(n_samples,) -> (n_filters, TIME_BINS) normalised spectrogram in [0, 1] (GPU)."""
    return _frontend().audio_to_spectrogram(audio, n_filters, filterbank)


def convert_spectrogram_to_spikes_hysteresis(spectrogram, thresholds, hysteresis_gap=0.05):
    """(F, n_time) -> (F, n_time * len(thresholds)) uint8, thresholds interleaved in time (GPU)."""
    return _frontend().convert_spectrogram_to_spikes_hysteresis(spectrogram, thresholds, hysteresis_gap)


def create_pure_redundancy(spike_train: np.ndarray, redundancy_factor: int) -> np.ndarray:
    return np.repeat(spike_train, redundancy_factor, axis=0)


def _collect_audio(commands, root: Path, per_class: int):
    clips, labels = [], []
    for label, word in enumerate(commands):
        print(f"Processing '{word}'...")
        folder = root / word
        if not folder.is_dir():
            print(f"  Warning: Directory not found, skipping: {folder}")
            continue
        files = sorted(folder.glob("*.wav"))[:per_class]
        if not files:
            print(f"  Warning: No files found for '{word}'")
            continue
        for f in files:
            audio = load_audio_file(f)
            if audio is not None:
                clips.append(audio)
                labels.append(label)
    return clips, labels


def _synthetic_audio(commands, per_class: int):
    from lsm_speech_classifier_amd import synth
    labels = np.repeat(np.arange(len(commands)), per_class)
    return list(synth.class_chirps(labels, seed=1234)), list(labels)


def collect_audio(commands=None, dataset_root=None, max_per_class: int = MAX_SAMPLES_PER_CLASS,
                  synthetic_per_class: int = 0):
    """The clips create_dataset() would encode, as arrays: (n, 16000) float32 and int32 labels (label =
    position in the class list) -- the entry of the in-memory route (extract_lsm_features.main_from_audio),
    which skips File 1.  Empty arrays when nothing could be read."""
    commands = list(COMMANDS if commands is None else commands)
    root = Path(DATASET_ROOT if dataset_root is None else dataset_root)
    if synthetic_per_class > 0:
        clips, labels = _synthetic_audio(commands, synthetic_per_class)
    else:
        clips, labels = _collect_audio(commands, root, max_per_class)
    if not clips:
        return np.zeros((0, int(SAMPLE_RATE * DURATION)), dtype=np.float32), np.zeros(0, dtype=np.int32)
    return np.stack(clips).astype(np.float32, copy=False), np.asarray(labels, dtype=np.int32)


def create_dataset(n_filters: int, filterbank: str, commands=None, dataset_root=None,
                   max_per_class: int = MAX_SAMPLES_PER_CLASS, synthetic_per_class: int = 0,
                   output_file: str = OUTPUT_FILE, packed: bool = False):
    """Build File 1.  The first two arguments are the reference's; the keyword arguments expose
    what the reference hard-codes (class list, corpus folder, per-class cap) plus a synthetic
    corpus for machines without Speech Commands.  ``packed=True`` writes the bit-packed schema of
    ``lsm_speech_classifier_amd.spikefile`` (rasters packed on the GPU, 8x fewer bytes off the
    device and on disk); the default is the reference's uint8 schema."""
    commands = list(COMMANDS if commands is None else commands)
    root = Path(DATASET_ROOT if dataset_root is None else dataset_root)
    print(f"Creating dataset with filterbank: {filterbank}, filters: {n_filters}")
    if synthetic_per_class > 0:
        clips, labels = _synthetic_audio(commands, synthetic_per_class)
    else:
        clips, labels = _collect_audio(commands, root, max_per_class)
    if not clips:
        print("\nERROR: No audio files were successfully processed.")
        return

    fe = _frontend().SpikeFrontEnd(n_filters, filterbank, redundancy=REDUNDANCY_FACTOR,
                                   thresholds=SPIKE_THRESHOLDS, gap=HYSTERESIS_GAP,
                                   time_bins=TIME_BINS, n_samples=int(SAMPLE_RATE * DURATION))
    from lsm_speech_classifier_amd import spikefile
    parts, n_spikes = [], 0
    for lo in range(0, len(clips), ENCODE_BATCH):
        batch = np.stack(clips[lo:lo + ENCODE_BATCH])
        raster = fe.encode(batch)
        n_spikes += int(raster.count_nonzero())
        parts.append((_frontend().pack_raster(raster) if packed else raster).cpu().numpy())
    X = np.concatenate(parts).astype(np.uint8, copy=False)
    y_labels = np.asarray(labels, dtype=np.int32)

    print("\nDataset created successfully.")
    print(f"  Shape: {(len(X), fe.n_channels, fe.n_steps)}" + (" (bit-packed on disk)" if packed else ""))
    print(f"  Avg spikes per sample: {n_spikes / len(X):.1f}")
    if packed:
        spikefile.save(output_file, packed=X, time_steps=fe.n_steps, y_labels=y_labels)
    else:
        spikefile.save(output_file, X_spikes=X, y_labels=y_labels)
    print(f"Saved to '{output_file}'")


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Create a spike train dataset from audio files.")
    ap.add_argument("--n-filters", type=int, default=128, help="Number of filters for the filterbank.")
    ap.add_argument("--filterbank", type=str, default="gammatone", choices=["mel", "gammatone"],
                    help="Type of filterbank to use.")
    ap.add_argument("--synthetic-per-class", type=int,
                    default=int(os.environ.get("LSM_SYNTHETIC_PER_CLASS", "0")),
                    help="Generate this many synthetic clips per class instead of reading wav files.")
    ap.add_argument("--packed", action="store_true",
                    default=os.environ.get("LSM_PACKED_DATASET", "0") == "1",
                    help="Write the bit-packed File 1 schema (8x fewer raster bytes).")
    a = ap.parse_args()
    create_dataset(n_filters=a.n_filters, filterbank=a.filterbank,
                   synthetic_per_class=a.synthetic_per_class, packed=a.packed)
