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


def _list_files(commands, root: Path, per_class: int, verbose: bool = True):
    """The (wav path, label) pairs the reference's loop visits, in its order: classes in list order, the first
    `per_class` files of each folder in sorted order (create_dataset.py:130-143)."""
    listing = []
    for label, word in enumerate(commands):
        if verbose:
            print(f"Processing '{word}'...")
        folder = root / word
        if not folder.is_dir():
            if verbose:
                print(f"  Warning: Directory not found, skipping: {folder}")
            continue
        files = sorted(folder.glob("*.wav"))[:per_class]
        if not files and verbose:
            print(f"  Warning: No files found for '{word}'")
        listing += [(f, label) for f in files]
    return listing


def _load_listing(listing):
    clips, labels = [], []
    for f, label in listing:
        audio = load_audio_file(f)
        if audio is not None:
            clips.append(audio)
            labels.append(label)
    return clips, labels


def _collect_audio(commands, root: Path, per_class: int):
    return _load_listing(_list_files(commands, root, per_class))


def _synthetic_audio(commands, per_class: int):
    from lsm_speech_classifier_amd import synth
    labels = np.repeat(np.arange(len(commands)), per_class)
    return list(synth.class_chirps(labels, seed=1234)), list(labels)


def read_commands_file(path) -> list:
    """One class name per line (blank lines and #-comments skipped)."""
    with open(path) as fh:
        return [ln.strip() for ln in fh if ln.strip() and not ln.lstrip().startswith("#")]


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
    device and on disk); the default is the reference's uint8 schema.

    Under a launcher (torchrun: RANK / WORLD_SIZE) the clip loop of create_dataset.py:143 shards: rank r reads
    and encodes the r-th contiguous block of the file listing on its own GPU, the raster blocks are all-gathered
    in rank order (= the single-process order) and rank 0 writes the file -- the same bytes as one process."""
    from lsm_speech_classifier_amd import dist as lsm_dist
    rank, _, world = lsm_dist.init()
    commands = list(COMMANDS if commands is None else commands)
    root = Path(DATASET_ROOT if dataset_root is None else dataset_root)
    if rank == 0:
        print(f"Creating dataset with filterbank: {filterbank}, filters: {n_filters}"
              + (f" ({world} ranks)" if world > 1 else ""))
    if synthetic_per_class > 0:
        clips, labels = _synthetic_audio(commands, synthetic_per_class)
        lo, hi = lsm_dist.shard_range(len(clips), rank, world)
        clips, labels = clips[lo:hi], labels[lo:hi]
    else:
        listing = _list_files(commands, root, max_per_class, verbose=rank == 0)
        lo, hi = lsm_dist.shard_range(len(listing), rank, world)
        clips, labels = _load_listing(listing[lo:hi])
    if not clips and world == 1:
        print("\nERROR: No audio files were successfully processed.")
        return

    import torch
    dev = lsm_dist.local_device() if world > 1 else None
    fe = _frontend().SpikeFrontEnd(n_filters, filterbank, device=dev, redundancy=REDUNDANCY_FACTOR,
                                   thresholds=SPIKE_THRESHOLDS, gap=HYSTERESIS_GAP,
                                   time_bins=TIME_BINS, n_samples=int(SAMPLE_RATE * DURATION))
    from lsm_speech_classifier_amd import spikefile
    parts, n_spikes = [], 0
    for a in range(0, len(clips), ENCODE_BATCH):
        batch = np.stack(clips[a:a + ENCODE_BATCH])
        raster = fe.encode(batch)
        n_spikes += int(raster.count_nonzero())
        part = _frontend().pack_raster(raster) if packed else raster
        parts.append(part if world > 1 else part.cpu().numpy())
    if world > 1:
        # one exchange: every rank's rows (device, flattened) and labels, in rank order
        width = fe.n_channels * ((fe.n_steps + 7) // 8 if packed else fe.n_steps)
        local = (torch.cat(parts).reshape(len(clips), width) if parts
                 else torch.empty((0, width), dtype=torch.uint8, device=fe.device))
        rows = lsm_dist.gather_varrows(local)
        y_all = lsm_dist.gather_varrows(torch.tensor(labels, dtype=torch.int32, device=fe.device).reshape(-1, 1))
        tot = lsm_dist.gather_varrows(torch.tensor([[n_spikes]], dtype=torch.int64, device=fe.device))
        lsm_dist.finish()
        if rank != 0:
            return
        if rows.shape[0] == 0:
            print("\nERROR: No audio files were successfully processed.")
            return
        X = rows.cpu().numpy().reshape(rows.shape[0], fe.n_channels, -1)
        y_labels = y_all.cpu().numpy().reshape(-1).astype(np.int32)
        n_spikes = int(tot.sum())
    else:
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


def add_corpus_flags(ap):
    """What the reference hard-codes at create_dataset.py:15,108-120 as flags, defaults = the reference's values
    (shared with main.py, which forwards them)."""
    ap.add_argument("--commands", type=str, default=None,
                    help="Comma-separated class list (default: the reference's 12 words).")
    ap.add_argument("--commands-file", type=str, default=None, help="File with one class name per line.")
    ap.add_argument("--dataset-root", type=str, default=None,
                    help=f"Corpus folder with one sub-folder per class (default: {DATASET_ROOT}).")
    ap.add_argument("--max-per-class", type=int, default=MAX_SAMPLES_PER_CLASS,
                    help="Per-class cap on the sorted file list.")
    ap.add_argument("--synthetic-per-class", type=int,
                    default=int(os.environ.get("LSM_SYNTHETIC_PER_CLASS", "0")),
                    help="Generate this many synthetic clips per class instead of reading wav files.")


def commands_from_args(a):
    if a.commands_file:
        return read_commands_file(a.commands_file)
    if a.commands:
        return [w.strip() for w in a.commands.split(",") if w.strip()]
    return None


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Create a spike train dataset from audio files.")
    ap.add_argument("--n-filters", type=int, default=128, help="Number of filters for the filterbank.")
    ap.add_argument("--filterbank", type=str, default="gammatone", choices=["mel", "gammatone"],
                    help="Type of filterbank to use.")
    add_corpus_flags(ap)
    ap.add_argument("--packed", action="store_true",
                    default=os.environ.get("LSM_PACKED_DATASET", "0") == "1",
                    help="Write the bit-packed File 1 schema (8x fewer raster bytes).")
    a = ap.parse_args()
    create_dataset(n_filters=a.n_filters, filterbank=a.filterbank, commands=commands_from_args(a),
                   dataset_root=a.dataset_root, max_per_class=a.max_per_class,
                   synthetic_per_class=a.synthetic_per_class, packed=a.packed)
