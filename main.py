"""Run the three pipeline stages in order, each as its own Python process (like the reference's
main.py, which shells out and ignores exit codes: /root/reference/main.py:19-27).

Beyond the reference's four flags, the constants it hard-codes in the stage scripts are flags here too and are
forwarded to the stages (defaults = the reference's values, so `python main.py` is unchanged): the class list,
corpus folder and per-class cap of stage 1 (create_dataset.py:15,108-120), the reservoir shape and seed of
stage 2 (extract_lsm_features.py:10-16,30), the readout of stage 3.  `--nproc N` starts stages 1 and 2 (or the
in-memory route) as N ranks, one per GPU, through torch.distributed.run; the readout stays one process."""
import argparse
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))     # the stage scripts live next to this file;
                                                      # data files stay relative to the working directory


def _corpus_args(a):
    out = []
    if a.commands:
        out += ["--commands", a.commands]
    if a.commands_file:
        out += ["--commands-file", a.commands_file]
    if a.dataset_root:
        out += ["--dataset-root", a.dataset_root]
    if a.max_per_class is not None:
        out += ["--max-per-class", str(a.max_per_class)]
    if a.synthetic_per_class:
        out += ["--synthetic-per-class", str(a.synthetic_per_class)]
    return out


def _reservoir_args(a):
    out = []
    for flag, v in (("--num-neurons", a.num_neurons), ("--num-output-neurons", a.num_output_neurons),
                    ("--small-world-k", a.small_world_k), ("--seed", a.seed)):
        if v is not None:
            out += [flag, str(v)]
    return out


STAGES = (
    ("Step 1: Creating Spike Train Dataset",
     lambda a: ["create_dataset.py", "--n-filters", str(a.n_filters), "--filterbank", a.filterbank]
     + _corpus_args(a) + (["--packed"] if a.packed else [])),
    ("Step 2: Extracting LSM Features",
     lambda a: ["extract_lsm_features.py", "--feature-set", a.feature_set, "--multiplier", str(a.multiplier)]
     + _reservoir_args(a)),
    ("Step 3: Training and Evaluating Classifier",
     lambda a: ["train_classifier.py"] + (["--readout", a.readout] if a.readout else [])
     + (["--commands", a.commands] if a.commands else [])
     + (["--commands-file", a.commands_file] if a.commands_file else [])),
)

IN_MEMORY = """import argparse, sys, create_dataset as cd, extract_lsm_features as ex
a = argparse.Namespace(**{ns!r})
audio, labels = cd.collect_audio(commands=cd.commands_from_args(a), dataset_root=a.dataset_root,
                                 max_per_class=cd.MAX_SAMPLES_PER_CLASS if a.max_per_class is None else a.max_per_class,
                                 synthetic_per_class=a.synthetic_per_class)
ex.main_from_audio(audio, labels, a.n_filters, a.filterbank, a.feature_set, a.multiplier,
                   num_neurons=a.num_neurons, num_output_neurons=a.num_output_neurons,
                   small_world_k=a.small_world_k, seed=a.seed, readout=a.device_readout,
                   class_names=cd.commands_from_args(a))
"""


def _launch(script_args, nproc: int):
    """One process, or `nproc` ranks of it (one per GPU) through torch.distributed.run."""
    if nproc <= 1:
        return [sys.executable] + script_args
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # inherited by the ranks: dmabuf IPC, which RCCL needs here
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
            "--master-addr", "127.0.0.1", "--master-port", os.environ.get("LSM_MASTER_PORT", "29517")] + script_args


def run_pipeline(n_filters: int, filterbank: str, feature_set: str, multiplier: float, in_memory: bool = False,
                 **extra):
    """1. spike-train dataset, 2. LSM features, 3. readout.  A failing stage does not stop the
    next one (the reference discards exit codes too); it shows up as that stage's own message.
    `in_memory` (not in the reference): stages 1 and 2 run as ONE process (or one per GPU) that keeps audio,
    rasters and features on the GPU (extract_lsm_features.main_from_audio; no File 1), then the readout as usual --
    except that a PyTorch readout (`readout="torch-ridge"` / `"torch-logistic"`) runs inside that process on the
    feature rows still on the GPU and prints the final report itself (no stage 3 process, no File 2 reload).
    `extra`: the forwarded constants (commands, commands_file, dataset_root, max_per_class, synthetic_per_class,
    packed, num_neurons, num_output_neurons, small_world_k, seed, readout, nproc)."""
    args = argparse.Namespace(n_filters=n_filters, filterbank=filterbank, feature_set=feature_set,
                              multiplier=multiplier, commands=None, commands_file=None, dataset_root=None,
                              max_per_class=None, synthetic_per_class=int(os.environ.get("LSM_SYNTHETIC_PER_CLASS", "0")),
                              packed=False, num_neurons=None, num_output_neurons=None, small_world_k=None, seed=None,
                              readout=None, nproc=1)
    unknown = set(extra) - set(vars(args))
    if unknown:
        raise TypeError(f"run_pipeline: unknown arguments {sorted(unknown)}")
    vars(args).update(extra)
    print("--- Running Pipeline ---")
    if in_memory:
        print("\n--- Steps 1+2: audio -> LSM features on the GPU (no dataset file) ---", flush=True)
        ns = {k: v for k, v in vars(args).items() if k not in ("readout", "nproc", "packed")}
        # a PyTorch readout on the in-memory route runs where the features already are: scaler, readout and
        # predictions on the GPU, no File 2 reload; File 2 is still written
        device_readout = args.readout if args.readout in ("torch-ridge", "torch-logistic") else None
        ns["device_readout"] = device_readout
        code = f"import sys; sys.path.insert(0, {HERE!r})\n" + IN_MEMORY.format(ns=ns)
        if args.nproc > 1:            # torch.distributed.run needs a file to start
            import tempfile
            with tempfile.NamedTemporaryFile("w", suffix="_lsm_in_memory.py", delete=False) as fh:
                fh.write(code)
            try:
                subprocess.call(_launch([fh.name], args.nproc))
            finally:
                os.unlink(fh.name)
        else:
            subprocess.call([sys.executable, "-c", code])
        stages = () if device_readout else STAGES[2:]
    else:
        stages = STAGES
    for i, (title, command) in enumerate(stages):
        print(f"\n--- {title} ---", flush=True)
        cmd = command(args)
        sharded = not in_memory and i < 2
        subprocess.call(_launch([os.path.join(HERE, cmd[0])] + cmd[1:], args.nproc if sharded else 1))
    print("\n--- Pipeline Finished ---")


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Run the entire speech recognition pipeline.")
    ap.add_argument("--n-filters", type=int, default=128, help="Number of filters for the filterbank.")
    ap.add_argument("--filterbank", type=str, default="gammatone", choices=["mel", "gammatone"],
                    help="Type of filterbank to use.")
    ap.add_argument("--feature-set", type=str, default="original",
                    choices=['all', 'rate', 'timing', 'rhythm', 'original'],
                    help="The set of features to extract.")
    ap.add_argument("--multiplier", type=float, default=0.6, help="Multiplier for w_critico.")
    ap.add_argument("--in-memory", action="store_true", default=os.environ.get("LSM_IN_MEMORY", "0") == "1",
                    help="Run stages 1 and 2 in one process on the GPU without writing the dataset file.")
    # --- the reference's hard-coded constants, forwarded to the stages (defaults: the reference's values) ---
    ap.add_argument("--commands", type=str, default=None, help="Comma-separated class list (stage 1, report names).")
    ap.add_argument("--commands-file", type=str, default=None, help="File with one class name per line.")
    ap.add_argument("--dataset-root", type=str, default=None, help="Corpus folder (stage 1).")
    ap.add_argument("--max-per-class", type=int, default=None, help="Per-class cap on the sorted file list (stage 1).")
    ap.add_argument("--synthetic-per-class", type=int, default=int(os.environ.get("LSM_SYNTHETIC_PER_CLASS", "0")),
                    help="Synthetic clips per class instead of wav files (stage 1).")
    ap.add_argument("--packed", action="store_true", help="Bit-packed File 1 (stage 1).")
    ap.add_argument("--num-neurons", type=int, default=None, help="Reservoir size (stage 2; default 1000).")
    ap.add_argument("--num-output-neurons", type=int, default=None, help="Read-out neurons (stage 2; default 400).")
    ap.add_argument("--small-world-k", type=int, default=None, help="Ring neighbours (stage 2; default int(0.2 N)).")
    ap.add_argument("--seed", type=int, default=None, help="Reservoir wiring seed (stage 2; default 42).")
    ap.add_argument("--readout", type=str, default=None, choices=["sklearn", "torch-ridge", "torch-logistic"],
                    help="Readout of stage 3 (default: scikit-learn logistic regression, or LSM_READOUT).")
    ap.add_argument("--nproc", type=int, default=int(os.environ.get("LSM_NPROC", "1")),
                    help="Ranks (one per GPU) for stages 1 and 2.")
    a = ap.parse_args()
    run_pipeline(n_filters=a.n_filters, filterbank=a.filterbank, feature_set=a.feature_set,
                 multiplier=a.multiplier, in_memory=a.in_memory, commands=a.commands, commands_file=a.commands_file,
                 dataset_root=a.dataset_root, max_per_class=a.max_per_class,
                 synthetic_per_class=a.synthetic_per_class, packed=a.packed, num_neurons=a.num_neurons,
                 num_output_neurons=a.num_output_neurons, small_world_k=a.small_world_k, seed=a.seed,
                 readout=a.readout, nproc=a.nproc)
