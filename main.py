"""Run the three pipeline stages in order, each as its own Python process (like the reference's
main.py, which shells out and ignores exit codes: /root/reference/main.py:19-27)."""
import argparse
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))     # the stage scripts live next to this file;
                                                      # data files stay relative to the working directory

STAGES = (
    ("Step 1: Creating Spike Train Dataset",
     lambda a: ["create_dataset.py", "--n-filters", str(a.n_filters), "--filterbank", a.filterbank]),
    ("Step 2: Extracting LSM Features",
     lambda a: ["extract_lsm_features.py", "--feature-set", a.feature_set, "--multiplier", str(a.multiplier)]),
    ("Step 3: Training and Evaluating Classifier", lambda a: ["train_classifier.py"]),
)


IN_MEMORY = """import create_dataset as cd, extract_lsm_features as ex, os
audio, labels = cd.collect_audio(synthetic_per_class=int(os.environ.get("LSM_SYNTHETIC_PER_CLASS", "0")))
ex.main_from_audio(audio, labels, {n_filters}, {filterbank!r}, {feature_set!r}, {multiplier})"""


def run_pipeline(n_filters: int, filterbank: str, feature_set: str, multiplier: float, in_memory: bool = False):
    """1. spike-train dataset, 2. LSM features, 3. readout.  A failing stage does not stop the
    next one (the reference discards exit codes too); it shows up as that stage's own message.
    `in_memory` (not in the reference): stages 1 and 2 run as ONE process that keeps audio, rasters and
    features on the GPU (extract_lsm_features.main_from_audio; no File 1), then the readout as usual."""
    args = argparse.Namespace(n_filters=n_filters, filterbank=filterbank, feature_set=feature_set,
                              multiplier=multiplier)
    print("--- Running Pipeline ---")
    if in_memory:
        print("\n--- Steps 1+2: audio -> LSM features on the GPU (no dataset file) ---", flush=True)
        code = IN_MEMORY.format(n_filters=n_filters, filterbank=filterbank, feature_set=feature_set,
                                multiplier=multiplier)
        subprocess.call([sys.executable, "-c", f"import sys; sys.path.insert(0, {HERE!r}); " + code.replace("\n", "; ")])
        title, command = STAGES[2]
        print(f"\n--- {title} ---", flush=True)
        subprocess.call([sys.executable, os.path.join(HERE, command(args)[0])])
        print("\n--- Pipeline Finished ---")
        return
    for title, command in STAGES:
        print(f"\n--- {title} ---", flush=True)
        cmd = command(args)
        subprocess.call([sys.executable, os.path.join(HERE, cmd[0])] + cmd[1:])
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
    a = ap.parse_args()
    run_pipeline(n_filters=a.n_filters, filterbank=a.filterbank, feature_set=a.feature_set,
                 multiplier=a.multiplier, in_memory=a.in_memory)
