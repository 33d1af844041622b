#!/bin/bash
# Round 5, call 24: the round's evidence re-taken on the final kernels (exp/r05_profiles.sh), then the whole GPU suite.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
bash exp/r05_profiles.sh > gpurun_out/prof5.log 2>&1; rc=$?; tail -40 gpurun_out/prof5.log; echo "profiles rc=$rc"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_call24
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r05_call24/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_call24/pytest.log
