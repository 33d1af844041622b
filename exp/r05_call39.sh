#!/bin/bash
# Round 5, call 39 (last): the whole GPU suite on the final library, then the round's evidence re-taken (exp/r05_profiles.sh).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_call39
timeout -k 10 600 python3 -m pytest tests -m gpu -q > gpurun_out/r05_call39/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r05_call39/pytest.log
[ $rc -eq 0 ] || exit $rc
bash exp/r05_profiles.sh > gpurun_out/prof5.log 2>&1; rc=$?; tail -25 gpurun_out/prof5.log; echo "profiles rc=$rc"
