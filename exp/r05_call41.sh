#!/bin/bash
# Round 5, call 41: the mel tests with the sweep over clip lengths (odd, shorter than the window, odd hops).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_call41
timeout -k 10 600 python3 -m pytest tests/test_gpu_mel.py -q > gpurun_out/r05_call41/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r05_call41/pytest.log
