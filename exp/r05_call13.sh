#!/bin/bash
# Round 5, call 13: the whole GPU suite on the round's code; mid-size reservoirs: pair blocks against quads and dense rows.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call13; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest.log | tee -a $O/summary.txt
timeout -k 10 300 python3 exp/r02_midsize.py > $O/midsize.txt 2>&1; cat $O/midsize.txt
