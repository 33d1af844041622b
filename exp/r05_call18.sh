#!/bin/bash
# Round 5, call 18: the whole GPU suite on the round's final code, then the driver's bench command and cfg4.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call18; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/summary.txt
tail -5 $O/pytest.log | tee -a $O/summary.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err && cat $O/bench_driver.json
timeout -k 10 200 python3 bench.py --config cfg4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err && cat $O/bench_cfg4.json
