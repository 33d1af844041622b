#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for CFG in "cfg5 512 8" "cfg4 1024 4" "cfg4 1024 8"; do
  set -- $CFG
  LSM_KERNEL=ring timeout -k 10 300 python exp/big_cfg.py $1 $2 0 $3 2>&1 | grep -E "^wpc|rror" | sed "s/^/new-input-drive $1: /" | tee -a gpurun_out/r02_second.log
done
timeout -k 10 1000 python -m pytest tests/test_gpu_configs.py tests/test_gpu_graph.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r02_pytest2.log 2>&1
rc=$?; tail -25 gpurun_out/r02_pytest2.log; exit $rc
