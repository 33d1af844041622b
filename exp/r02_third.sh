#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for CFG in "cfg5 512 8" "cfg5 512 16" "cfg4 1024 8" "cfg4 1024 16"; do
  set -- $CFG
  LSM_KERNEL=ring timeout -k 10 300 python exp/big_cfg.py $1 $2 2 $3 2>&1 | grep -E "^wpc|rror|bit-exact" | sed "s/^/strided $1: /" | tee -a gpurun_out/r02_third.log
done
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_graph.py tests/test_gpu_configs.py -m gpu -q -x > gpurun_out/r02_pytest3.log 2>&1
rc=$?; tail -25 gpurun_out/r02_pytest3.log | cut -c1-250; exit $rc
