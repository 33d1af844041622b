#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r02_pytest_full.log 2>&1
rc=$?; tail -25 gpurun_out/r02_pytest_full.log | cut -c1-250
for CFG in "cfg5 512 0" "cfg4 1024 0"; do
  set -- $CFG
  LSM_KERNEL=auto timeout -k 10 300 python exp/big_cfg.py $1 $2 0 $3 2>&1 | grep -E "^wpc|rror|layout" | sed "s/^/auto $1: /" | tee -a gpurun_out/r02_full.log
done
exit $rc
