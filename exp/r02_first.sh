#!/bin/bash
# round 2, first GPU pass: ring-row kernel parity, then dense vs ring timings at cfg4 / cfg5
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.log 2>&1 || { tail -30 gpurun_out/r02_smoke.log; exit 1; }
tail -2 gpurun_out/r02_smoke.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_c_abi.py tests/test_gpu_abi_errors.py -m gpu -q > gpurun_out/r02_pytest1.log 2>&1
rc=$?; tail -15 gpurun_out/r02_pytest1.log
for K in dense ring; do
  for W in 0 4 8 16; do
    LSM_KERNEL=$K timeout -k 10 300 python exp/big_cfg.py cfg4 1024 2 $W 2>&1 | grep -E "wpc|bit-exact|rror" | sed "s/^/cfg4 $K: /" | tee -a gpurun_out/r02_big.log
  done
done
for K in dense ring; do
  for W in 8 16; do
    LSM_KERNEL=$K timeout -k 10 400 python exp/big_cfg.py cfg5 512 2 $W 2>&1 | grep -E "wpc|bit-exact|rror" | sed "s/^/cfg5 $K: /" | tee -a gpurun_out/r02_big.log
  done
done
