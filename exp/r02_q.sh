#!/bin/bash
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
for CFG in "cfg5 512 8 ring" "cfg5 512 16 ring" "cfg4 1024 8 ring" "cfg4 1024 4 ring"; do
  set -- $CFG
  LSM_KERNEL=$4 timeout -k 10 300 python exp/big_cfg.py $1 $2 1 $3 2>&1 | grep -E "^wpc|rror|bit-exact" | sed "s/^/branch $1 $4: /" | cut -c1-120 | tee -a gpurun_out/r02_q.log
done
