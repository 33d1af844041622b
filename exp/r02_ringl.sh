#!/bin/bash
# ring rows with LDS accumulators (liblsm_hip.so) against the register/scratch version (liblsm_hip_base.so)
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_c_abi.py -m gpu -q -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for V in base new base new; do
  LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip_$V.so
  [ $V = new ] && LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip.so
  for CFG in "cfg5 512 8 ring" "cfg4 1024 8 ring" "cfg4 1024 4 ring" "cfg5 512 16 ring"; do
    set -- $CFG
    LSM_HIP_LIB=$LIB LSM_KERNEL=$4 timeout -k 10 300 python exp/big_cfg.py $1 $2 0 $3 2>&1 | grep -E "^wpc|rror" | sed "s/^/$V $1 $4: /" | cut -c1-140
  done
done
