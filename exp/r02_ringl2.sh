#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
for V in ${VARIANTS:-nopf new nopf new}; do
  LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip_$V.so
  [ $V = new ] && LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip.so
  for CFG in "cfg4 1024 8 ring" "cfg5 512 8 ring"; do
    set -- $CFG
    LSM_HIP_LIB=$LIB LSM_KERNEL=$4 timeout -k 10 300 python exp/big_cfg.py $1 $2 0 $3 2>&1 | grep -E "^wpc|rror" | sed "s/^/$V $1 $4: /" | cut -c1-160
  done
done
