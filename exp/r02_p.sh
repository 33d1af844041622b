#!/bin/bash
for V in p6 p8 p12; do
  LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip_$V.so
  for CFG in "cfg5 512 8 ring" "cfg4 1024 8 ring"; do
    set -- $CFG
    LSM_HIP_LIB=$LIB LSM_KERNEL=$4 timeout -k 10 300 python exp/big_cfg.py $1 $2 0 $3 2>&1 | grep -E "^wpc|rror" | sed "s/^/$V $1 $4: /" | cut -c1-120 | tee -a gpurun_out/r02_p.log
  done
done
