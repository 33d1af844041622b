#!/bin/bash
# LDS-accumulator ring kernel: ablation builds (1 = no window loads, 2 = no accumulator read-modify-write, 8 = no list
# loads, 11 = none of them; wrong results, time only)
for V in ${VARIANTS:-new abl1 abl8 abl2 abl11}; do
  LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip_$V.so
  [ $V = new ] && LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip.so
  for CFG in "cfg4 1024 8 ring" "cfg5 512 8 ring"; do
    set -- $CFG
    LSM_HIP_LIB=$LIB LSM_KERNEL=$4 timeout -k 10 300 python exp/big_cfg.py $1 $2 0 $3 2>&1 | grep -E "^wpc|rror" | sed "s/^/$V $1 $4: /" | cut -c1-160
  done
done
