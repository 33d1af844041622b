#!/bin/bash
# ring-row kernel: where does a step's time go?  (ablation builds: 1/2/8 compute wrong results, only time is read)
set -o pipefail
for V in base scalar a1 a2 a4 a8 a3 a11; do
  LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip_$V.so
  [ $V = base ] && LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip.so
  for CFG in "cfg5 512 8" "cfg4 1024 4"; do
    set -- $CFG
    LSM_HIP_LIB=$LIB LSM_KERNEL=ring timeout -k 10 300 python exp/big_cfg.py $1 $2 0 $3 2>&1 | grep -E "^wpc|rror" | sed "s/^/$V $1: /" | tee -a gpurun_out/r02_ablate.log
  done
done
