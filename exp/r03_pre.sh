#!/bin/bash
# Round 3: fused front end with the per-sample pair {x, b0*x} precomputed by a lane-parallel prologue and fetched with one broadcast
# 16-byte load per sample (68 instead of 70 ALU instructions per sample and lane; new) against the same source built with
# -DLSM_GTF_PRE=0 (prev), same box, alternating: front ends alone and whole path.
OUT=gpurun_out/r03_pre.txt
PREV=/root/repo/lsm-speech-classifier_amd/liblsm_hip_prev.so
for rep in 1 2 3; do
  for V in prev new; do
    L="LSM_X=0"; [ $V = prev ] && L="LSM_HIP_LIB=$PREV"
    for ARGS in "--stage frontend --steps 200 --warmup 12" "--steps 200 --warmup 12" "--steps 20 --warmup 5"; do
      env $L python3 bench.py $ARGS --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$V $ARGS ->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step')" | tee -a $OUT
    done
  done
done
