#!/bin/bash
# Round 3: fused front end with the first-section product b0*x shared between the two chains of a lane (70 instead of 71 instructions per
# sample and lane; new) against the previous commit (prev), same box, alternating: front ends alone and whole path.
OUT=gpurun_out/r03_shb0.txt
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
