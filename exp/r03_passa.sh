#!/bin/bash
# fused front end, pass A of the epilogue unrolled by two (u2: 32 scratch loads in flight per iteration instead of 16) against the build (u1), same box
OUT=gpurun_out/r03_passa.txt
U2=/root/repo/lsm-speech-classifier_amd/liblsm_hip_u2.so
for rep in 1 2 3; do
  for V in u1 u2; do
    L="LSM_X=0"; [ $V = u2 ] && L="LSM_HIP_LIB=$U2"
    for ARGS in "--stage frontend --streams 1 --steps 40 --warmup 5" "--stage frontend --steps 200 --warmup 12" "--steps 200 --warmup 12"; do
      env $L python3 bench.py $ARGS --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$V $ARGS ->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step')" | tee -a $OUT
    done
  done
done
