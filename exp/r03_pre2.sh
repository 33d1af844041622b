#!/bin/bash
# Where the {x, b0*x} variant loses: prev (no pairs), new (prologue + loads), skip (loads only, prologue compiled out: garbage results,
# timing only). Front ends alone, same box, alternating.
OUT=gpurun_out/r03_pre2.txt
D=/root/repo/lsm-speech-classifier_amd
for rep in 1 2 3; do
  for V in prev new skip; do
    L="LSM_X=0"; [ $V = prev ] && L="LSM_HIP_LIB=$D/liblsm_hip_prev.so"; [ $V = skip ] && L="LSM_HIP_LIB=$D/liblsm_hip_skip.so"
    env $L python3 bench.py --stage frontend --steps 200 --warmup 12 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$V frontend ->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step')" | tee -a $OUT
  done
done
