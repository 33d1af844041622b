#!/bin/bash
# Round 3: at most K front-end launches hold CUs at a time (the K-th older one must have finished: an event wait on the front-end stream),
# so that a fifth queued launch does not split the chip five ways with four running ones.  LSM_FE_MAX_IN_FLIGHT=K, 0 = off.  Same box.
OUT=gpurun_out/r03_fe_cap.txt
for rep in 1 2; do
  for V in "0 5" "4 5" "4 6" "3 5" "5 6"; do
    set -- $V
    for ARGS in "--stage frontend --steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
      env LSM_FE_MAX_IN_FLIGHT=$1 python3 bench.py $ARGS --fe-streams $2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('cap $1 fe_streams $2 $ARGS ->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step')" | tee -a $OUT
    done
  done
done
