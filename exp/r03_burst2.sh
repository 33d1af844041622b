#!/bin/bash
# Round 3: the 20-step burst, front end alone and whole path: 4 or 5 front-end streams x first launch wide or not
OUT=gpurun_out/r03_burst2.txt
for ST in frontend full; do
  for F in 4 5; do
    for W in 0 1; do
      for rep in 1 2; do
      LSM_FE_WIDE_WHEN_IDLE=$W python3 bench.py --stage $ST --fe-streams $F --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('stage $ST fe_streams $F wide_when_idle $W ->', d['ms_per_step'], 'ms/step =', round(d['ms_per_step'] * d['steps'], 3), 'ms in all')" | tee -a $OUT
      done
    done
  done
done
