#!/bin/bash
# Round 3: front-end stream count x low-latency first launch once the clock is at its working point (bench.py --prime-ms 40, the default)
OUT=gpurun_out/r03_burst3.txt
for rep in 1 2 3; do
  for F in 4 5; do
    for W in 0 1; do
      LSM_FE_WIDE_WHEN_IDLE=$W python3 bench.py --fe-streams $F --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('fe_streams $F wide_when_idle $W ->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step')" | tee -a $OUT
    done
  done
done
