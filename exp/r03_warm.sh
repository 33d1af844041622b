#!/bin/bash
# Round 3: is the 20-step burst slow because the clocks are still ramping?  Same command with more untimed warm-up steps.
OUT=gpurun_out/r03_warm.txt
for W in 5 20 60 200; do
  for rep in 1 2 3; do
    python3 bench.py --steps 20 --warmup $W --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('warmup $W steps 20 ->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step')" | tee -a $OUT
  done
done
for W in 5 60; do
  python3 bench.py --stage frontend --steps 20 --warmup $W --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('front end alone, warmup $W steps 20 ->', d['ms_per_step'], 'ms/step =', round(d['ms_per_step'] * 20, 2), 'ms')" | tee -a $OUT
done
