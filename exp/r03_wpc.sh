#!/bin/bash
# Round 3: reservoir layout inside the two-stage pipeline (waves per clip: the library's choice for a shared GPU is 4)
OUT=gpurun_out/r03_wpc.txt
for rep in 1 2; do
for W in ${WS:--1 8 16}; do
  for A in "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
    python3 bench.py $A --waves-per-clip $W --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('waves_per_clip $W ->', d['config']['waves_per_clip'], '$A', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lif in-region', r['kernel_ms'], 'frac', r['frac'])" | tee -a $OUT
  done
done
done
