#!/bin/bash
# Round 3: where do the driver's 20 steps lose?  The burst with the front end alone, the reservoir alone and both, against 200 steps.
OUT=gpurun_out/r03_burst.txt
for ST in frontend reservoir full; do
  for A in "--steps 20 --warmup 5" "--steps 40 --warmup 5" "--steps 200 --warmup 12"; do
    for rep in 1 2; do
    python3 bench.py --stage $ST $A --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('stage $ST $A ->', d['ms_per_step'], 'ms/step =', round(d['ms_per_step'] * d['steps'], 3), 'ms in all')" | tee -a $OUT
    done
  done
done
