#!/bin/bash
# Round 3: more front-end streams than four need more hardware queues than 12 (front-end + reservoir + default streams)
OUT=gpurun_out/r03_queues.txt
for Q in 12 16 24; do
  for TOPO in "5 6" "6 6" "8 6" "8 4" "6 4"; do
    set -- $TOPO
    for ST in frontend full; do
      for A in "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
        GPU_MAX_HW_QUEUES=$Q python3 bench.py --stage $ST --fe-streams $1 --streams $2 $A --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('queues $Q fe_streams $1 streams $2 stage $ST $A ->', d['ms_per_step'], 'ms/step =', round(d['ms_per_step'] * d['steps'], 2), 'ms; hw_queues', d['config']['hw_queues'])" | tee -a $OUT
      done
    done
  done
done
