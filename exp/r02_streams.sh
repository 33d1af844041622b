#!/bin/bash
# the driver's command (20 timed steps from an idle GPU) against the depth of the rotation and the hardware-queue count
for Q in 8 12 16; do
  for S in 4 5 6 7 8; do
    for R in 1 2 3; do
      GPU_MAX_HW_QUEUES=$Q python3 bench.py --steps 20 --warmup 5 --streams $S --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('hwq $Q streams $S run $R ->', d['value'], d['ms_per_step'])
"
    done
  done
done | tee gpurun_out/r02_streams.log
