#!/bin/bash
# Does the longest-first order pay inside the cfg2 pipeline too (256 clips per launch = one per CU, but six launches and four front ends
# share the CUs)?  LSM_ORDER_ALWAYS=1 ranks every batch; same box, alternating.
OUT=gpurun_out/r03_order_cfg2.txt
for rep in 1 2 3; do
  for V in plain always; do
    L="LSM_ORDER_ALWAYS=0"; [ $V = always ] && L="LSM_ORDER_ALWAYS=1"
    for ARGS in "--steps 200 --warmup 12" "--steps 20 --warmup 5"; do
      env $L python3 bench.py $ARGS --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$V $ARGS ->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lif in-region', d['roofline']['kernel_ms'], 'host enqueue', d['config']['host_enqueue_ms_per_step'])" | tee -a $OUT
    done
  done
done
