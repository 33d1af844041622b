#!/bin/bash
# Round 3: HotPath.prime(min_ms): untimed pipeline time before the W warm-up steps against the driver's 20-step figure
OUT=gpurun_out/r03_prime.txt
for rep in 1 2 3; do
  for P in 0 25 50 100 200; do
    python3 bench.py --steps 20 --warmup 5 --prime-ms $P --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('prime_ms $P --steps 20 --warmup 5 ->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step;', d['config']['prime'])" | tee -a $OUT
  done
done
for P in 0 50; do
  python3 bench.py --steps 200 --warmup 12 --prime-ms $P --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('prime_ms $P --steps 200 --warmup 12 ->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step')" | tee -a $OUT
done
