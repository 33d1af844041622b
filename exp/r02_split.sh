#!/bin/bash
for A in "--steps 20 --warmup 5 --batch 256" "--steps 40 --warmup 10 --batch 128" "--steps 80 --warmup 20 --batch 64" "--steps 40 --warmup 10 --batch 128 --streams 8" "--steps 40 --warmup 10 --batch 128 --streams 12" "--steps 20 --warmup 5 --batch 256" "--steps 10 --warmup 3 --batch 512" ; do
  python3 bench.py $A --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$A', '->', d['value'], 'clips/s; total ms', round(d['ms_per_step']*d['steps'],3))
"
done
