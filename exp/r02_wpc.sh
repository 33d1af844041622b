#!/bin/bash
# whole pipeline at cfg2 against the reservoir layout inside the rotation (-1 = the library's choice for a shared chip)
for R in 1 2; do
for A in "" "--waves-per-clip 4" "--waves-per-clip 8" "--waves-per-clip 16"; do
  for S in "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
  python3 bench.py $S $A --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = d.get('roofline', {})
print('[$A] $S ->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lif in-region', r.get('kernel_ms'), 'waves', d['config']['waves_per_clip'])
"
  done
done
done
