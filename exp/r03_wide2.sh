#!/bin/bash
OUT=gpurun_out/r03_wide2.txt
run() {
  local label=$1; shift
  env "$@" 2>/dev/null | python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$label FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$label', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lif in-region', r.get('kernel_ms'))
" | tee -a $OUT
}
for rep in 1 2 3; do
  for W in 1 2 3; do
    run "wide_below $W --steps 20 --warmup 5" LSM_FE_WIDE_BELOW=$W python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
  done
done
for W in 1 2 3; do
  run "wide_below $W --steps 200 --warmup 12" LSM_FE_WIDE_BELOW=$W python3 bench.py --steps 200 --warmup 12 --no-cpu-baseline
done
