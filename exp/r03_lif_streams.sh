#!/bin/bash
# Round 3: number of reservoir streams behind 5 front-end streams: throughput and the in-region duration of a reservoir launch
OUT=gpurun_out/r03_lif_streams.txt
run() {
  local label=$1; shift
  env "$@" 2>/dev/null | python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$label FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$label', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lif in-region', r.get('kernel_ms'), 'frac', r.get('frac'))
" | tee -a $OUT
}
for rep in 1 2; do
  for L in 2 3 4 6; do
    for A in "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
      run "fe_streams 5 streams $L $A" python3 bench.py $A --fe-streams 5 --streams $L --no-cpu-baseline
    done
  done
done
