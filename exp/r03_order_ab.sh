#!/bin/bash
# Round 3: reservoir launches with their clips started longest first (lsm_reservoir_run_ordered, the default of SNN.run_batch for
# batches of more clips than compute units) against the plain launch order (LSM_RESERVOIR_ORDER=0), same box, alternating.
OUT=gpurun_out/r03_order_ab.txt
run() {
  local label=$1; shift
  env "$@" 2>/dev/null | python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$label FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$label', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; kernel', d['config'].get('reservoir_kernel'), 'lone', r.get('lone_launch_kernel_ms'), 'frac', r.get('frac'))
" | tee -a $OUT
}
for rep in 1 2; do
  for V in plain ordered; do
    L="LSM_RESERVOIR_ORDER=1"; [ $V = plain ] && L="LSM_RESERVOIR_ORDER=0"
    run "$V cfg4 B1024 reservoir" $L python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline
    run "$V cfg5 B512 reservoir" $L python3 bench.py --config cfg5 --batch 512 --stage reservoir --streams 1 --steps 6 --warmup 2 --no-cpu-baseline
    run "$V cfg2 B1024 reservoir" $L python3 bench.py --config cfg2 --batch 1024 --stage reservoir --streams 1 --steps 20 --warmup 3 --no-cpu-baseline
  done
done
for V in plain ordered; do
  L="LSM_RESERVOIR_ORDER=1"; [ $V = plain ] && L="LSM_RESERVOIR_ORDER=0"
  run "$V cfg5 B4096 reservoir" $L python3 bench.py --config cfg5 --stage reservoir --streams 1 --steps 3 --warmup 1 --no-cpu-baseline
  run "$V cfg4 whole path" $L python3 bench.py --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline
  run "$V cfg5 whole path" $L python3 bench.py --config cfg5 --steps 4 --warmup 1 --no-cpu-baseline
done
