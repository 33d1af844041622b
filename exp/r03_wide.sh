#!/bin/bash
# Round 3: a front end that meets an idle GPU goes out in the low-latency (one-chain) layout (LSM_FE_WIDE_WHEN_IDLE=1, the default)
# against always the throughput layout (=0); the driver's 20-step command and 200 steps, same box, alternating.
OUT=gpurun_out/r03_wide.txt
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
for rep in 1 2 3 4; do
  for W in 0 1; do
    run "wide_when_idle $W --steps 20 --warmup 5" LSM_FE_WIDE_WHEN_IDLE=$W python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
  done
done
for rep in 1 2; do
  for W in 0 1; do
    run "wide_when_idle $W --steps 200 --warmup 12" LSM_FE_WIDE_WHEN_IDLE=$W python3 bench.py --steps 200 --warmup 12 --no-cpu-baseline
  done
done
run "split rotation --steps 20 --warmup 5" LSM_FRONTEND_SPLIT=1 python3 bench.py --steps 20 --warmup 5 --fe-streams 0 --no-cpu-baseline
run "split rotation --steps 20 --warmup 5" LSM_FRONTEND_SPLIT=1 python3 bench.py --steps 20 --warmup 5 --fe-streams 0 --no-cpu-baseline
