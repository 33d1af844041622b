#!/bin/bash
# Round 3: HotPath-owned raster/scratch buffers (LSM_HOTPATH_POOL=1, the default) against per-step allocations + record_stream (=0)
OUT=gpurun_out/r03_pool.txt
run() {
  local label=$1; shift
  env "$@" 2>/dev/null | python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$label FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$label', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lif in-region', r.get('kernel_ms'), 'host enqueue', d['config']['host_enqueue_ms_per_step'])
" | tee -a $OUT
}
for rep in 1 2 3 4; do
  for W in 0 1; do
    run "pool $W --steps 20 --warmup 5" LSM_HOTPATH_POOL=$W python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
  done
done
for rep in 1 2; do
  for W in 0 1; do
    run "pool $W --steps 200 --warmup 12" LSM_HOTPATH_POOL=$W python3 bench.py --steps 200 --warmup 12 --no-cpu-baseline
  done
done
