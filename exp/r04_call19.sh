#!/bin/bash
# Round 4, call 19: call 18 again after bench.py pays RCCL's first barrier before the warm-up steps.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call19; mkdir -p $O
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); x = d.get('exchange') or {}
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; fences', d.get('fences'), 'exchange_ms', x.get('exchange_ms'), 'host ms per collective', x.get('host_enqueue_ms_per_collective'), 'enqueue', d['config'].get('host_enqueue_ms_per_step'))
"; }
D="RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 LSM_BENCH_FORCE_DIST=1"
for rep in 1 2 3; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-unprimed 2>/dev/null | line "plain" >> $O/x.txt
  for M in once chunked per-step; do
    env $D python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --exchange $M 2>/dev/null | line "one RCCL rank, $M" >> $O/x.txt
  done
done
cat $O/x.txt
