#!/bin/bash
# Round 4, call 17: what the distributed code path costs ONE rank on the driver's command: the rehearsal switch LSM_BENCH_FORCE_DIST=1
# (RCCL process group of one rank, broadcast, exchange, barrier in the fences, 16 hardware queues) against the plain N = 1 run, three exchange modes.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call17; mkdir -p $O
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); x = d.get('exchange') or {}
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; exchange', {k: x.get(k) for k in ('mode', 'exchange_ms', 'exchange_bytes', 'collectives')}, 'hw queues', d.get('config', {}).get('hw_queues'))
"; }
for rep in 1 2 3; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-unprimed 2>/dev/null | line "plain N=1 (12 queues)" >> $O/x.txt
  GPU_MAX_HW_QUEUES=16 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-unprimed 2>/dev/null | line "plain N=1, 16 queues" >> $O/x.txt
  for M in chunked once per-step; do
    RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 LSM_BENCH_FORCE_DIST=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --exchange $M 2>$O/err_$M.txt | line "one RCCL rank, exchange $M" >> $O/x.txt
  done
done
cat $O/x.txt
