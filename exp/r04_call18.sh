#!/bin/bash
# Round 4, call 18: where the distributed code path loses 6-10 % of ONE rank's driver-command rate (call 17): host timings of the fences, and
# the plain run with an idle GPU of 0.2 / 0.5 / 1 / 3 ms right before the timed region (what a barrier leaves behind).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call18; mkdir -p $O
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); x = d.get('exchange') or {}
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; fences', d.get('fences'), 'exchange_ms', x.get('exchange_ms'), 'enqueue', d['config'].get('host_enqueue_ms_per_step'))
"; }
D="RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 LSM_BENCH_FORCE_DIST=1"
for rep in 1 2 3; do if false; then
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-unprimed 2>/dev/null | line "plain" >> $O/x.txt
  for I in 0.2 0.5 1 3; do
    LSM_BENCH_DIAG_IDLE_MS=$I python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-unprimed 2>/dev/null | line "plain, idle $I ms before the region" >> $O/x.txt
  done
  fi; env $D python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --exchange once 2>/dev/null | line "one RCCL rank, once" >> $O/x.txt
  env $D python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --exchange chunked 2>/dev/null | line "one RCCL rank, chunked" >> $O/x.txt
  env $D LSM_BENCH_BACKEND=gloo python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --exchange once 2>/dev/null | line "one gloo rank, once" >> $O/x.txt
done
cat $O/x.txt
