#!/bin/bash
# Round 4, call 13: wave priority of the ring kernel's step loop inside the pipeline (cfg4, cfg5 whole path) and alone.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call13; mkdir -p $O
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'), 'in-region', r.get('in_region_kernel_ms'))
"; }
for rep in 1 2; do
  for P in 0 1 2 3; do
    L=exp/variants/liblsm_ring_prio$P.so; [ $P = 0 ] && L=
    LSM_HIP_LIB=$L python3 bench.py --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 whole path ring prio $P" >> $O/prio.txt
  done
done
for P in 0 1 2 3; do
  L=exp/variants/liblsm_ring_prio$P.so; [ $P = 0 ] && L=
  LSM_HIP_LIB=$L python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 reservoir alone ring prio $P" >> $O/prio.txt
  LSM_HIP_LIB=$L python3 bench.py --config cfg5 --steps 4 --warmup 1 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg5 whole path ring prio $P" >> $O/prio.txt
done
cat $O/prio.txt
