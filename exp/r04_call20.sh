#!/bin/bash
# Round 4, call 20: with the steady state at the power limit, does a reservoir format that moves fewer bytes (ring rows, CSC lists) help the WHOLE path at cfg2?
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call20; mkdir -p $O
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'), 'in-region', r.get('in_region_kernel_ms'))
"; }
for rep in 1 2; do for K in auto ring sparse; do
  python3 bench.py --kernel $K --steps 20 --warmup 5 --no-cpu-baseline --no-unprimed 2>$O/err_$K.txt | line "kernel $K driver" >> $O/x.txt
  python3 bench.py --kernel $K --steps 2000 --warmup 12 --no-cpu-baseline --no-unprimed 2>>$O/err_$K.txt | line "kernel $K 2000 steps" >> $O/x.txt
done; done
cat $O/x.txt
