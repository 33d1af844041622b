#!/bin/bash
# Round 5, call 26: does the pair-block kernel beat dense rows at cfg2 (N = 1000: 8 blocks, 4 waves x 2 blocks)?  Reservoir stage alone and whole path.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call26; mkdir -p $O
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'), d['config'].get('waves_per_clip'), d['config'].get('lds_bytes_per_clip'))
"; }
for rep in 1 2; do
for K in dense ring-pairs ring-quads; do
  timeout -k 10 300 python3 bench.py --config cfg2 --kernel $K --stage reservoir --streams 1 --steps 40 --warmup 5 --no-cpu-baseline --no-unprimed 2>$O/err_$K.txt | line "cfg2 reservoir $K" | tee -a $O/cfg2.txt
done
done
for K in dense ring-pairs; do
  timeout -k 10 300 python3 bench.py --config cfg2 --kernel $K --steps 200 --warmup 5 --no-cpu-baseline --no-unprimed 2>$O/err_full_$K.txt | line "cfg2 whole path 200 steps $K" | tee -a $O/cfg2.txt
done
