#!/bin/bash
# Round 5, call 1: which issue port bounds the ring kernel's row loop at cfg4?  Same source with 8 extra scalar / 8 extra vector /
# 2 extra LDS-store instructions per row and wave (results unchanged), alternating with the product build on one box.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call1; mkdir -p $O
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'))
"; }
for rep in 1 2; do
  for V in product lsm_ring_dummy_salu_8 lsm_ring_dummy_valu_8 lsm_ring_dummy_lds_2; do
    if [ $V = product ]; then L=""; else L=exp/variants/lib_$V.so; fi
    LSM_HIP_LIB=$L timeout -k 10 300 python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>$O/err_$V.txt | line "cfg4 reservoir $V" >> $O/ports.txt
  done
done
cat $O/ports.txt
