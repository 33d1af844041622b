#!/bin/bash
# Round 5, call 7: which port bounds the PAIR kernel's row loop?  Same source with 8 extra scalar / 8 extra vector instructions,
# 2 extra LDS stores, 2 extra zero-byte buffer loads per row and wave, and with per-lane dump words; alternating on one box.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call7; mkdir -p $O
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'))
"; }
for rep in 1 2; do
  for V in product lsm_pair_dummy_salu_8 lsm_pair_dummy_valu_8 lsm_pair_dummy_lds_2 lsm_pair_dummy_vmem_2 lsm_pair_own_dump_1; do
    if [ $V = product ]; then L=""; else L=exp/variants/lib_$V.so; fi
    LSM_HIP_LIB=$L timeout -k 10 300 python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>$O/err_$V.txt | line "cfg4 reservoir $V" >> $O/ports.txt
  done
done
cat $O/ports.txt
