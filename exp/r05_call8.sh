#!/bin/bash
# Round 5, call 8: (1) residency ablation of the pair kernel (exp/r05_residency.py: replay of recorded spikes; full build, lean
# build at 2 / 3 / 4 clips per CU); (2) the zero-byte buffer loads again, this time inside the row pipeline.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call8; mkdir -p $O
timeout -k 10 400 python3 exp/r05_residency.py record cfg4 1024 2>$O/err_record.txt | tee -a $O/residency.txt
for rep in 1 2; do
for V in replay replay_lean2 replay_lean3 replay_lean4; do
  LSM_HIP_LIB=exp/variants/lib_$V.so timeout -k 10 300 python3 exp/r05_residency.py replay cfg4 1024 2>$O/err_$V.txt | tee -a $O/residency.txt
done
done
rm -f /tmp/r05_sm.npy
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'))
"; }
for V in product lsm_pair_dummy_vmem_2; do
  if [ $V = product ]; then L=""; else L=exp/variants/lib_$V.so; fi
  LSM_HIP_LIB=$L timeout -k 10 300 python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>$O/err_$V.txt | line "cfg4 reservoir $V" | tee -a $O/ports.txt
done
