#!/bin/bash
# Round 5, call 22: the step's input counts and leak terms computed while its first rows (product, PRE=2) or its row records
# (pre1) are in the air, against computing them in the update (pre0).  Parity first.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call22; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_ordered.py tests/test_gpu_fuzz.py -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/summary.txt
tail -3 $O/pytest.log | tee -a $O/summary.txt
[ $rc -eq 0 ] || exit $rc
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'))
"; }
for rep in 1 2; do
for V in product pre1 pre0; do
  if [ $V = product ]; then L=""; else L=exp/variants/lib_$V.so; fi
  LSM_HIP_LIB=$L timeout -k 10 300 python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>$O/err_$V.txt | line "cfg4 reservoir $V" | tee -a $O/pre.txt
done
done
for rep in 1 2; do
for V in product pre1 pre0; do
  if [ $V = product ]; then L=""; else L=exp/variants/lib_$V.so; fi
  LSM_HIP_LIB=$L timeout -k 10 300 python3 bench.py --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed 2>$O/err_full_$V.txt | line "cfg4 whole path $V" | tee -a $O/pre.txt
done
done
