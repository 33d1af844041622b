#!/bin/bash
# Round 5, call 43: the LIF kernels' raster-packing prologue with eight loads in flight per thread (pack_raster_bits) instead of one load and its wait per turn:
# the whole GPU suite, then cfg2's reservoir stage and whole path against the previous library (lib_prev.so), alternating.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05_call43; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'))
"; }
for rep in 1 2; do
for V in product prev; do
  if [ $V = product ]; then export LSM_HIP_LIB=; else export LSM_HIP_LIB=$GRAFT_REPO_ROOT/exp/variants/lib_$V.so; fi
  timeout -k 10 300 python3 bench.py --config cfg2 --stage reservoir --streams 1 --steps 60 --warmup 5 --no-cpu-baseline --no-unprimed 2>$O/err_$V.txt | line "cfg2 reservoir $V" | tee -a $O/pack.txt
done
done
for rep in 1 2 3; do
for V in product prev; do
  if [ $V = product ]; then export LSM_HIP_LIB=; else export LSM_HIP_LIB=$GRAFT_REPO_ROOT/exp/variants/lib_$V.so; fi
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-unprimed 2>$O/err_d_$V.txt | line "cfg2 driver command $V" | tee -a $O/pack.txt
done
done
for V in product prev; do
  if [ $V = product ]; then export LSM_HIP_LIB=; else export LSM_HIP_LIB=$GRAFT_REPO_ROOT/exp/variants/lib_$V.so; fi
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-unprimed 2>$O/err_l_$V.txt | line "cfg2 200 steps $V" | tee -a $O/pack.txt
done
