#!/bin/bash
# Round 5, call 9: does a buffer load's cost on the texture-address path follow the lanes that take part?  Two zero-byte loads per row
# and wave inside the row pipeline, with all 64 / 32 / 12 lanes in the execution mask; alternating with the product build.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call9; mkdir -p $O
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'))
"; }
for rep in 1 2; do
for V in product lsm_pair_dummy_vmem_2 vmem2_lanes32 vmem2_lanes12; do
  if [ $V = product ]; then L=""; else L=exp/variants/lib_$V.so; fi
  LSM_HIP_LIB=$L timeout -k 10 300 python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>$O/err_$V.txt | line "cfg4 reservoir $V" | tee -a $O/ports.txt
done
done
