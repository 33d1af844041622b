#!/bin/bash
# size of the gammatone LDS reservation = how many 21.5 KB reservoir workgroups fit beside it on a CU
set -e
for i in 1 2; do
for L in 83000 100000 121000 140000; do
  LSM_GT_LDS=$L timeout -k 10 120 python bench.py --no-cpu-baseline 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('gammatone reservation', $L, '->', d['value'], d['ms_per_step'], 'lif in-region', r['kernel_ms'], r['frac'])"
done
done
