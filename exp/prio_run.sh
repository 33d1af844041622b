#!/bin/bash
# same-box A/B: reservoir kernel with (default build) and without (exp/lib_noprio.so, -DLSM_LIF_NO_PRIO=1)
# the raised wave priority during the fetch phase; alternating runs
set -e
for i in 1 2 3; do
for L in "" exp/lib_noprio.so; do
  LSM_HIP_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('lib', '${L:-default (priority phases)}', d['value'], d['ms_per_step'], 'lif in-region', r['kernel_ms'], r['frac'], 'idle', r['idle_gpu_kernel_ms'])"
done
done
