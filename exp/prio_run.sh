#!/bin/bash
# reservoir waves at raised instruction priority (s_setprio) inside the whole pipeline, vs stream count
set -e
for L in ${LIBS:-"" exp/lib_prio_1.so}; do
  for ST in ${STS:-6 7 8 10}; do
  LSM_HIP_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --streams $ST 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('lib', '${L:-default}', 'streams', $ST, d['value'], d['ms_per_step'], 'lif in-region', r['kernel_ms'], r['frac'], 'idle', r['idle_gpu_kernel_ms'])"
  done
done
