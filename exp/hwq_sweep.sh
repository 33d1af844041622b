#!/bin/bash
# hardware queue count (GPU_MAX_HW_QUEUES) x streams: whole pipeline at cfg2
set -e
for Q in ${QS:-6 8 12 16}; do
  for ST in ${STS:-5 6 7 9 12}; do
    GPU_MAX_HW_QUEUES=$Q timeout -k 10 120 python bench.py --steps ${STEPS:-200} --warmup ${WARM:-24} --no-cpu-baseline --streams $ST 2>/dev/null \
     | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('hwq', $Q, 'streams', $ST, 'step_ms', d['ms_per_step'], 'clips/s', d['value'], 'lif in-region', r['kernel_ms'])"
  done
done
