#!/bin/bash
# whole pipeline / front end alone vs number of streams in the rotation
set -e
for ST in 1 2 3 5 6 8; do
  for STAGE in frontend full; do
    timeout -k 10 120 python bench.py --steps 60 --warmup 12 --no-cpu-baseline --stage $STAGE --streams $ST 2>/dev/null \
     | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('streams', $ST, 'stage', '$STAGE', 'step_ms', d['ms_per_step'], 'clips/s', d['value'], 'lif in-region', r.get('kernel_ms'), 'idle', r.get('idle_gpu_kernel_ms'))"
  done
done
