#!/bin/bash
# 8-wave gammatone workgroups (two waves per SIMD, one workgroup per CU) need four launches in flight
set -e
for W in 4 8; do
  for ST in 6 8 10 12; do
    for STAGE in frontend full; do
      GPU_MAX_HW_QUEUES=16 LSM_GT_WPB=$W timeout -k 10 120 python bench.py --steps 200 --warmup 24 --no-cpu-baseline --stage $STAGE --streams $ST 2>/dev/null \
       | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wpb', $W, 'streams', $ST, 'stage', '$STAGE', 'step_ms', d['ms_per_step'], 'clips/s', d['value'])"
    done
  done
done
