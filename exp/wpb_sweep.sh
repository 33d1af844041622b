#!/bin/bash
# gammatone workgroup size (LSM_GT_WPB waves) x LDS reservation (LSM_GT_LDS bytes, caps workgroups per
# CU) x stream count: front end alone and whole pipeline (bench default: 8 hardware queues)
set -e
for CFG in ${CFGS:-"4 83000" "4 55000" "4 0" "8 83000" "8 55000" "2 83000"}; do
  set -- $CFG
  for ST in ${STS:-6 8}; do
    for STAGE in frontend full; do
      LSM_GT_WPB=$1 LSM_GT_LDS=$2 timeout -k 10 120 python bench.py --steps 100 --warmup 12 --no-cpu-baseline --stage $STAGE --streams $ST 2>/dev/null \
       | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wpb', $1, 'lds', $2, 'streams', $ST, 'stage', '$STAGE', 'step_ms', d['ms_per_step'], 'clips/s', d['value'])"
    done
  done
done
