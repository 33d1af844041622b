#!/bin/bash
# front-end-only time vs batch size and gammatone workgroup size (LSM_GT_WPB waves), one stream
for W in 1 4 8; do
for B in 256 512 1024 2048 4096; do
  LSM_GT_WPB=$W timeout -k 10 120 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --stage frontend --streams 1 --batch $B 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); B=$B; print('wpb', $W, 'frontend only B', B, 'step_ms', d['ms_per_step'], 'per256', round(d['ms_per_step']*256/B,4))" || exit 1
done
done
