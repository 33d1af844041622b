#!/bin/bash
# whole pipeline: streams x reservoir waves per clip, alternating (same box)
set -e
run() {
  LSM_GT_WPB=$1 timeout -k 10 120 python bench.py --no-cpu-baseline --streams $2 --waves-per-clip $3 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('gt wpb', $1, 'streams', $2, 'lif wpc', $3, '->', d['value'], d['ms_per_step'], 'lif in-region', r['kernel_ms'], r['frac'], 'idle', r['idle_gpu_kernel_ms'])"
}
for i in 1 2; do
run 4 6 0
run 4 8 4
run 4 8 0
run 4 6 4
run 4 7 4
done
