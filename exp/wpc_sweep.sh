#!/bin/bash
# waves per clip of the reservoir kernel: alone (one stream) and inside the whole pipeline
set -e
for W in ${WS:-4 8 16}; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --stage reservoir --streams 1 --steps 30 --waves-per-clip $W ${EXTRA} 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wpc', $W, 'reservoir alone kernel_ms', d['roofline']['kernel_ms'])"
  timeout -k 10 120 python bench.py --no-cpu-baseline --waves-per-clip $W ${EXTRA} 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wpc', $W, 'whole path', d['value'], d['ms_per_step'], 'lif in-region', d['roofline']['kernel_ms'])"
done
