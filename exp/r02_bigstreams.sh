#!/bin/bash
# whole path at the large configurations against the depth of the stream rotation
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -2
for A in "cfg4 8 2 1" "cfg4 8 2 2" "cfg4 8 2 3" "cfg4 8 2 6" "cfg5 4 1 1" "cfg5 4 1 2" "cfg5 4 1 3" "cfg5 4 1 6"; do
  set -- $A
  python3 bench.py --config $1 --steps $2 --warmup $3 --streams $4 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = d.get('roofline', {})
print('$1 streams $4 ->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lif in-region', r.get('kernel_ms'), 'frac', r.get('frac'), 'lone', r.get('lone_launch_kernel_ms'))
"
done
