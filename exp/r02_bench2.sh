#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for A in "--steps 20 --warmup 5 --frontends-in-flight 0" "--steps 20 --warmup 5 --frontends-in-flight 2" "--steps 20 --warmup 5 --frontends-in-flight 3" "--steps 20 --warmup 5 --frontends-in-flight 2 --streams 8" "--steps 20 --warmup 5 --frontends-in-flight 2 --streams 4" "--steps 200 --warmup 12 --frontends-in-flight 2" "--steps 200 --warmup 12 --frontends-in-flight 0" "--steps 20 --warmup 5 --frontends-in-flight 2" "--steps 20 --warmup 5 --frontends-in-flight 0"; do
  python3 bench.py $A --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = d.get('roofline', {})
print('$A', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lif in-region', r.get('kernel_ms'))
" | tee -a gpurun_out/r02_bench2.log
done
