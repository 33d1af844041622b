#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for A in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 200 --warmup 12 --no-cpu-baseline" "--steps 20 --warmup 5 --no-cpu-baseline --streams 4" "--steps 20 --warmup 5 --no-cpu-baseline --streams 8"; do
  python3 bench.py $A 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = d.get('roofline', {})
print('$A', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lif in-region', r.get('kernel_ms'), 'lone', r.get('lone_launch_kernel_ms'), 'idle', r.get('idle_gpu_kernel_ms'), 'wpc', d['config']['waves_per_clip'], 'cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('all_cores'))
" | tee -a gpurun_out/r02_bench1.log
done
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --stage reservoir --streams 1 --kernel ring 2>/dev/null | cut -c1-600 | tee -a gpurun_out/r02_bench1.log
