#!/bin/bash
for V in base pf; do
  LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip_$V.so
  [ $V = base ] && LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip.so
  LSM_HIP_LIB=$LIB timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "full_size or all_layouts" 2>&1 | tail -1
  for A in "--stage reservoir --streams 1 --steps 50 --warmup 5" "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
  LSM_HIP_LIB=$LIB python3 bench.py $A --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = d.get('roofline', {})
print('$V $A', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lif in-region', r.get('kernel_ms'), 'lone', r.get('lone_launch_kernel_ms'), 'idle', r.get('idle_gpu_kernel_ms'))
"
  done
done
