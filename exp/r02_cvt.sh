#!/bin/bash
# gammatone with the scalar float32->float64 conversion (liblsm_hip.so) against the previous commit (liblsm_hip_base.so)
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -m gpu -q -x -k "gammatone or frontend or front or cfg5 or fuzz" 2>&1 | tail -2
for V in base new base new; do
  LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip_$V.so
  [ $V = new ] && LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip.so
  for A in "--stage frontend --streams 1 --steps 30 --warmup 5" "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
  LSM_HIP_LIB=$LIB python3 bench.py $A --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = d.get('roofline', {}); g = r.get('dominant_kernel_by_time', {})
print('$V $A', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; gammatone idle', g.get('idle_gpu_ms'), 'pipeline frac', g.get('pipeline_frac'))
"
  done
done
