#!/bin/bash
# same-box A/B of two builds of the library: liblsm_hip_base.so (previous commit) against liblsm_hip.so
for V in ${VARIANTS:-base new base new}; do
  LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip_$V.so
  [ $V = new ] && LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip.so
  for A in "--stage reservoir --streams 1 --steps 50 --warmup 5" "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
  LSM_HIP_LIB=$LIB python3 bench.py $A --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = d.get('roofline', {})
print('$V $A', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lif in-region', r.get('kernel_ms'), 'lone', r.get('lone_launch_kernel_ms'), 'idle', r.get('idle_gpu_kernel_ms'))
"
  done
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_c_abi.py -m gpu -q -x 2>&1 | tail -2
