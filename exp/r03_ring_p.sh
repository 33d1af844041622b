#!/bin/bash
# Round 3: ring kernel (three packed row words), 6 rows in flight (-DLSM_RING_P=6: 119 registers at 2 quads per wave) against the shipped 4
OUT=gpurun_out/r03_ring_p.txt
run() {
  local label=$1; shift
  env "$@" 2>/dev/null | python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$label FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$label', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('lone_launch_kernel_ms'))
" | tee -a $OUT
}
for rep in 1 2; do
  for V in p4 p6; do
    L="LSM_X=0"; [ $V = p6 ] && L="LSM_HIP_LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip_p6.so"
    run "$V cfg4 B1024 reservoir" $L python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline
    run "$V cfg5 B512 reservoir" $L python3 bench.py --config cfg5 --batch 512 --stage reservoir --streams 1 --steps 6 --warmup 2 --no-cpu-baseline
  done
done
