#!/bin/bash
# Round 3: dense reservoir kernel with 4 instead of 8 rows in flight at SL = 4 (-DLSM_DENSE_G4=4: 101 instead of 113 registers, so that
# THREE reservoir workgroups fit beside a front-end wave of 168 registers on a SIMD instead of two), same box, alternating.
OUT=gpurun_out/r03_g4.txt
G4=/root/repo/lsm-speech-classifier_amd/liblsm_hip_g4.so
run() {
  local label=$1; shift
  env "$@" 2>/dev/null | python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$label FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$label', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lif in-region', r.get('kernel_ms'), 'lone', r.get('lone_launch_kernel_ms'), 'idle(4 waves)', r.get('idle_gpu_kernel_ms'))
" | tee -a $OUT
}
for rep in 1 2 3; do
  for V in g8 g4; do
    L="LSM_X=0"; [ $V = g4 ] && L="LSM_HIP_LIB=$G4"
    run "$V --steps 20 --warmup 5" $L python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
  done
done
for rep in 1 2; do
  for V in g8 g4; do
    L="LSM_X=0"; [ $V = g4 ] && L="LSM_HIP_LIB=$G4"
    run "$V --steps 200 --warmup 12" $L python3 bench.py --steps 200 --warmup 12 --no-cpu-baseline
  done
done
