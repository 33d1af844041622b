#!/bin/bash
# Round 3: ring-row reservoir kernel, new build against the previous commit (liblsm_hip_prev.so), same box.
# (used for: row descriptors through scalar loads; three packed row words; fixed-pitch lists + early input counts)
OUT=gpurun_out/r03_ring_ab.txt
PREV=/root/repo/lsm-speech-classifier_amd/liblsm_hip_prev.so
run() {
  local label=$1; shift
  env "$@" 2>/dev/null | python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$label FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$label', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; kernel', d['config'].get('reservoir_kernel'), 'wpc', d['config'].get('waves_per_clip'), 'lone', r.get('lone_launch_kernel_ms'))
" | tee -a $OUT
}
for rep in 1 2; do
  for V in prev new; do
    L=""; [ $V = prev ] && L="LSM_HIP_LIB=$PREV"
    run "$V cfg4 B1024 reservoir" $L python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline
    run "$V cfg5 B512 reservoir" $L python3 bench.py --config cfg5 --batch 512 --stage reservoir --streams 1 --steps 6 --warmup 2 --no-cpu-baseline
  done
done
for V in prev new; do
  L=""; [ $V = prev ] && L="LSM_HIP_LIB=$PREV"
  run "$V cfg5 B4096 reservoir" $L python3 bench.py --config cfg5 --stage reservoir --streams 1 --steps 3 --warmup 1 --no-cpu-baseline
  run "$V cfg4 whole path" $L python3 bench.py --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline
done
