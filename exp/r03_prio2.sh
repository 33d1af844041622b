#!/bin/bash
# reservoir waves at raised priority for the WHOLE step (-DLSM_LIF_PRIO_WHOLE=1) against the product (raised for list read + row fetch only)
OUT=gpurun_out/r03_prio2.txt
run() {
  local label=$1; shift
  env "$@" 2>/dev/null | python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$label FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {}); g = r.get('dominant_kernel_by_time', {})
print('$label', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lif in-region', r.get('kernel_ms'), 'frac', r.get('frac'))
" | tee -a $OUT
}
for rep in 1 2; do
for LIB in hooks hooks_priowhole; do
  for TOPO in "4 6" "4 4" "5 6"; do
    set -- $TOPO
    for A in "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
      run "lib $LIB fe_streams $1 streams $2 $A" LSM_HIP_LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip_$LIB.so python3 bench.py $A --fe-streams $1 --streams $2 --no-cpu-baseline
    done
  done
done
done
