#!/bin/bash
# Round 3: wave priority of the front end's filter loop (hooks build, LSM_GTF_PRIO) against the reservoir kernel's raised-priority
# phases (liblsm_hip_hooks_noprio.so: built with -DLSM_LIF_NO_PRIO=1), both stream topologies, 20 and 200 steps.
OUT=gpurun_out/r03_prio.txt
run() {
  local label=$1; shift
  env "$@" 2>/dev/null | python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$label FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {}); g = r.get('dominant_kernel_by_time', {})
print('$label', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lif in-region', r.get('kernel_ms'), 'fe idle', g.get('frontend_idle_gpu_ms'))
" | tee -a $OUT
}
for LIB in hooks hooks_noprio; do
  for PR in 0 1 2 3; do
    for TOPO in "4 6" "0 6"; do
      set -- $TOPO
      for A in "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
        run "lib $LIB fe_prio $PR fe_streams $1 streams $2 $A" LSM_HIP_LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip_$LIB.so LSM_GTF_PRIO=$PR python3 bench.py $A --fe-streams $1 --streams $2 --no-cpu-baseline
      done
    done
  done
done
