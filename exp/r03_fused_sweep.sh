#!/bin/bash
# Round 3: the one-launch front end (lsm_gammatone_spikes_f64) against the two split launches, same box, and its
# layout knobs (hooks build: LSM_GTF_NCH chains per lane, LSM_GTF_WPB waves per workgroup, LSM_GTF_LDS reservation).
# Usage: exp/r03_fused_sweep.sh [ab|sweep|all]
OUT=gpurun_out/r03_fused_sweep.txt
HOOKS=/root/repo/lsm-speech-classifier_amd/liblsm_hip_hooks.so
run() {   # label, env..., -- bench args
  local label=$1; shift
  env "$@" 2>/dev/null | python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$label FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {}); g = r.get('dominant_kernel_by_time', {})
print('$label', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lif in-region', r.get('kernel_ms'), 'fe idle', g.get('frontend_idle_gpu_ms'))
" | tee -a $OUT
}
MODE=${1:-all}
if [ $MODE = ab ] || [ $MODE = all ]; then
  for rep in 1 2; do
    for A in "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
      run "split $A" LSM_FRONTEND_SPLIT=1 python3 bench.py $A --no-cpu-baseline
      run "fused $A" python3 bench.py $A --no-cpu-baseline
    done
  done
fi
if [ $MODE = sweep ] || [ $MODE = all ]; then
  for CFG in ${CFGS:-"2 4 82944" "2 4 0" "2 4 41000" "2 2 82944" "2 2 41000" "2 1 82944" "2 1 41000" "2 1 20000" "2 8 82944" "1 4 82944" "1 4 41000" "1 8 82944"}; do
    set -- $CFG
    for A in "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
      run "nch $1 wpb $2 lds $3 $A" LSM_HIP_LIB=$HOOKS LSM_GTF_NCH=$1 LSM_GTF_WPB=$2 LSM_GTF_LDS=$3 python3 bench.py $A --no-cpu-baseline
    done
  done
fi
