#!/bin/bash
# Round 3, second epilogue (parallel compare pass + serial LDS latch pass): split vs fused, and the layouts worth a second look.
OUT=gpurun_out/r03_fused_sweep3.txt
HOOKS=/root/repo/lsm-speech-classifier_amd/liblsm_hip_hooks.so
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
for A in "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
  run "split streams 6 $A" LSM_FRONTEND_SPLIT=1 python3 bench.py $A --no-cpu-baseline
done
for CFG in ${CFGS:-"2 4 82944" "2 4 41000" "2 2 41000" "2 1 20000" "1 4 82944" "1 4 41000" "1 8 82944"}; do
  set -- $CFG
  for A in "--stage frontend --steps 100 --warmup 12" "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
    for ST in ${STS:-6 10}; do
      run "nch $1 wpb $2 lds $3 streams $ST $A" LSM_HIP_LIB=$HOOKS LSM_GTF_NCH=$1 LSM_GTF_WPB=$2 LSM_GTF_LDS=$3 python3 bench.py $A --streams $ST --no-cpu-baseline
    done
  done
done
