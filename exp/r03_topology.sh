#!/bin/bash
# Round 3: stream topology of pipeline.HotPath (rotation vs front ends on their own streams) x front-end layout,
# and the fused kernel's filter loop alone (LSM_GTF_SKIP_EPILOGUE, hooks build).
OUT=gpurun_out/r03_topology.txt
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
# filter loop alone
for N in 1 2; do
  run "nch $N skip-epilogue frontend 1 stream" LSM_HIP_LIB=$HOOKS LSM_GTF_NCH=$N LSM_GTF_SKIP_EPILOGUE=1 python3 bench.py --stage frontend --streams 1 --steps 30 --warmup 5 --no-cpu-baseline
  run "nch $N with epilogue frontend 1 stream" LSM_HIP_LIB=$HOOKS LSM_GTF_NCH=$N python3 bench.py --stage frontend --streams 1 --steps 30 --warmup 5 --no-cpu-baseline
done
run "split frontend 1 stream" LSM_FRONTEND_SPLIT=1 python3 bench.py --stage frontend --streams 1 --steps 30 --warmup 5 --no-cpu-baseline
for CFG in ${CFGS:-"2 4 82944" "1 4 82944" "1 4 41000" "1 8 82944"}; do
  set -- $CFG
  for TOPO in ${TOPOS:-"0 6" "2 4" "3 4" "4 4" "4 6" "6 6" "3 8"}; do
    set -- $CFG $TOPO
    for A in "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
      run "nch $1 wpb $2 lds $3 fe_streams $4 streams $5 $A" LSM_HIP_LIB=$HOOKS LSM_GTF_NCH=$1 LSM_GTF_WPB=$2 LSM_GTF_LDS=$3 python3 bench.py $A --streams $5 --fe-streams $4 --no-cpu-baseline
    done
  done
done
