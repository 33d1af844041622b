#!/bin/bash
# gammatone placement with the FINAL kernels (hooks build): waves per workgroup x LDS reservation (caps workgroups per CU)
# x streams, whole pipeline at 20 and 200 steps.  83000 = the product's choice (one workgroup per CU).
export LSM_HIP_LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip_hooks.so
for CFG in "4 83000" "4 41000" "4 0" "8 83000" "8 41000" "8 0" "4 83000"; do
  set -- $CFG
  for ST in 6 8; do
    for S in "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
      LSM_GT_WPB=$1 LSM_GT_LDS=$2 timeout -k 10 120 python3 bench.py $S --no-cpu-baseline --streams $ST 2>/dev/null \
       | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('wpb', $1, 'lds', $2, 'streams', $ST, '$S', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step')"
    done
  done
done
