#!/bin/bash
# Round 4, call 8: does the fused front end's LDS (51 KB per 4-wave workgroup, or the 81 KB reservation) displace ring-kernel clips
# (74 KB each, two per CU) at cfg4?  Waves per front-end workgroup 4 / 2 / 1 (51 / 26 / 13 KB) and the reservation on / off.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call8; mkdir -p $O
python3 -m pytest tests/test_gpu_mel.py -m gpu -q --maxfail=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.log | tee -a $O/summary.txt
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'), 'in-region', r.get('in_region_kernel_ms'), 'fe alone', (r.get('dominant_kernel_by_time') or {}).get('frontend_idle_gpu_ms'))
"; }
H=exp/variants/liblsm_hooks.so
for rep in 1 2; do
  for V in "4 -1" "4 0" "2 -1" "1 -1"; do
    set -- $V
    E="LSM_GTF_WPB=$1"; [ "$2" != "-1" ] && E="$E LSM_GTF_LDS=$2"
    env LSM_HIP_LIB=$H $E python3 bench.py --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 whole path, fe waves per workgroup $1, reservation ${2}" >> $O/cfg4_fe_lds.txt
  done
  for FS in 2 3; do
    env LSM_HIP_LIB=$H python3 bench.py --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed --fe-streams $FS --streams 3 2>/dev/null | line "cfg4 whole path, fe_streams $FS, streams 3" >> $O/cfg4_fe_lds.txt
  done
  env LSM_HIP_LIB=$H python3 bench.py --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed --fe-streams 0 --streams 3 2>/dev/null | line "cfg4 whole path, rotation over 3 streams" >> $O/cfg4_fe_lds.txt
done
cat $O/cfg4_fe_lds.txt
for V in "4 -1" "1 -1"; do
  set -- $V
  env LSM_HIP_LIB=$H LSM_GTF_WPB=$1 python3 bench.py --config cfg5 --steps 4 --warmup 1 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg5 whole path, fe waves per workgroup $1" >> $O/cfg5_fe_lds.txt
done
cat $O/cfg5_fe_lds.txt
