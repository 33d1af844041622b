#!/bin/bash
# Round 4, call 15: one-chain front end at two/three workgroups per CU inside the two-stage topology (the pairing the r03 sweep found
# fastest for the front end alone, 445 k clips/s, never re-run with front-end streams of their own and this round's reservoir kernel).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call15; mkdir -p $O
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'), 'in-region', r.get('in_region_kernel_ms'))
"; }
H=exp/variants/liblsm_hooks.so
run() {  # label, env..., -- bench args
  local label="$1"; shift
  env "$@" LSM_HIP_LIB=$H python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-unprimed $EXTRA 2>/dev/null | line "$label driver" >> $O/fe1.txt
  env "$@" LSM_HIP_LIB=$H python3 bench.py --no-cpu-baseline --no-unprimed $EXTRA 2>/dev/null | line "$label 200steps" >> $O/fe1.txt
}
EXTRA="" run "product layout (nch 2, 82944), fe_streams 5" LSM_X=0
for LDS in 82944 41000 33000; do for FS in 3 4 5 6; do
  EXTRA="--fe-streams $FS" run "nch 1 wpb 4 lds $LDS fe_streams $FS" LSM_GTF_NCH=1 LSM_GTF_LDS=$LDS
done; done
for FS in 4 6; do
  EXTRA="--fe-streams $FS --waves-per-clip 8" run "nch 1 wpb 4 lds 41000 fe_streams $FS wpc 8" LSM_GTF_NCH=1 LSM_GTF_LDS=41000
  EXTRA="--fe-streams $FS" run "nch 1 wpb 8 lds 82944 fe_streams $FS" LSM_GTF_NCH=1 LSM_GTF_WPB=8 LSM_GTF_LDS=82944
done
EXTRA="" run "product layout (nch 2, 82944), fe_streams 5 (again)" LSM_X=0
cat $O/fe1.txt
