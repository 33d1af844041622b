#!/bin/bash
# Round 4, call 10: ring kernel with lane masks from the compares (registers 114 -> 102 / 158 -> 130): parity, then rows in flight 4 / 6 / 8 x entries
# in registers or streamed, cfg4 and cfg5, against the build before the change.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call10; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_ordered.py tests/test_gpu_fuzz.py tests/test_gpu_round3.py -m gpu -q --maxfail=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log | tee -a $O/summary.txt
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'))
"; }
V=exp/variants
for rep in 1 2; do
  for C in "before $V/liblsm_hooks.so 0" "masks_P4_inreg $V/liblsm_hooks_new.so 0" "masks_P4_streamed $V/liblsm_hooks_new.so 1" "masks_P6_inreg $V/liblsm_ring_p6.so 0" "masks_P6_streamed $V/liblsm_ring_p6.so 1" "masks_P8_streamed $V/liblsm_ring_p8.so 1"; do
    set -- $C
    LSM_HIP_LIB=$2 LSM_RING_NO_INREG=$3 python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 reservoir $1" >> $O/ring_p.txt
  done
done
for C in "before $V/liblsm_hooks.so 0" "masks_P4 $V/liblsm_hooks_new.so 0" "masks_P6 $V/liblsm_ring_p6.so 0" "masks_P8 $V/liblsm_ring_p8.so 0"; do
  set -- $C
  LSM_HIP_LIB=$2 python3 bench.py --config cfg5 --batch 512 --stage reservoir --streams 1 --steps 6 --warmup 2 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg5 B512 reservoir $1" >> $O/ring_p.txt
  LSM_HIP_LIB=$2 python3 bench.py --config cfg5 --stage reservoir --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg5 B4096 reservoir $1" >> $O/ring_p.txt
  LSM_HIP_LIB=$2 LSM_RING_NO_INREG=1 python3 bench.py --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 whole path $1 (streamed)" >> $O/ring_p.txt
done
cat $O/ring_p.txt
