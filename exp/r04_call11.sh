#!/bin/bash
# Round 4, call 11: ring kernel counting its input drive from per-neuron channel masks in registers (uniform leak, C <= 128) against the
# packed-entry drive (LSM_RING_NO_INMASK=1 on a hooks build of the same source): parity, then cfg4 reservoir alone and whole path.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call11; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_ordered.py tests/test_gpu_fuzz.py tests/test_gpu_round3.py tests/test_gpu_sharded.py -m gpu -q --maxfail=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log | tee -a $O/summary.txt
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'))
"; }
H=exp/variants/liblsm_hooks.so
for rep in 1 2 3; do
  for V in 1 0; do
    LSM_HIP_LIB=$H LSM_RING_NO_INMASK=$V python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 reservoir no_inmask=$V" >> $O/ring_mask.txt
  done
done
for V in 1 0; do
  LSM_HIP_LIB=$H LSM_RING_NO_INMASK=$V python3 bench.py --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 whole path no_inmask=$V" >> $O/ring_mask.txt
  LSM_HIP_LIB=$H LSM_RING_NO_INMASK=$V python3 bench.py --config cfg4 --batch 256 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 B256 reservoir no_inmask=$V" >> $O/ring_mask.txt
done
cat $O/ring_mask.txt
