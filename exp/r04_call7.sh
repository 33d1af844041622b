#!/bin/bash
# Round 4, call 7: one-launch mel front end with G frames per workgroup (release fences per workgroup) against the split route.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call7; mkdir -p $O
python3 -m pytest tests/test_gpu_mel.py tests/test_gpu_ordered.py tests/test_gpu_configs.py -m gpu -q --maxfail=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log | tee -a $O/summary.txt
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step')
"; }
H=exp/variants/liblsm_hooks.so
for rep in 1 2; do
  LSM_HIP_LIB=$H LSM_FRONTEND_SPLIT=1 python3 bench.py --config cfg1 --steps 60 --warmup 8 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg1 whole path, split route" >> $O/mel_ab.txt
  LSM_HIP_LIB=$H LSM_FRONTEND_SPLIT=1 python3 bench.py --config cfg1 --stage frontend --steps 60 --warmup 8 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg1 front ends alone, split route" >> $O/mel_ab.txt
  LSM_HIP_LIB=$H LSM_FRONTEND_SPLIT=1 python3 bench.py --config cfg1 --stage frontend --streams 1 --steps 60 --warmup 8 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg1 front end, one stream, split route" >> $O/mel_ab.txt
  for G in 1 4 8 16 32 101; do
    LSM_HIP_LIB=$H LSM_MEL_FRAMES_PER_WG=$G python3 bench.py --config cfg1 --steps 60 --warmup 8 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg1 whole path, one launch, $G frames per workgroup" >> $O/mel_ab.txt
    LSM_HIP_LIB=$H LSM_MEL_FRAMES_PER_WG=$G python3 bench.py --config cfg1 --stage frontend --steps 60 --warmup 8 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg1 front ends alone, one launch, $G frames per workgroup" >> $O/mel_ab.txt
    LSM_HIP_LIB=$H LSM_MEL_FRAMES_PER_WG=$G python3 bench.py --config cfg1 --stage frontend --streams 1 --steps 60 --warmup 8 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg1 front end, one stream, one launch, $G frames per workgroup" >> $O/mel_ab.txt
  done
done
cat $O/mel_ab.txt
