#!/bin/bash
# Round 4, call 3: whole GPU suite; coloured input masks (INMODE 3) same-box A/B; ring input drive rewritten (packed entries, top of step).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call3; mkdir -p $O
python3 -m pytest tests -m gpu -q --maxfail=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -8 $O/pytest.log | tee -a $O/summary.txt
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'), 'idle-gpu', r.get('idle_gpu_kernel_ms'), 'in-region', r.get('in_region_kernel_ms'), 'unprimed', (d.get('unprimed') or {}).get('value'))
"; }
H=exp/variants/liblsm_hooks.so
for rep in 1 2; do
  for V in 1 0; do
    LSM_HIP_LIB=$H LSM_RING_NO_INREG=$V python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 reservoir no_inreg=$V" >> $O/ring_ab.txt
  done
  LSM_HIP_LIB=exp/variants/liblsm_ring_input_twice.so python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 reservoir input drive twice" >> $O/ring_ab.txt
done
python3 bench.py --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 whole path" >> $O/ring_ab.txt
python3 bench.py --config cfg5 --batch 512 --stage reservoir --streams 1 --steps 6 --warmup 2 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg5 B512 reservoir" >> $O/ring_ab.txt
python3 bench.py --config cfg5 --stage reservoir --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg5 B4096 reservoir" >> $O/ring_ab.txt
cat $O/ring_ab.txt
for rep in 1 2 3; do
  for V in 1 0; do
    LSM_HIP_LIB=$H LSM_DENSE_NO_INCOL=$V python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "no_incol=$V driver" >> $O/incol_ab.txt
    LSM_HIP_LIB=$H LSM_DENSE_NO_INCOL=$V python3 bench.py --no-cpu-baseline --no-unprimed 2>/dev/null | line "no_incol=$V 200steps" >> $O/incol_ab.txt
  done
done
for V in 1 0; do
  LSM_HIP_LIB=$H LSM_DENSE_NO_INCOL=$V python3 bench.py --stage reservoir --streams 1 --steps 40 --warmup 5 --no-cpu-baseline --no-unprimed 2>/dev/null | line "no_incol=$V reservoir alone (8 waves)" >> $O/incol_ab.txt
  LSM_HIP_LIB=$H LSM_DENSE_NO_INCOL=$V python3 bench.py --stage reservoir --steps 200 --warmup 12 --no-cpu-baseline --no-unprimed 2>/dev/null | line "no_incol=$V reservoir stage, pipeline topology" >> $O/incol_ab.txt
done
cat $O/incol_ab.txt
