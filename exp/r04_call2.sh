#!/bin/bash
# Round 4, call 2: whole GPU suite; REFM (scalar-mask refractory countdown) same-box A/B; cfg4 ring ablations; step-time windows.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call2; mkdir -p $O
python3 -m pytest tests -m gpu -q --maxfail=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -12 $O/pytest.log | tee -a $O/summary.txt
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'), 'idle-gpu', r.get('idle_gpu_kernel_ms'), 'in-region', r.get('in_region_kernel_ms'), 'unprimed', (d.get('unprimed') or {}).get('value'))
"; }
H=exp/variants/liblsm_hooks.so
for rep in 1 2 3; do
  for V in 1 0; do
    LSM_HIP_LIB=$H LSM_DENSE_NO_REFM=$V python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line "no_refm=$V driver" >> $O/refm_ab.txt
    LSM_HIP_LIB=$H LSM_DENSE_NO_REFM=$V python3 bench.py --no-cpu-baseline --no-unprimed 2>/dev/null | line "no_refm=$V 200steps" >> $O/refm_ab.txt
  done
done
for V in 1 0; do
  LSM_HIP_LIB=$H LSM_DENSE_NO_REFM=$V python3 bench.py --stage reservoir --streams 1 --steps 40 --warmup 5 --no-cpu-baseline --no-unprimed 2>/dev/null | line "no_refm=$V reservoir alone (8 waves)" >> $O/refm_ab.txt
  LSM_HIP_LIB=$H LSM_DENSE_NO_REFM=$V python3 bench.py --stage reservoir --steps 200 --warmup 12 --no-cpu-baseline --no-unprimed 2>/dev/null | line "no_refm=$V reservoir stage, pipeline topology" >> $O/refm_ab.txt
  LSM_HIP_LIB=$H LSM_DENSE_NO_REFM=$V python3 bench.py --config cfg1 --no-cpu-baseline --no-unprimed 2>/dev/null | line "no_refm=$V cfg1" >> $O/refm_ab.txt
done
cat $O/refm_ab.txt
for rep in 1 2; do
  for L in "" exp/variants/liblsm_ring_input_twice.so exp/variants/liblsm_ring_nofeat.so; do
    LSM_HIP_LIB=$L python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 reservoir lib=${L:-product}" >> $O/ring_ablate.txt
  done
done
cat $O/ring_ablate.txt
python3 exp/r04_step_times.py 200 > $O/step_times.txt 2>&1; cat $O/step_times.txt
