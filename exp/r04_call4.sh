#!/bin/bash
# Round 4, call 4: ring input drive with per-lane dump-word padding (A/B + cost), tail-step sweep, dense kernel phase stamps, ring/dense parity tests.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call4; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_ordered.py tests/test_gpu_fuzz.py tests/test_gpu_round3.py tests/test_gpu_hotpath.py -m gpu -q --maxfail=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log | tee -a $O/summary.txt
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
  for TS in 0 1 2 3 4; do
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-unprimed --tail-steps $TS 2>/dev/null | line "tail_steps=$TS driver" >> $O/tail.txt
  done
done
for TS in 0 1 2; do
  LSM_TAIL_LONE_LAYOUT=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-unprimed --tail-steps $TS 2>/dev/null | line "tail_steps=$TS driver, reservoir layout unchanged" >> $O/tail.txt
  python3 bench.py --no-cpu-baseline --no-unprimed --tail-steps $TS 2>/dev/null | line "tail_steps=$TS 200steps" >> $O/tail.txt
done
cat $O/tail.txt
for W in 8 4; do LSM_HIP_LIB=exp/variants/liblsm_stamp.so python3 exp/stamp_run.py $W 256 >> $O/stamps.txt 2>&1; done
cat $O/stamps.txt
