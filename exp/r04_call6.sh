#!/bin/bash
# Round 4, call 6: whole GPU suite (one-launch mel, chunked exchange, colouring, tail hint); ring input drive behind the first row loads
# against at the top of the step, with the entries in registers or streamed; cfg1 with the one-launch mel front end against the split route.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call6; mkdir -p $O
python3 -m pytest tests -m gpu -q --maxfail=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -6 $O/pytest.log | tee -a $O/summary.txt
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'), 'idle-gpu', r.get('idle_gpu_kernel_ms'), 'in-region', r.get('in_region_kernel_ms'))
"; }
H=exp/variants/liblsm_hooks.so
for rep in 1 2; do
  LSM_HIP_LIB= python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 reservoir: drive behind the first rows, entries in registers (product)" >> $O/ring_ab.txt
  LSM_HIP_LIB=$H LSM_RING_NO_INREG=1 python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 reservoir: drive behind the first rows, entries streamed" >> $O/ring_ab.txt
  LSM_HIP_LIB=exp/variants/liblsm_ring_drive_top.so python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 reservoir: drive at the top of the step, entries in registers" >> $O/ring_ab.txt
  LSM_HIP_LIB=exp/variants/liblsm_ring_input_twice.so python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 reservoir: drive behind the first rows, issued twice" >> $O/ring_ab.txt
done
for L in "" exp/variants/liblsm_ring_drive_top.so; do
  LSM_HIP_LIB=$L python3 bench.py --config cfg5 --batch 512 --stage reservoir --streams 1 --steps 6 --warmup 2 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg5 B512 reservoir lib=${L:-product}" >> $O/ring_ab.txt
  LSM_HIP_LIB=$L python3 bench.py --config cfg5 --stage reservoir --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg5 B4096 reservoir lib=${L:-product}" >> $O/ring_ab.txt
done
python3 bench.py --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 whole path" >> $O/ring_ab.txt
cat $O/ring_ab.txt
for rep in 1 2 3; do
  for V in 1 0; do
    LSM_FRONTEND_SPLIT=$V python3 bench.py --config cfg1 --steps 60 --warmup 8 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg1 whole path, frontend_split=$V" >> $O/mel_ab.txt
    LSM_FRONTEND_SPLIT=$V python3 bench.py --config cfg1 --stage frontend --steps 60 --warmup 8 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg1 front ends alone, frontend_split=$V" >> $O/mel_ab.txt
  done
done
cat $O/mel_ab.txt
for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('driver', d['value'], d['ms_per_step'], d['unprimed'])" >> $O/driver.txt; done; cat $O/driver.txt
