#!/bin/bash
# Round 4, call 5: ring drive with exec-masked atomics (A/B), colouring test, FE placement sweep with the slimmer reservoir,
# waves-per-clip sweep, dense floor.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call5; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_ordered.py tests/test_gpu_fuzz.py tests/test_gpu_round3.py -m gpu -q --maxfail=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log | tee -a $O/summary.txt
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'), 'idle-gpu', r.get('idle_gpu_kernel_ms'), 'in-region', r.get('in_region_kernel_ms'))
"; }
for rep in 1 2; do
  for L in "" exp/variants/liblsm_ring_all_lanes.so exp/variants/liblsm_ring_input_twice.so; do
    LSM_HIP_LIB=$L python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 reservoir lib=${L:-product}" >> $O/ring_ab.txt
  done
done
python3 bench.py --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 whole path" >> $O/ring_ab.txt
for L in "" exp/variants/liblsm_ring_all_lanes.so; do
  LSM_HIP_LIB=$L python3 bench.py --config cfg5 --batch 512 --stage reservoir --streams 1 --steps 6 --warmup 2 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg5 B512 reservoir lib=${L:-product}" >> $O/ring_ab.txt
  LSM_HIP_LIB=$L python3 bench.py --config cfg5 --stage reservoir --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg5 B4096 reservoir lib=${L:-product}" >> $O/ring_ab.txt
done
cat $O/ring_ab.txt
H=exp/variants/liblsm_hooks.so
for rep in 1 2; do
  for LDS in 82944 60000 45000; do for FS in 5 8; do
    LSM_HIP_LIB=$H LSM_GTF_LDS=$LDS python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-unprimed --fe-streams $FS 2>/dev/null | line "fe_lds=$LDS fe_streams=$FS driver" >> $O/fe_place.txt
    LSM_HIP_LIB=$H LSM_GTF_LDS=$LDS python3 bench.py --no-cpu-baseline --no-unprimed --fe-streams $FS 2>/dev/null | line "fe_lds=$LDS fe_streams=$FS 200steps" >> $O/fe_place.txt
  done; done
done
cat $O/fe_place.txt
for rep in 1 2; do for W in 4 8; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-unprimed --waves-per-clip $W 2>/dev/null | line "waves_per_clip=$W driver" >> $O/wpc.txt
  python3 bench.py --no-cpu-baseline --no-unprimed --waves-per-clip $W 2>/dev/null | line "waves_per_clip=$W 200steps" >> $O/wpc.txt
done; done
cat $O/wpc.txt
python3 exp/r04_dense_floor.py > $O/dense_floor.txt 2>&1; grep -v amdgpu.ids $O/dense_floor.txt
