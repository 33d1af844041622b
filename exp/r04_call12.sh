#!/bin/bash
# Round 4, call 12: front-end epilogue with the latch in registers (27 KB instead of 51 KB of LDS per workgroup) + no LDS reservation beside
# the ring kernel (whose mask form holds 64 KB per clip): parity, then cfg4 whole path with the flag on / off, cfg2 and cfg5 unchanged?
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call12; mkdir -p $O
python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_hotpath.py tests/test_gpu_graph.py tests/test_gpu_pipeline.py -m gpu -q --maxfail=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log | tee -a $O/summary.txt
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'), 'lds/clip', d['config'].get('lds_bytes_per_clip'), 'fe alone', (r.get('dominant_kernel_by_time') or {}).get('frontend_idle_gpu_ms'))
"; }
for rep in 1 2 3; do
  for V in 0 1; do
    LSM_FE_SHARE_LDS=$V python3 bench.py --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 whole path share_lds=$V" >> $O/share.txt
  done
done
for V in 0 1; do
  LSM_FE_SHARE_LDS=$V python3 bench.py --config cfg5 --steps 4 --warmup 1 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg5 whole path share_lds=$V" >> $O/share.txt
  LSM_FE_SHARE_LDS=$V python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg2 driver share_lds=$V" >> $O/share.txt
  LSM_FE_SHARE_LDS=$V python3 bench.py --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg2 200 steps share_lds=$V" >> $O/share.txt
done
python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 reservoir alone" >> $O/share.txt
cat $O/share.txt
