#!/bin/bash
# Round 5, call 23: chunks of 64 rows (possible now that no row past a chunk's end is selected).  Parity, then cfg4 alone and whole.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call23; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_ordered.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/summary.txt
tail -3 $O/pytest.log | tee -a $O/summary.txt
[ $rc -eq 0 ] || exit $rc
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'))
"; }
for rep in 1 2 3; do
  timeout -k 10 300 python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>$O/err.txt | line "cfg4 reservoir chunk64" | tee -a $O/chunk.txt
done
for rep in 1 2; do
  timeout -k 10 300 python3 bench.py --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed 2>$O/err_full.txt | line "cfg4 whole path chunk64" | tee -a $O/chunk.txt
done
