#!/bin/bash
# Round 5, call 15: per-neuron leak coefficients in the pair kernel (parity + timing), then the cfg4 pipeline's stream counts.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call15; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_ordered.py tests/test_gpu_configs.py -m gpu -q --maxfail=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log | tee -a $O/summary.txt
timeout -k 10 300 python3 exp/r05_leakv.py > $O/leakv.txt 2>&1; grep "^N=" $O/leakv.txt
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'))
"; }
for rep in 1 2; do
for S in "5 6" "4 6" "6 6" "3 6" "5 4" "5 8" "4 4" "6 8"; do
  set -- $S
  timeout -k 10 300 python3 bench.py --config cfg4 --fe-streams $1 --streams $2 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 whole path fe-streams=$1 streams=$2" | tee -a $O/streams.txt
done
done
