#!/bin/bash
# Round 5, call 10: pair kernel with the lists of four rows in one load: parity (every ring form), timing against one list load per row
# and against the quad kernel, phases.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call10; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_ordered.py tests/test_gpu_fuzz.py tests/test_gpu_round3.py tests/test_gpu_graph.py -m gpu -q --maxfail=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -8 $O/pytest.log | tee -a $O/summary.txt
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'))
"; }
for rep in 1 2; do
  for K in ring ring-pairs-rowlists ring-quads; do
    timeout -k 10 300 python3 bench.py --config cfg4 --kernel $K --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>$O/err_$K.txt | line "cfg4 reservoir $K" | tee -a $O/pairs.txt
  done
done
timeout -k 10 300 python3 bench.py --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed 2>/dev/null | line "cfg4 whole path (auto)" | tee -a $O/pairs.txt
LSM_HIP_LIB=exp/variants/lib_pair_phases.so timeout -k 10 300 python3 exp/r03_ring_phases.py cfg4 1024 > $O/phases.txt 2>&1; tail -14 $O/phases.txt
