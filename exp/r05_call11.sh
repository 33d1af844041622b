#!/bin/bash
# Round 5, call 11: the whole cfg4 path (front ends overlapped with the reservoir) with the three ring forms, same box, alternating;
# and the failing full-size test again.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call11; mkdir -p $O
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'))
"; }
for rep in 1 2 3; do
  for K in ring ring-pairs-rowlists ring-quads; do
    timeout -k 10 300 python3 bench.py --config cfg4 --kernel $K --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed 2>$O/err_$K.txt | line "cfg4 whole path $K" | tee -a $O/whole.txt
  done
done
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "full_size or mel or nan" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
