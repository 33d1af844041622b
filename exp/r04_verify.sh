#!/bin/bash
# Round 4: the whole GPU suite, smoke() and the driver's command on the final code.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_verify; mkdir -p $O
python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest.log | tee -a $O/summary.txt
python3 __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -2 $O/smoke.log | tee -a $O/summary.txt
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench $i rc=$?" >> $O/summary.txt; python3 -c "
import json; d=json.loads([l for l in open('$O/bench_$i.json') if l.startswith('{')][-1]); print('driver', d['value'], d['ms_per_step'], 'unprimed', d['unprimed']['value'], 'frac', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'])" | tee -a $O/summary.txt; done
