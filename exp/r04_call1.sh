#!/bin/bash
# Round 4, call 1: the whole GPU suite on the host-side changes, the driver's command, counter list, prime sweep.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call1; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest.log | tee -a $O/summary.txt
for i in 1 2; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_$i.json 2> $O/bench_driver_$i.err; echo "driver $i rc=$?" >> $O/summary.txt; done
python3 bench.py --no-cpu-baseline > $O/bench_200.json 2> $O/bench_200.err; echo "200 rc=$?" >> $O/summary.txt
for pm in 0 40 150 400; do for i in 1 2; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-unprimed --prime-ms $pm 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('prime_ms $pm ->', d['value'], d['ms_per_step'])" >> $O/prime_sweep.txt
done; done
(cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters_all.txt 2>&1)
grep -i -E "mall|hbm|umc|dram|EA0_RDREQ|EA0_WRREQ" $O/counters_all.txt | cut -c1-220 > $O/counters_mem.txt
cat $O/summary.txt; python3 -c "
import json
for f in ['bench_driver_1','bench_driver_2','bench_200']:
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d.get('unprimed'))
"; cat $O/prime_sweep.txt; wc -l $O/counters_mem.txt
