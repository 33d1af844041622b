#!/bin/bash
# Round 5, call 37: the same run by sixteen waves per clip (one wave per SIMD was bound by instruction latency): the whole GPU
# suite, then the cfg1 front end's kernels and the cfg1 line.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05_call37; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/summary.txt
tail -15 $O/pytest.log | tee -a $O/summary.txt
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fe -- $B --config cfg1 --stage frontend --steps 40 --warmup 5 --streams 1 --no-cpu-baseline --no-unprimed > $O/fe.json 2> $O/fe.err
c=$(find $O/stats_fe -name "*kernel_stats.csv" | head -1); head -5 $c | cut -c1-200
for i in 1 2; do $B --config cfg1 --steps 60 --warmup 8 --no-cpu-baseline > $O/cfg1_$i.json 2> $O/cfg1_$i.err && python3 -c "
import json
d=json.loads([l for l in open('$O/cfg1_$i.json') if l.startswith('{')][-1]); print('cfg1', d['value'], d['ms_per_step'], 'unprimed', d['unprimed']['value'])"; done
