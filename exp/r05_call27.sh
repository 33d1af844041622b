#!/bin/bash
# Round 5, call 27: the mel front end with one wave per frame (radix 16 / 16 / 4 passes in registers, no workgroup barrier): parity, then time.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call27; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_mel.py tests/test_gpu_graph.py tests/test_gpu_c_abi.py -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/summary.txt
tail -15 $O/pytest.log | tee -a $O/summary.txt
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py"
O=$GRAFT_REPO_ROOT/$O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fe -- $B --config cfg1 --stage frontend --steps 40 --warmup 5 --streams 1 --no-cpu-baseline --no-unprimed > $O/fe.json 2> $O/fe.err && echo "front-end stats done"
c=$(find $O/stats_fe -name "*kernel_stats.csv" | head -1); head -5 $c | cut -c1-220
$B --config cfg1 --steps 60 --warmup 8 --no-cpu-baseline > $O/cfg1.json 2> $O/cfg1.err && python3 -c "
import json
d=json.loads([l for l in open('$O/cfg1.json') if l.startswith('{')][-1]); print('cfg1', d['value'], d['ms_per_step'], 'unprimed', d['unprimed']['value'])"
$B --config cfg1 --steps 60 --warmup 8 --no-cpu-baseline > $O/cfg1b.json 2> $O/cfg1b.err && python3 -c "
import json
d=json.loads([l for l in open('$O/cfg1b.json') if l.startswith('{')][-1]); print('cfg1', d['value'], d['ms_per_step'], 'unprimed', d['unprimed']['value'])"
