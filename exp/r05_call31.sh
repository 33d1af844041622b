#!/bin/bash
# Round 5, call 31: mel front end, one wave per frame, fourth form (as the third, sample loads without branches): parity,
# parts (time-only ablation builds), cfg1 line.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05_call31; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_mel.py tests/test_gpu_graph.py tests/test_gpu_c_abi.py -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/summary.txt
tail -15 $O/pytest.log | tee -a $O/summary.txt
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py"
for V in product nofft noproj nounpack none; do
  if [ $V = product ]; then export LSM_HIP_LIB=; else export LSM_HIP_LIB=$GRAFT_REPO_ROOT/exp/variants/lib_mel_$V.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$V -- $B --config cfg1 --stage frontend --steps 40 --warmup 5 --streams 1 --no-cpu-baseline --no-unprimed > $O/$V.json 2> $O/$V.err
  c=$(find $O/stats_$V -name "*kernel_stats.csv" | head -1); echo "$V: $(grep mel_power $c | cut -d, -f2-4)" | tee -a $O/parts.txt
done
export LSM_HIP_LIB=
for i in 1 2; do $B --config cfg1 --steps 60 --warmup 8 --no-cpu-baseline > $O/cfg1_$i.json 2> $O/cfg1_$i.err && python3 -c "
import json
d=json.loads([l for l in open('$O/cfg1_$i.json') if l.startswith('{')][-1]); print('cfg1', d['value'], d['ms_per_step'], 'unprimed', d['unprimed']['value'])"; done
