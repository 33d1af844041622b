#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_trace3
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for V in "fe5 5 1" "fe4 4 0"; do
  set -- $V
  LSM_FE_WIDE_WHEN_IDLE=$3 rocprofv3 --kernel-trace --output-format csv -d $OUT/$1 -- python3 $ROOT/bench.py --stage frontend --fe-streams $2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/$1.json 2> $OUT/$1.err
  tail -1 $OUT/$1.json | cut -c1-140
done
