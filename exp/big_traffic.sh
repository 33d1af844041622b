#!/bin/bash
# measured HBM traffic of the reservoir kernel at the large configurations (FETCH_SIZE / WRITE_SIZE in their
# own rocprofv3 passes; duration from the same pass's kernel trace)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_big
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for CFG in "cfg4 1024" "cfg5 512"; do
  set -- $CFG
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$1_$C -- python3 $ROOT/exp/big_cfg.py $1 $2 0 > $OUT/$1_$C.log 2> $OUT/$1_$C.err && echo "$1 $C done"
  done
done
