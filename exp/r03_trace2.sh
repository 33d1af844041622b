#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_trace2
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/def -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/def.json 2> $OUT/def.err
tail -1 $OUT/def.json | cut -c1-160
