#!/bin/bash
# kernel traces of the driver's command under the two stream topologies (timeline by exp/r03_timeline.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_trace
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for V in "rot 0 6" "two 4 4" "two46 4 6"; do
  set -- $V
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$1 -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --fe-streams $2 --streams $3 > $OUT/$1.json 2> $OUT/$1.err
  f=$(find $OUT/$1 -name "*kernel_trace.csv" | head -1)
  python3 $ROOT/exp/r03_timeline.py $f 20 > $OUT/$1_timeline.txt
  tail -1 $OUT/$1.json | cut -c1-200
done
