#!/bin/bash
# bench lines kept under profiles/ (traffic figures come from the committed profiles/lif_traffic.json)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/lines
mkdir -p $OUT
cd /tmp
B="python3 $ROOT/bench.py"
$B --steps 20 --warmup 5 > $OUT/r02_bench_driver.json 2> $OUT/drv.err && echo "driver line done"
$B --config cfg4 --steps 8 --warmup 2 --no-cpu-baseline > $OUT/r02_cfg4.json 2> $OUT/cfg4.err && echo "cfg4 line done"
$B --config cfg5 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/r02_cfg5.json 2> $OUT/cfg5.err && echo "cfg5 line done"
