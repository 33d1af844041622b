#!/bin/bash
# Re-take the committed bench lines with the final bench.py (one call, same box): driver command, default, serial, cfg1/4/5.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_lines
rm -rf $OUT && mkdir -p $OUT
export LSM_TRAFFIC_FILE=$ROOT/profiles/lif_traffic.json
B="python3 $ROOT/bench.py"
$B --steps 20 --warmup 5 > $OUT/r03_bench_driver_first.json 2>/dev/null && echo driver-first
$B > $OUT/r03_bench_default.json 2>/dev/null && echo default
$B --no-cpu-baseline --streams 1 > $OUT/r03_bench_serial.json 2>/dev/null && echo serial
$B --config cfg1 --steps 60 --warmup 8 > $OUT/r03_cfg1.json 2>/dev/null && echo cfg1
$B --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline > $OUT/r03_cfg4.json 2>/dev/null && echo cfg4
$B --config cfg5 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/r03_cfg5.json 2>/dev/null && echo cfg5
$B --steps 20 --warmup 5 > $OUT/r03_bench_driver.json 2>/dev/null && echo driver
for f in $OUT/*.json; do python3 -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); json.dump(d, open('$f','w'), indent=1)
print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('dominant_kernel_by_time',{}).get('kernel'))"; done
