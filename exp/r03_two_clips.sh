#!/bin/bash
# VERDICT r2 item 5 (dense kernel at cfg2: two clips per workgroup, target lone launch <= 0.42 ms at B = 256): what a second
# clip per CU can buy.  The hardware already co-schedules two clips per CU when the batch has them (B = 512 on 256 CUs),
# without the shared step barrier a two-clip workgroup would add; a lone B = 256 launch has one clip per CU.
OUT=gpurun_out/r03_two_clips.txt
: > $OUT
for CASE in "256 0" "256 8" "256 4" "512 8" "512 4" "1024 4" "128 8" "128 4"; do
  set -- $CASE
  python3 bench.py --stage reservoir --streams 1 --batch $1 --waves-per-clip $2 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('B = $1, waves per clip', d['config']['waves_per_clip'], '(requested $2):', d['ms_per_step'], 'ms per launch =', round(d['ms_per_step'] * 256 / $1, 4), 'ms per 256 clips')" | tee -a $OUT
done
