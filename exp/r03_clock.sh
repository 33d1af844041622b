#!/bin/bash
# Is the float64 front end clock/power-limited?  Per-launch duration of the fused front end with 1..4 launches in flight on
# disjoint CUs (64 CUs each: CU-exclusive placement), and the clocks/power rocm-smi reports during a long front-end-only run.
OUT=gpurun_out/r03_clock.txt
: > $OUT
for N in 1 2 3 4; do
  python3 bench.py --stage frontend --streams $N --fe-streams 0 --steps 60 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('front ends in flight $N: ms/step', d['ms_per_step'], '-> per launch', round(d['ms_per_step'] * $N, 4), 'ms')" | tee -a $OUT
done
( python3 bench.py --stage frontend --streams 4 --fe-streams 0 --steps 4000 --warmup 8 --no-cpu-baseline > /dev/null 2>&1 ) &
BP=$!
sleep 6
for i in 1 2 3 4 5; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr -s ' ' | head -6 >> $OUT
  echo "--" >> $OUT
  sleep 0.3
done
wait $BP
echo "idle:" >> $OUT
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr -s ' ' | head -6 >> $OUT
cat $OUT
