#!/bin/bash
# Round 4: clocks and socket power rocm-smi reports WHILE the pipeline runs (cfg2), sampled every 0.4 s through three long runs:
# the whole overlapped path, the front ends alone on the pipeline's five streams, one front-end stream (a quarter of the chip).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_clock_power; mkdir -p $O
sample() {   # label, bench args...
  local label="$1"; shift
  ( python3 bench.py "$@" --no-cpu-baseline --no-unprimed > $O/line.json 2>/dev/null ) &
  local BP=$!
  echo "## $label: python3 bench.py $*" >> $O/out.txt
  local t0=$(date +%s.%N)
  while kill -0 $BP 2>/dev/null; do
    local s=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -E 's/.*\((.*Mhz)\).*/\1/; s/.*Power \(W\): //' | tr '\n' ' ')
    echo "t=$(python3 -c "import time; print(f'{time.time() - $t0:6.2f}')") s  $s" >> $O/out.txt
    sleep 0.4
  done
  wait $BP
  python3 -c "
import json
d = json.loads([l for l in open('$O/line.json') if l.startswith('{')][-1])
print('   ->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step over', d['steps'], 'steps')" >> $O/out.txt
}
sample "whole path" --steps 14000 --warmup 12
sample "front ends alone, five streams" --stage frontend --steps 14000 --warmup 12
sample "front ends alone, one stream (64 of 256 CUs)" --stage frontend --streams 1 --steps 3500 --warmup 12
sample "reservoir alone, six streams" --stage reservoir --steps 30000 --warmup 12
cat $O/out.txt
