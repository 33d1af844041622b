#!/bin/bash
# dense LIF kernel, cfg2, one stream: time of ablated builds (exp/lib_ablate_<mask>.so; results wrong by design)
set -e
run() {
  LSM_HIP_LIB=$1 timeout -k 10 120 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --stage reservoir --streams ${3:-1} 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', 'streams ${3:-1} kernel_ms', d['roofline']['kernel_ms'], 'step_ms', d['ms_per_step'], 'spikes/clip', d['config']['mean_output_spikes_per_clip'])"
}
for L in "$@"; do
  run "$L" "$(basename "$L" .so)" 1
  run "$L" "$(basename "$L" .so)" 3
done
