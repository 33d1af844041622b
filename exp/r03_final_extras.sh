#!/bin/bash
# Round 3 extras in one call: extended fuzz of the fused front end, soak of the two-stage pipeline, PCIe-inclusive rate.
mkdir -p gpurun_out
python3 exp/r03_fuzz_fused.py 160 2026 > gpurun_out/r03_fuzz_fused.txt 2>&1; tail -1 gpurun_out/r03_fuzz_fused.txt
python3 exp/r03_soak.py > gpurun_out/r03_soak.txt 2>&1; tail -1 gpurun_out/r03_soak.txt
for A in "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
  python3 bench.py $A --from-host --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('from-host $A ->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step;', d['config']['inputs'])" | tee -a gpurun_out/r03_from_host.txt
  python3 bench.py $A --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('resident  $A ->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step;', d['config']['inputs'])" | tee -a gpurun_out/r03_from_host.txt
done
