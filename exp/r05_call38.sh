#!/bin/bash
# Round 5, call 38: threads of a spec_to_spikes_kernel workgroup (one clip) inside the overlapped cfg1 pipeline: 1024 (product) / 512 / 256, alternating.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05_call38; mkdir -p $O
for rep in 1 2 3; do
for V in product t512 t256; do
  if [ $V = product ]; then export LSM_HIP_LIB=; else export LSM_HIP_LIB=$GRAFT_REPO_ROOT/exp/variants/lib_spk_$V.so; fi
  timeout -k 10 200 python3 bench.py --config cfg1 --steps 60 --warmup 8 --no-cpu-baseline --no-unprimed > $O/$V.json 2> $O/$V.err && python3 -c "
import json
d=json.loads([l for l in open('$O/$V.json') if l.startswith('{')][-1]); print('cfg1 $V', d['value'], d['ms_per_step'])" | tee -a $O/threads.txt
done
done
