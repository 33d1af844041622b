#!/bin/bash
# rehearsal of the N>1 code path of bench.py on the 1-GPU box: (a) one rank through RCCL (process group, broadcast,
# per-step all-gather on RCCL's stream, barrier), launched by torchrun as the driver does; (b) bench.py's own launcher
# with two ranks sharing the card (gloo exchange)
set -o pipefail
LSM_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -E "^\{|rror|Traceback" | cut -c1-400
LSM_BENCH_SHARE_GPU=1 LSM_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -E "^\{|rror|Traceback" | cut -c1-400
