#!/bin/bash
# Round 4, call 21: kernel + memory-copy trace of the one-rank RCCL rehearsal, chunked and once, to see what a chunk's collective puts on the GPU.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_call21; mkdir -p $O
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 LSM_BENCH_FORCE_DIST=1
for M in chunked once; do
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/$M -o run -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --exchange $M > $O/$M.json 2> $O/$M.err
  echo "$M rc=$?"; ls $O/$M/* | head
done
