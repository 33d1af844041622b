#!/bin/bash
# single-rank RCCL rehearsal of the N>1 bench path: hardware queues x gather mode
p=29540
for Q in ${QS:-12}; do
  for G in sync; do
    p=$((p+1))
    GPU_MAX_HW_QUEUES=$Q LSM_BENCH_GATHER=$G LSM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$p RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
      timeout -k 10 300 python bench.py --no-cpu-baseline ${EXTRA} 2>/dev/null | tail -1 \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hwq', $Q, 'gather', '$G', d['value'], d['ms_per_step'])" || exit 1
  done
done
# two ranks sharing the one GPU, exchange through gloo: exercises rank > 0, the gather order and the max-over-ranks
# timing (the throughput is meaningless: the exchange goes through the host)
LSM_BENCH_BACKEND=gloo LSM_BENCH_SHARE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 40 --warmup 6 2>/dev/null | tail -1 \
  | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2 ranks on one GPU (gloo):', d['n_gpus'], 'ranks,', d['config']['sharding'], '- ok')"
