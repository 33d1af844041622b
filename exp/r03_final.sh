#!/bin/bash
# Last call of the round: the whole GPU suite, the bench lines of the final code, and a kernel trace of the cfg4 reservoir stage
# (what the two ranking launches of lsm_reservoir_run_ordered cost next to the LIF launch).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $ROOT/gpurun_out/r03_pytest_gpu_final.log 2>&1; tail -3 $ROOT/gpurun_out/r03_pytest_gpu_final.log
timeout -k 10 500 bash exp/r03_lines.sh || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/stats_cfg4 && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/stats_cfg4 -- python3 $ROOT/bench.py --config cfg4 --stage reservoir --streams 1 --steps 6 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/stats_cfg4.json 2> $ROOT/gpurun_out/stats_cfg4.err
cp $(find $ROOT/gpurun_out/stats_cfg4 -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/r03_kernel_stats_cfg4_reservoir.csv && cat $ROOT/gpurun_out/r03_kernel_stats_cfg4_reservoir.csv | cut -c1-150
