#!/bin/bash
# Round-3 evidence kept under profiles/ (one gpurun call from the repo root; PMC passes never combined with API traces;
# the program itself follows `--`).  Results land in gpurun_out/prof3/; exp/r03_summarise.py derives the summaries.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof3
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
$B --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err && echo "bench (driver command) done"
$B > $OUT/bench_default.json 2> $OUT/bench_default.err && echo "bench default (200 steps) done"
$B --no-cpu-baseline --streams 1 > $OUT/bench_serial.json 2> $OUT/bench_serial.err && echo "bench serial done"
# same-box A/B: the two split launches of round 2 against the one-launch front end, rotation against two stages
for rep in 1 2; do
  for V in "split_rotation LSM_FRONTEND_SPLIT=1 --fe-streams 0" "fused_rotation LSM_X=0 --fe-streams 0" "fused_two_stages LSM_X=0 --fe-streams 5"; do
    set -- $V
    for A in "--steps 20 --warmup 5" "--steps 200 --warmup 12"; do
      env $2 $B $A $3 $4 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('$1 $A ->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; reservoir in-region', r['kernel_ms'], 'ms; front end alone', r['dominant_kernel_by_time']['frontend_idle_gpu_ms'], 'ms')" >> $OUT/ab_frontend.txt
    done
  done
done
echo "A/B done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_driver -- $B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_stats_driver.json 2> $OUT/bench_stats_driver.err && echo "stats (driver command) done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --no-cpu-baseline > $OUT/bench_stats.json 2> $OUT/bench_stats.err && echo "stats default done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -- $B --no-cpu-baseline --streams 1 > $OUT/bench_stats_serial.json 2> $OUT/bench_stats_serial.err && echo "stats serial done"
python3 $ROOT/exp/r03_timeline.py $(find $OUT/stats_driver -name "*kernel_trace.csv" | head -1) 20 > $OUT/timeline_driver.txt
# memory-side traffic of the reservoir kernel: FETCH_SIZE / WRITE_SIZE / L2 hits, separate passes, reservoir stage alone
for CASE in "cfg2 256 auto" "cfg4 1024 auto" "cfg5 512 auto" "cfg5 4096 auto"; do
  for P in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
    set -- $CASE; cfg=$1; bsz=$2; ker=$3
    set -- $P; name=$1; shift
    D=$OUT/pmc_${cfg}_B${bsz}_${ker}_$name
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $D -- $B --config $cfg --batch $bsz --kernel $ker --stage reservoir --steps 2 --warmup 1 --streams 1 --no-cpu-baseline > $D.json 2> $D.err && echo "pmc $cfg B=$bsz $ker $name done"
  done
done
# instruction mix: the product kernels at cfg2 (one stream) and the ring kernel at cfg4
for P in "valu SQ_INSTS_VALU SQ_INSTS_SALU" "mem SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "smem SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "waves SQ_WAVES SQ_WAVE_CYCLES" "busy SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "wait SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  set -- $P; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/sq_cfg2_$name -- $B --steps 4 --warmup 1 --no-cpu-baseline --streams 1 > $OUT/sq_cfg2_$name.json 2> $OUT/sq_cfg2_$name.err && echo "sq cfg2 $name done"
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/sq_cfg4_$name -- $B --config cfg4 --stage reservoir --steps 2 --warmup 1 --streams 1 --no-cpu-baseline > $OUT/sq_cfg4_$name.json 2> $OUT/sq_cfg4_$name.err && echo "sq cfg4 $name done"
done
cd $ROOT
python3 exp/r03_summarise.py $OUT || exit 1
export LSM_TRAFFIC_FILE=$OUT/summary/lif_traffic.json
cd /tmp
$B --config cfg1 --steps 60 --warmup 8 > $OUT/summary/r03_cfg1.json 2> $OUT/cfg1.err && echo "cfg1 line done"
$B --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline > $OUT/summary/r03_cfg4.json 2> $OUT/cfg4.err && echo "cfg4 line done"
$B --config cfg5 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/summary/r03_cfg5.json 2> $OUT/cfg5.err && echo "cfg5 line done"
$B --steps 20 --warmup 5 > $OUT/summary/r03_bench_driver.json 2> $OUT/bench_driver2.err && echo "driver line with traffic done"
cp $OUT/ab_frontend.txt $OUT/summary/r03_frontend_ab.txt
cp $OUT/timeline_driver.txt $OUT/summary/r03_timeline_driver.txt
