#!/bin/bash
# Round-2 evidence kept under profiles/ (one gpurun call from the repo root; PMC passes never combined with API
# traces; the program itself follows `--`).  Results land in gpurun_out/prof2/; exp/r02_summarise.py (run at the
# end, on the box) derives the traffic figures, then the cfg4/cfg5 bench lines are taken WITH those figures.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof2
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
$B --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err && echo "bench (driver command) done"
$B > $OUT/bench_default.json 2> $OUT/bench_default.err && echo "bench default done"
$B --no-cpu-baseline --streams 1 > $OUT/bench_serial.json 2> $OUT/bench_serial.err && echo "bench serial done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_driver -- $B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_stats_driver.json 2> $OUT/bench_stats_driver.err && echo "stats (driver command) done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --no-cpu-baseline > $OUT/bench_stats.json 2> $OUT/bench_stats.err && echo "stats default done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -- $B --no-cpu-baseline --streams 1 > $OUT/bench_stats_serial.json 2> $OUT/bench_stats_serial.err && echo "stats serial done"
# memory-side traffic of the reservoir kernel: FETCH_SIZE / WRITE_SIZE / L2 hits, separate passes, reservoir stage alone
for CASE in "cfg2 256 auto" "cfg4 1024 auto" "cfg4 1024 dense" "cfg5 512 auto" "cfg5 4096 auto"; do
  set -- $CASE
  for P in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
    set -- $CASE; cfg=$1; bsz=$2; ker=$3
    set -- $P; name=$1; shift
    D=$OUT/pmc_${cfg}_B${bsz}_${ker}_$name
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $D -- $B --config $cfg --batch $bsz --kernel $ker --stage reservoir --steps 2 --warmup 1 --streams 1 --no-cpu-baseline > $D.json 2> $D.err && echo "pmc $cfg B=$bsz $ker $name done"
  done
done
# instruction mix of the product kernels at cfg2 (one stream) and of the ring kernel at cfg5
for P in "valu SQ_INSTS_VALU SQ_INSTS_SALU" "mem SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "smem SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "waves SQ_WAVES SQ_WAVE_CYCLES" "busy SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "wait SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  set -- $P; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/sq_cfg2_$name -- $B --steps 4 --warmup 1 --no-cpu-baseline --streams 1 > $OUT/sq_cfg2_$name.json 2> $OUT/sq_cfg2_$name.err && echo "sq cfg2 $name done"
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/sq_cfg5_$name -- $B --config cfg5 --batch 512 --stage reservoir --steps 2 --warmup 1 --streams 1 --no-cpu-baseline > $OUT/sq_cfg5_$name.json 2> $OUT/sq_cfg5_$name.err && echo "sq cfg5 $name done"
done
cd $ROOT
python3 exp/r02_summarise.py $OUT || exit 1
export LSM_TRAFFIC_FILE=$OUT/summary/lif_traffic.json
cd /tmp
$B --config cfg4 --steps 8 --warmup 2 --no-cpu-baseline > $OUT/summary/r02_cfg4.json 2> $OUT/cfg4.err && echo "cfg4 line done"
$B --config cfg5 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/summary/r02_cfg5.json 2> $OUT/cfg5.err && echo "cfg5 line done"
$B --config cfg4 --kernel dense --steps 8 --warmup 2 --no-cpu-baseline > $OUT/summary/r02_cfg4_dense.json 2> $OUT/cfg4d.err && echo "cfg4 dense line done"
$B --steps 20 --warmup 5 > $OUT/summary/r02_bench_driver.json 2> $OUT/bench_driver2.err && echo "driver line with traffic done"
ls -la $OUT/summary
