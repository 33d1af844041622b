#!/bin/bash
# Round-4 evidence kept under profiles/ (one gpurun call from the repo root; PMC passes never combined with API traces;
# the program itself follows `--`).  Results land in gpurun_out/prof4/; exp/r04_summarise.py derives the summaries.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof4
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
$B --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err && echo "bench (driver command) done"
$B > $OUT/bench_default.json 2> $OUT/bench_default.err && echo "bench default (200 steps) done"
$B --no-cpu-baseline --no-unprimed --streams 1 > $OUT/bench_serial.json 2> $OUT/bench_serial.err && echo "bench serial done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_driver -- $B --steps 20 --warmup 5 --no-cpu-baseline --no-unprimed > $OUT/bench_stats_driver.json 2> $OUT/bench_stats_driver.err && echo "stats (driver command) done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --no-cpu-baseline --no-unprimed > $OUT/bench_stats.json 2> $OUT/bench_stats.err && echo "stats default done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -- $B --no-cpu-baseline --no-unprimed --streams 1 > $OUT/bench_stats_serial.json 2> $OUT/bench_stats_serial.err && echo "stats serial done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg4 -- $B --config cfg4 --stage reservoir --streams 1 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed > $OUT/bench_stats_cfg4.json 2> $OUT/bench_stats_cfg4.err && echo "stats cfg4 reservoir done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg1 -- $B --config cfg1 --steps 60 --warmup 8 --no-cpu-baseline --no-unprimed > $OUT/bench_stats_cfg1.json 2> $OUT/bench_stats_cfg1.err && echo "stats cfg1 done"
# memory-side traffic of the reservoir kernel: FETCH_SIZE / WRITE_SIZE / L2 hits, separate passes, reservoir stage alone,
# the launch the product makes (clips started longest first above one clip per CU)
for CASE in "cfg2 256 auto" "cfg4 1024 auto" "cfg5 512 auto" "cfg5 4096 auto"; do
  for P in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
    set -- $CASE; cfg=$1; bsz=$2; ker=$3
    set -- $P; name=$1; shift
    D=$OUT/pmc_${cfg}_B${bsz}_${ker}_$name
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $D -- $B --config $cfg --batch $bsz --kernel $ker --stage reservoir --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-unprimed --prime-ms 0 > $D.json 2> $D.err && echo "pmc $cfg B=$bsz $ker $name done"
  done
done
# instruction mix: the product kernels at cfg2 (one stream) and the ring kernel at cfg4
for P in "valu SQ_INSTS_VALU SQ_INSTS_SALU" "mem SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "smem SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "waves SQ_WAVES SQ_WAVE_CYCLES" "busy SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "wait SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  set -- $P; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/sq_cfg2_$name -- $B --steps 4 --warmup 1 --no-cpu-baseline --no-unprimed --prime-ms 0 --streams 1 > $OUT/sq_cfg2_$name.json 2> $OUT/sq_cfg2_$name.err && echo "sq cfg2 $name done"
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/sq_cfg4_$name -- $B --config cfg4 --stage reservoir --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-unprimed --prime-ms 0 > $OUT/sq_cfg4_$name.json 2> $OUT/sq_cfg4_$name.err && echo "sq cfg4 $name done"
done
cd $ROOT
python3 exp/r04_summarise.py $OUT > $OUT/summarise.log 2>&1 || { tail -20 $OUT/summarise.log; exit 1; }
export LSM_TRAFFIC_FILE=$OUT/summary/lif_traffic.json
cd /tmp
$B --config cfg1 --steps 60 --warmup 8 > $OUT/summary/r04_cfg1.json 2> $OUT/cfg1.err && echo "cfg1 line done"
$B --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline > $OUT/summary/r04_cfg4.json 2> $OUT/cfg4.err && echo "cfg4 line done"
$B --config cfg5 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/summary/r04_cfg5.json 2> $OUT/cfg5.err && echo "cfg5 line done"
$B --steps 20 --warmup 5 > $OUT/summary/r04_bench_driver.json 2> $OUT/bench_driver2.err && echo "driver line with traffic done"
$B > $OUT/summary/r04_bench_default.json 2> $OUT/bench_default2.err && echo "default line with traffic done"
for f in r04_cfg1 r04_cfg4 r04_cfg5 r04_bench_driver r04_bench_default; do python3 -c "
import json
p='$OUT/summary/$f.json'
d=json.loads([l for l in open(p) if l.startswith('{')][-1]); json.dump(d, open(p,'w'), indent=1)
r=d.get('roofline',{})
print('$f', d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'pipeline_frac', r.get('pipeline_frac'), 'unprimed', (d.get('unprimed') or {}).get('value'))"; done
for f in stats_cfg4 stats_cfg1; do c=$(find $OUT/$f -name "*kernel_stats.csv" | head -1); [ -n "$c" ] && cp $c $OUT/summary/r04_kernel_${f}.csv; done
tail -40 $OUT/summarise.log
