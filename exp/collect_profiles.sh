#!/bin/bash
# Collect the evidence kept under profiles/ (run through gpurun from the repo root):
#   bench line (default + one stream), rocprofv3 kernel stats of both, and the PMC passes
#   (FETCH_SIZE / WRITE_SIZE / L2 hit+miss, each in its own run, never combined with API traces).
# Results land in gpurun_out/prof/; exp/summarise_profiles.py copies the summaries into profiles/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench default done"
python3 $ROOT/bench.py --no-cpu-baseline --streams 1 > $OUT/bench_serial.json 2> $OUT/bench_serial.err
echo "bench serial done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_stats.json 2> $OUT/bench_stats.err
echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -- python3 $ROOT/bench.py --no-cpu-baseline --streams 1 > $OUT/bench_stats_serial.json 2> $OUT/bench_stats_serial.err
echo "stats serial done"
for P in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
  set -- $P; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --streams 1 > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  echo "pmc $name done"
done
