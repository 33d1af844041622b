#!/bin/bash
# Round-5 evidence kept under profiles/ (one gpurun call from the repo root; PMC passes never combined with API traces; the program
# itself follows `--`).  Results land in gpurun_out/prof5/; exp/r05_summarise.py derives the summaries.  The cfg2 kernels are round 4's
# (the front end gained one NaN flag), so the cfg2 counter passes are limited to the traffic of the dense kernel; cfg4 is re-taken in full
# (its reservoir kernel is new: lif_pair_kernel), cfg5 gets its bench line only (kernel unchanged).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof5
rm -rf "$OUT" && mkdir -p "$OUT/summary"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
$B --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err && echo "bench (driver command) done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_driver -- $B --steps 20 --warmup 5 --no-cpu-baseline --no-unprimed > $OUT/bench_stats_driver.json 2> $OUT/bench_stats_driver.err && echo "stats (driver command) done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -- $B --no-cpu-baseline --no-unprimed --streams 1 > $OUT/bench_stats_serial.json 2> $OUT/bench_stats_serial.err && echo "stats serial done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg4 -- $B --config cfg4 --stage reservoir --streams 1 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed > $OUT/bench_stats_cfg4.json 2> $OUT/bench_stats_cfg4.err && echo "stats cfg4 reservoir done"
for CASE in "cfg2 256 auto" "cfg4 1024 auto"; do
  for P in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
    set -- $CASE; cfg=$1; bsz=$2; ker=$3
    set -- $P; name=$1; shift
    D=$OUT/pmc_${cfg}_B${bsz}_${ker}_$name
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $D -- $B --config $cfg --batch $bsz --kernel $ker --stage reservoir --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-unprimed --prime-ms 0 > $D.json 2> $D.err && echo "pmc $cfg B=$bsz $ker $name done"
  done
done
for P in "valu SQ_INSTS_VALU SQ_INSTS_SALU" "mem SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "smem SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "waves SQ_WAVES SQ_WAVE_CYCLES" "busy SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "wait SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  set -- $P; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/sq_cfg4_$name -- $B --config cfg4 --stage reservoir --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-unprimed --prime-ms 0 > $OUT/sq_cfg4_$name.json 2> $OUT/sq_cfg4_$name.err && echo "sq cfg4 $name done"
done
cd $ROOT
python3 exp/r05_summarise.py $OUT > $OUT/summarise.log 2>&1 || { tail -20 $OUT/summarise.log; exit 1; }
python3 - <<PY
import json
new = json.load(open("$OUT/summary/lif_traffic.json")); old = json.load(open("$ROOT/profiles/lif_traffic.json"))
old.update({k: v for k, v in new.items() if k.startswith(("cfg2", "cfg4", "_how"))})
old["_how"] = new["_how"].replace("r05_profiles.sh", "r05_profiles.sh (cfg2, cfg4; the cfg5 entries are round 4's: kernel unchanged)")
json.dump(old, open("$OUT/summary/lif_traffic.json", "w"), indent=1)
print({k: v for k, v in old.items() if not k.endswith("detail") and not k.startswith("_")})
PY
export LSM_TRAFFIC_FILE=$OUT/summary/lif_traffic.json
LSM_HIP_LIB=$ROOT/exp/variants/lib_pair_phases.so python3 exp/r03_ring_phases.py cfg4 1024 > $OUT/summary/r05_pair_phases.txt 2>&1; tail -12 $OUT/summary/r05_pair_phases.txt
cd /tmp
$B --config cfg1 --steps 60 --warmup 8 > $OUT/summary/r05_cfg1.json 2> $OUT/cfg1.err && echo "cfg1 line done"
$B --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline > $OUT/summary/r05_cfg4.json 2> $OUT/cfg4.err && echo "cfg4 line done"
$B --config cfg5 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/summary/r05_cfg5.json 2> $OUT/cfg5.err && echo "cfg5 line done"
$B --steps 20 --warmup 5 > $OUT/summary/r05_bench_driver.json 2> $OUT/bench_driver2.err && echo "driver line with traffic done"
$B > $OUT/summary/r05_bench_default.json 2> $OUT/bench_default2.err && echo "default line with traffic done"
for f in r05_cfg1 r05_cfg4 r05_cfg5 r05_bench_driver r05_bench_default; do python3 -c "
import json
p='$OUT/summary/$f.json'
d=json.loads([l for l in open(p) if l.startswith('{')][-1]); json.dump(d, open(p,'w'), indent=1)
r=d.get('roofline',{})
print('$f', d['value'], d['ms_per_step'], 'bound', r.get('bound'), 'frac', r.get('frac'), 'hbm_frac_measured', r.get('hbm_frac_measured'), 'kernel_ms', r.get('kernel_ms'), 'unprimed', (d.get('unprimed') or {}).get('value'))"; done
for f in stats_cfg4; do c=$(find $OUT/$f -name "*kernel_stats.csv" | head -1); [ -n "$c" ] && cp $c $OUT/summary/r05_kernel_stats_cfg4_reservoir.csv; done
tail -30 $OUT/summarise.log
