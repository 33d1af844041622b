#!/bin/bash
# Round 4: the cfg4 records re-taken after the ring kernel's mask-based input drive (one call): kernel stats, memory-side traffic, instruction mix,
# phase clocks, the bench lines of cfg4 / cfg5 and the driver's command, and the whole GPU suite.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof4b
rm -rf "$OUT" && mkdir -p "$OUT/summary"
cd $ROOT
python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg4 -- $B --config cfg4 --stage reservoir --streams 1 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed > $OUT/bench_stats_cfg4.json 2> $OUT/bench_stats_cfg4.err && echo "stats cfg4 reservoir done"
for P in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
  set -- $P; name=$1; shift
  D=$OUT/pmc_cfg4_B1024_auto_$name
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $D -- $B --config cfg4 --batch 1024 --kernel auto --stage reservoir --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-unprimed --prime-ms 0 > $D.json 2> $D.err && echo "pmc cfg4 $name done"
done
for P in "valu SQ_INSTS_VALU SQ_INSTS_SALU" "mem SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "smem SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "waves SQ_WAVES SQ_WAVE_CYCLES" "busy SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "wait SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  set -- $P; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/sq_cfg4_$name -- $B --config cfg4 --stage reservoir --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-unprimed --prime-ms 0 > $OUT/sq_cfg4_$name.json 2> $OUT/sq_cfg4_$name.err && echo "sq cfg4 $name done"
done
cd $ROOT
python3 exp/r04_summarise.py $OUT > $OUT/summarise.log 2>&1 || { tail -20 $OUT/summarise.log; exit 1; }
python3 - <<PY
import json
new = json.load(open("$OUT/summary/lif_traffic.json")); old = json.load(open("$ROOT/profiles/lif_traffic.json"))
old.update({k: v for k, v in new.items() if k.startswith("cfg4")})
json.dump(old, open("$OUT/summary/lif_traffic.json", "w"), indent=1)
print({k: v for k, v in old.items() if k.startswith("cfg4") and not k.endswith("detail")})
PY
export LSM_TRAFFIC_FILE=$OUT/summary/lif_traffic.json
LSM_HIP_LIB=$ROOT/exp/variants/liblsm_phases.so python3 exp/r03_ring_phases.py cfg4 1024 > $OUT/summary/r04_ring_phases.txt 2>&1; tail -12 $OUT/summary/r04_ring_phases.txt
cd /tmp
$B --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline > $OUT/summary/r04_cfg4.json 2> $OUT/cfg4.err && echo "cfg4 line done"
$B --config cfg5 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/summary/r04_cfg5.json 2> $OUT/cfg5.err && echo "cfg5 line done"
$B --steps 20 --warmup 5 > $OUT/summary/r04_bench_driver_check.json 2> $OUT/drv.err && echo "driver line done"
for f in r04_cfg4 r04_cfg5 r04_bench_driver_check; do python3 -c "
import json
p='$OUT/summary/$f.json'
d=json.loads([l for l in open(p) if l.startswith('{')][-1]); json.dump(d, open(p,'w'), indent=1)
r=d.get('roofline',{})
print('$f', d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'kernel_ms', r.get('kernel_ms'), 'traffic_over', r.get('traffic_over_algorithmic'), 'memside', r.get('memory_side_frac'))"; done
c=$(find $OUT/stats_cfg4 -name "*kernel_stats.csv" | head -1); [ -n "$c" ] && cp $c $OUT/summary/r04_kernel_stats_cfg4_reservoir.csv && head -4 $OUT/summary/r04_kernel_stats_cfg4_reservoir.csv | cut -c1-160
