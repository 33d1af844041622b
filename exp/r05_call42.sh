#!/bin/bash
# Round 5, call 42: memory-side traffic of cfg1's reservoir launch (three separate PMC passes, as exp/r05_profiles.sh takes cfg2's and cfg4's), so that
# the cfg1 line carries roofline.traffic / hbm_frac_measured too.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof5_cfg1
rm -rf "$OUT" && mkdir -p "$OUT/summary"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
for P in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum"; do
  set -- $P; name=$1; shift
  D=$OUT/pmc_cfg1_B200_auto_$name
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $D -- $B --config cfg1 --batch 200 --kernel auto --stage reservoir --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-unprimed --prime-ms 0 > $D.json 2> $D.err && echo "pmc cfg1 $name done"
done
cd $ROOT
python3 exp/r05_summarise.py $OUT > $OUT/summarise.log 2>&1 || { tail -20 $OUT/summarise.log; exit 1; }
python3 - <<PY
import json
new = json.load(open("$OUT/summary/lif_traffic.json")); old = json.load(open("$ROOT/profiles/lif_traffic.json"))
old.update({k: v for k, v in new.items() if k.startswith("cfg1")})
json.dump(old, open("$OUT/summary/lif_traffic.json", "w"), indent=1)
print({k: v for k, v in old.items() if k.startswith("cfg1")})
PY
export LSM_TRAFFIC_FILE=$OUT/summary/lif_traffic.json
cd /tmp
$B --config cfg1 --steps 60 --warmup 8 > $OUT/summary/r05_cfg1.json 2> $OUT/cfg1.err && python3 -c "
import json
p='$OUT/summary/r05_cfg1.json'
d=json.loads([l for l in open(p) if l.startswith('{')][-1]); json.dump(d, open(p,'w'), indent=1); r=d['roofline']
print('cfg1', d['value'], d['ms_per_step'], 'frac', r['frac'], 'hbm_frac_measured', r.get('hbm_frac_measured'), 'traffic', r.get('traffic'), 'cpu', d.get('cpu_baseline',{}).get('value'))"
