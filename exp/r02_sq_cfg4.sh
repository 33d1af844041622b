#!/bin/bash
# instruction mix of the ring kernel at cfg4 (B = 1024), the configuration that is neither bandwidth- nor front-end-bound
# (separate --pmc passes, the program itself after `--`); appended to profiles/r02_sq_counters.csv as config "cfg4"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_cfg4
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for P in "valu SQ_INSTS_VALU SQ_INSTS_SALU" "mem SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "smem SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "waves SQ_WAVES SQ_WAVE_CYCLES" "busy SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "wait SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  set -- $P; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/sq_cfg4_$name -- python3 $ROOT/bench.py --config cfg4 --stage reservoir --steps 2 --warmup 1 --streams 1 --no-cpu-baseline > $OUT/sq_cfg4_$name.json 2> $OUT/sq_cfg4_$name.err && echo "sq cfg4 $name done"
done
cd $ROOT
python3 - <<PY
import csv, glob, os
acc = {}
for d in sorted(glob.glob("$OUT/sq_cfg4_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "lif_ring_kernel" in r["Kernel_Name"]:
                acc.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
with open("$OUT/r02_sq_counters_cfg4.csv", "w", newline="") as out:
    wr = csv.writer(out)
    for (full, cn), v in sorted(acc.items()):
        wr.writerow(["cfg4", "lif_ring_kernel", full, cn, sum(v) / len(v), len(v)])
print(open("$OUT/r02_sq_counters_cfg4.csv").read())
PY
