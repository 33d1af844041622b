#!/bin/bash
# Lone-launch duration and SQ counters of the fused front end: prev (-DLSM_GTF_PRE=0) against new ({x, b0*x} pairs through scalar loads).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pre_sq
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for V in prev new; do
  if [ $V = prev ]; then export LSM_HIP_LIB=$ROOT/lsm-speech-classifier_amd/liblsm_hip_prev.so; else unset LSM_HIP_LIB; fi
  A="--stage frontend --streams 1 --steps 6 --warmup 2 --no-cpu-baseline"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${V}_stats -- python3 $ROOT/bench.py $A > $OUT/${V}_stats.json 2> $OUT/${V}_stats.err || exit 1
  for P in "valu SQ_INSTS_VALU SQ_INSTS_SALU" "smem SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "waves SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "wait SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "wait2 SQ_WAIT_ANY SQ_INST_CYCLES_SALU"; do
    set -- $P; name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${V}_$name -- python3 $ROOT/bench.py $A > $OUT/${V}_$name.json 2> $OUT/${V}_$name.err || exit 1
  done
  echo "$V done"
done
cd $ROOT
python3 - <<'PY' | tee gpurun_out/pre_sq/summary.txt
import csv, glob, collections
for V in ("prev", "new"):
    f = glob.glob(f"gpurun_out/pre_sq/{V}_stats/**/*kernel_stats.csv", recursive=True)
    for r in csv.DictReader(open(f[0])):
        if "gammatone" in r["Name"]:
            print(V, "lone launch", r["Name"][:60], "calls", r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"])
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pre_sq/{V}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gammatone" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(V, k, sum(v) / len(v))
PY
