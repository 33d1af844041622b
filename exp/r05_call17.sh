#!/bin/bash
# Round 5, call 17 (first half): where does mel_power_kernel's time go -- the five FFT passes, the unpack / power step or
# the mel projection?  Time-only ablation builds (-DLSM_MEL_ABLATE=1 / 4 / 2) against the product on the cfg1 front end alone, kernel
# times from rocprofv3 --kernel-trace --stats; instruction mix of the product kernel; then the parity tests touched since the last full run.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT || exit 1
O=$ROOT/gpurun_out/r05_call17; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_mel.py -m gpu -q --maxfail=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
for V in product mel_nofft mel_nounpack mel_noproj; do
  if [ $V = product ]; then export LSM_HIP_LIB=""; else export LSM_HIP_LIB=$ROOT/exp/variants/lib_$V.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$V -- python3 $ROOT/bench.py --config cfg1 --stage frontend --steps 40 --warmup 5 --streams 1 --no-cpu-baseline --no-unprimed > $O/bench_$V.json 2> $O/bench_$V.err
  c=$(find $O/stats_$V -name "*kernel_stats.csv" | head -1)
  echo "== $V" | tee -a $O/mel_parts.txt
  grep -E "mel_power_kernel|power_to_db_kernel|spec_to_spikes_kernel" $c | cut -d, -f1-4 | tee -a $O/mel_parts.txt
done
export LSM_HIP_LIB=""
for P in "valu SQ_INSTS_VALU SQ_INSTS_SALU" "mem SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "waves SQ_WAVES SQ_WAVE_CYCLES" "busy SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "wait SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  set -- $P; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/sq_$name -- python3 $ROOT/bench.py --config cfg1 --stage frontend --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-unprimed --prime-ms 0 > $O/sq_$name.json 2> $O/sq_$name.err
  c=$(find $O/sq_$name -name "*counter_collection.csv" | head -1)
  python3 - "$c" <<'PY' | tee -a $O/mel_parts.txt
import csv, sys
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "mel_power_kernel" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"mel_power_kernel {k}: mean per launch {sum(v) / len(v):.0f} over {len(v)} launches")
PY
done
