#!/bin/bash
# Round 5, call 28: parts of the wave-per-frame mel_power_kernel (time-only ablation builds, as call 17).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05_call28; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py"
for V in product nofft noproj nounpack none; do
  if [ $V = product ]; then export LSM_HIP_LIB=; else export LSM_HIP_LIB=$GRAFT_REPO_ROOT/exp/variants/lib_mel_$V.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$V -- $B --config cfg1 --stage frontend --steps 40 --warmup 5 --streams 1 --no-cpu-baseline --no-unprimed > $O/$V.json 2> $O/$V.err
  c=$(find $O/stats_$V -name "*kernel_stats.csv" | head -1); echo "$V: $(grep mel_power $c | cut -d, -f2-4)" | tee -a $O/parts.txt
done
