#!/bin/bash
# Round 5, call 36: where do spec_to_spikes_kernel's 33 us per 200 clips go?  Time-only ablation builds (-DLSM_SPK_ABLATE=1/2/4/8/15).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05_call36; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py"
for V in product 1 2 4 8 15; do
  if [ $V = product ]; then export LSM_HIP_LIB=; else export LSM_HIP_LIB=$GRAFT_REPO_ROOT/exp/variants/lib_spk_$V.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$V -- $B --config cfg1 --stage frontend --steps 40 --warmup 5 --streams 1 --no-cpu-baseline --no-unprimed > $O/$V.json 2> $O/$V.err
  c=$(find $O/stats_$V -name "*kernel_stats.csv" | head -1); echo "$V: $(grep spec_to_spikes $c | cut -d, -f2-4)" | tee -a $O/parts.txt
done
