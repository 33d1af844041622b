#!/bin/bash
# Instruction-mix counters of the three product kernels (one stream, 4 steps), each group in its own
# rocprofv3 pass (never combined with API traces).  Output: gpurun_out/${SQ_DIR:-prof_sq}/<group>/...
# SQ_EXTRA: extra bench.py arguments (e.g. "--waves-per-clip 4" for the reservoir layout of the pipeline)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${SQ_DIR:-prof_sq}
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for P in "valu SQ_INSTS_VALU SQ_INSTS_SALU" "mem SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "smem SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "waves SQ_WAVES SQ_WAVE_CYCLES" "busy SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  set -- $P; name=$1; shift
  if rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --streams 1 ${SQ_EXTRA} > $OUT/$name.json 2> $OUT/$name.err; then
    echo "pass $name done"
  else
    echo "pass $name FAILED (see $name.err)"
  fi
done
