#!/bin/bash
# Round 5, call 19: what would taking the accumulator read-modify-write out of the pair kernel's row loop buy?  Replay builds
# (exp/r05_residency.py): normw = no add / no LDS store, nolds = no LDS access at all in the row loop, regwin = window piece summed
# in a register + list entry by ds_add_f32 (the candidate design without its block switches), nolist = no list loads,
# regwin_nolist = both.  Time only; results of these builds are wrong.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call19; mkdir -p $O
timeout -k 10 400 python3 exp/r05_residency.py record cfg4 1024 2>$O/err_record.txt | tee -a $O/rmw.txt
for rep in 1 2; do
for V in replay replay_normw replay_nolds replay_regwin replay_nolist replay_regwin_nolist; do
  LSM_HIP_LIB=exp/variants/lib_$V.so timeout -k 10 300 python3 exp/r05_residency.py replay cfg4 1024 2>$O/err_$V.txt | tee -a $O/rmw.txt
done
done
rm -f /tmp/r05_sm.npy
