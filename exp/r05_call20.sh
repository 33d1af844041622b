#!/bin/bash
# Round 5, call 20: the row loop's LDS round trip split in two (replay builds, time only): regwin = the window piece summed in a
# register, reglist = the list entries summed in a register, regboth = both.  Plus: is ds_add_f32 the same function as v_add_f32?
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call20; mkdir -p $O
timeout -k 10 120 exp/micro/ds_add_check 2>&1 | tee $O/ds_add.txt
timeout -k 10 400 python3 exp/r05_residency.py record cfg4 1024 2>$O/err_record.txt | tee -a $O/rmw.txt
for rep in 1 2; do
for V in replay replay_regwin replay_reglist replay_regboth; do
  LSM_HIP_LIB=exp/variants/lib_$V.so timeout -k 10 300 python3 exp/r05_residency.py replay cfg4 1024 2>$O/err_$V.txt | tee -a $O/rmw.txt
done
done
rm -f /tmp/r05_sm.npy
