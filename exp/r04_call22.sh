#!/bin/bash
# Round 4, call 22: more seeds of the extended fuzz runs on the final code (reservoir: exp/r04_fuzz_small.py, exp/r02_fuzz_big.py; front end: exp/r04_fuzz_frontend.py).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call22; mkdir -p $O
timeout -k 10 560 python3 exp/r04_fuzz_small.py 31 45 > $O/small_31.txt 2>&1; echo "small 31 rc=$?"; tail -1 $O/small_31.txt | cut -c1-300
timeout -k 10 320 python3 exp/r02_fuzz_big.py 23 14 > $O/big_23.txt 2>&1; echo "big 23 rc=$?"; tail -1 $O/big_23.txt | cut -c1-300
for s in 7 8; do timeout -k 10 200 python3 exp/r04_fuzz_frontend.py $s 150 40 > $O/fe_$s.txt 2>&1; echo "fe $s rc=$?"; grep "all equal\|one launch equal" $O/fe_$s.txt; done
