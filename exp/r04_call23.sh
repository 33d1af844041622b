#!/bin/bash
# Round 4, call 23: a soak of the extended fuzz runs on the final code: 6 + 4 + 6 more seeds.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call23; mkdir -p $O
for s in 32 33 34 35 36 37; do timeout -k 10 300 python3 exp/r04_fuzz_small.py $s 45 > $O/small_$s.txt 2>&1; echo "small $s rc=$? $(tail -1 $O/small_$s.txt | cut -c1-60)"; done
for s in 24 25 26 27; do timeout -k 10 320 python3 exp/r02_fuzz_big.py $s 14 > $O/big_$s.txt 2>&1; echo "big $s rc=$? $(tail -1 $O/big_$s.txt | cut -c1-60)"; done
for s in 9 10 11 12 13 14; do timeout -k 10 200 python3 exp/r04_fuzz_frontend.py $s 150 40 > $O/fe_$s.txt 2>&1; echo "fe $s rc=$? $(grep 'all equal' $O/fe_$s.txt | cut -c1-70) $(grep 'one launch equal' $O/fe_$s.txt)"; done
