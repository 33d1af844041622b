#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_fuzz; mkdir -p $O
timeout -k 10 500 python3 exp/r04_fuzz_small.py 11 45 > $O/small.txt 2>&1; echo "small rc=$?"; tail -2 $O/small.txt | cut -c1-300
for seed in 21 22; do timeout -k 10 300 python3 exp/r02_fuzz_big.py $seed 14 > $O/big_$seed.txt 2>&1; echo "big $seed rc=$?"; tail -1 $O/big_$seed.txt; done
