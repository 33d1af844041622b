#!/bin/bash
# Round 5, call 16: extended fuzz of the pair-block kernel (three seeds) and the established large-reservoir fuzz with the round's library.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call16; mkdir -p $O
for seed in 51 52 53; do timeout -k 10 330 python3 exp/r05_fuzz_pairs.py $seed 26 > $O/pairs_$seed.txt 2>&1; echo "pairs $seed rc=$?"; tail -1 $O/pairs_$seed.txt | cut -c1-400; done
timeout -k 10 300 python3 exp/r02_fuzz_big.py 31 10 > $O/big_31.txt 2>&1; echo "big 31 rc=$?"; tail -1 $O/big_31.txt | cut -c1-300
