#!/bin/bash
# Round 4, call 16: extended fuzz of the front end (exp/r04_fuzz_frontend.py), two seeds.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call16; mkdir -p $O
timeout -k 10 420 python3 exp/r04_fuzz_frontend.py 5 70 24 > $O/fe_5.txt 2>&1; echo "seed 5 rc=$?"; tail -3 $O/fe_5.txt | cut -c1-300
timeout -k 10 420 python3 exp/r04_fuzz_frontend.py 6 70 24 > $O/fe_6.txt 2>&1; echo "seed 6 rc=$?"; tail -3 $O/fe_6.txt | cut -c1-300
