#!/bin/bash
# whole GPU suite, then the reservoir kernel alone at the two large configurations
timeout -k 10 1000 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for CFG in "cfg4 1024 8 ring" "cfg5 512 8 ring" "cfg4 1024 8 ring" "cfg5 512 8 ring"; do
  set -- $CFG
  LSM_KERNEL=$4 timeout -k 10 300 python exp/big_cfg.py $1 $2 0 $3 2>&1 | grep -E "^wpc|rror" | sed "s/^/$1 $4: /" | cut -c1-160
done
