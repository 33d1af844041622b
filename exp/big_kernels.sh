#!/bin/bash
# large reservoirs: dense rows vs band rows vs sparse CSC (reservoir kernel only, one stream, distinct clips)
set -e
for K in ${KS:-dense ring sparse}; do
  LSM_KERNEL=$K timeout -k 10 400 python exp/big_cfg.py cfg4 1024 2 2>&1 | grep -E "wpc|bit-exact" | sed "s/^/cfg4 $K: /"
done
for K in ${KS:-dense ring sparse}; do
  LSM_KERNEL=$K timeout -k 10 400 python exp/big_cfg.py cfg5 512 1 2>&1 | grep -E "wpc|bit-exact" | sed "s/^/cfg5 $K: /"
done
