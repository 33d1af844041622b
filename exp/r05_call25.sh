#!/bin/bash
# Round 5, call 25: extended fuzz of the pair kernel's final row loop (exact requests, chunks of 64; three new seeds) and the large-
# reservoir fuzz; cfg4 over the bench's default 200 steps on the final kernel; the driver command under round 2's protocol
# (no tail layouts, no unprimed pass before it: ADVICE r4) beside the headline protocol on the same box.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call25; mkdir -p $O
for seed in 61 62 63; do timeout -k 10 330 python3 exp/r05_fuzz_pairs.py $seed 26 > $O/pairs_$seed.txt 2>&1; echo "pairs $seed rc=$?"; tail -1 $O/pairs_$seed.txt | cut -c1-200; done
timeout -k 10 300 python3 exp/r02_fuzz_big.py 32 10 > $O/big_32.txt 2>&1; echo "big 32 rc=$?"; tail -1 $O/big_32.txt | cut -c1-300
export LSM_TRAFFIC_FILE=$GRAFT_REPO_ROOT/profiles/lif_traffic.json
timeout -k 10 300 python3 bench.py --config cfg4 --no-cpu-baseline > $O/cfg4_default.json 2> $O/cfg4_default.err && echo "cfg4 default done"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/driver_headline.json 2> $O/driver_headline.err && echo "driver headline done"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --tail-steps 0 --no-unprimed --no-cpu-baseline > $O/driver_round2_protocol.json 2> $O/driver_round2.err && echo "driver round-2 protocol done"
for f in cfg4_default driver_headline driver_round2_protocol; do python3 -c "
import json
d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); r=d.get('roofline',{})
print('$f', d['value'], d['ms_per_step'], d['steps'], 'kernel_ms', r.get('kernel_ms'), 'unprimed', (d.get('unprimed') or {}).get('value'))"; done
