#!/bin/bash
# Round 4, last call: the whole GPU suite and the bench lines of cfg4 / cfg5 / the driver's command on the final code.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_final; mkdir -p $O
python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.log | tee -a $O/summary.txt
python3 __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
python3 bench.py --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline > $O/r04_cfg4.json 2> $O/cfg4.err; echo "cfg4 rc=$?" >> $O/summary.txt
python3 bench.py --config cfg5 --steps 4 --warmup 1 --no-cpu-baseline > $O/r04_cfg5.json 2> $O/cfg5.err; echo "cfg5 rc=$?" >> $O/summary.txt
python3 bench.py --config cfg1 --steps 60 --warmup 8 > $O/r04_cfg1.json 2> $O/cfg1.err; echo "cfg1 rc=$?" >> $O/summary.txt
for i in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_$i.json 2> $O/driver_$i.err; echo "driver $i rc=$?" >> $O/summary.txt; done
python3 bench.py > $O/default.json 2> $O/default.err; echo "default rc=$?" >> $O/summary.txt
for f in r04_cfg4 r04_cfg5 r04_cfg1 driver_1 driver_2 default; do python3 -c "
import json
p='$O/$f.json'
d=json.loads([l for l in open(p) if l.startswith('{')][-1]); json.dump(d, open(p,'w'), indent=1)
r=d.get('roofline',{})
print('$f', d['value'], d['ms_per_step'], 'unprimed', (d.get('unprimed') or {}).get('value'), 'frac', r.get('frac'), 'kernel_ms', r.get('kernel_ms'), 'pipe_frac', r.get('pipeline_frac'), 'fe pipe', (r.get('dominant_kernel_by_time') or {}).get('pipeline_frac'))" | tee -a $O/summary.txt; done
