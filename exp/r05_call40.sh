#!/bin/bash
# Round 5, call 40: the front-end tests once more on the final library (mel tables' alignment check added after call 39).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_call40
timeout -k 10 600 python3 -m pytest tests/test_gpu_mel.py tests/test_gpu_c_abi.py tests/test_gpu_graph.py tests/test_gpu_parity.py -q > gpurun_out/r05_call40/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_call40/pytest.log
timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_call40/bench.json 2> gpurun_out/r05_call40/bench.err && python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r05_call40/bench.json') if l.startswith('{')][-1]); print('driver', d['value'], d['ms_per_step'])"
