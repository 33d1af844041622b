#!/bin/bash
# Round 5, call 5: NaN propagation tests (front end), whole parity suites on the build with the build id, phase clocks of the pair kernel.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call5; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_mel.py tests/test_gpu_configs.py tests/test_gpu_ordered.py -m gpu -q --maxfail=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -15 $O/pytest.log | tee -a $O/summary.txt
LSM_HIP_LIB=exp/variants/lib_pair_phases.so timeout -k 10 300 python3 exp/r03_ring_phases.py cfg4 1024 > $O/phases.txt 2>&1; tail -14 $O/phases.txt
