#!/bin/bash
# Round 5, call 12: pair kernel with the feature accumulators as spike-train bits in the workspace (FEATG) against the LDS records
# (hooks build, LSM_PAIR_NO_FEATG=1): parity suites, reservoir stage alone, whole cfg4 path, phases.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call12; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_ordered.py tests/test_gpu_fuzz.py tests/test_gpu_round3.py tests/test_gpu_graph.py tests/test_gpu_c_abi.py tests/test_gpu_hotpath.py -m gpu -q --maxfail=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -8 $O/pytest.log | tee -a $O/summary.txt
line() { python3 -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit(0)
d = json.loads(ls[-1]); r = d.get('roofline', {})
print('$1', '->', d['value'], 'clips/s', d['ms_per_step'], 'ms/step; lone', r.get('kernel_ms'))
"; }
H=exp/variants/liblsm_hooks.so
for rep in 1 2; do
  for V in 0 1; do
    LSM_HIP_LIB=$H LSM_PAIR_NO_FEATG=$V timeout -k 10 300 python3 bench.py --config cfg4 --stage reservoir --streams 1 --steps 12 --warmup 3 --no-cpu-baseline --no-unprimed 2>$O/err_r$V.txt | line "cfg4 reservoir no_featg=$V" | tee -a $O/featg.txt
  done
done
for rep in 1 2; do
  for V in 0 1; do
    LSM_HIP_LIB=$H LSM_PAIR_NO_FEATG=$V timeout -k 10 300 python3 bench.py --config cfg4 --steps 24 --warmup 4 --no-cpu-baseline --no-unprimed 2>$O/err_w$V.txt | line "cfg4 whole path no_featg=$V" | tee -a $O/featg.txt
  done
done
LSM_HIP_LIB=exp/variants/lib_pair_phases.so timeout -k 10 300 python3 exp/r03_ring_phases.py cfg4 1024 > $O/phases.txt 2>&1; tail -14 $O/phases.txt
