#!/bin/bash
# Phase clocks of the ring-row kernel (diagnostic build liblsm_hip_phases.so = -DLSM_RING_PHASES=1), cfg4 at its batch and cfg5 at 512 clips.
export LSM_HIP_LIB=/root/repo/lsm-speech-classifier_amd/liblsm_hip_phases.so
OUT=gpurun_out/r03_ring_phases.txt
rm -f $OUT
python3 exp/r03_ring_phases.py cfg4 1024 2>&1 | tee -a $OUT

python3 exp/r03_ring_phases.py cfg5 512 2>&1 | tee -a $OUT
