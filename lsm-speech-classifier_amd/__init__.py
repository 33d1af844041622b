"""MI355X-native hot path of the LSM speech pipeline (filterbank -> hysteresis spike encoder ->
LIF reservoir -> spike features) behind the reference's function surface.

Layout: ``csrc/`` HIP kernels + C-ABI (``include/lsm_hip.h``), ``_lib`` ctypes loader,
``frontend`` / ``snn`` host mirrors of the reference interface, ``reservoir`` wiring builder,
``dist`` clip sharding + RCCL feature gather, ``synth`` synthetic inputs.
"""
__version__ = "0.1.0"
