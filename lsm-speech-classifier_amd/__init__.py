"""MI355X-native hot path of the LSM speech pipeline (filterbank -> hysteresis spike encoder ->
LIF reservoir -> spike features) behind the reference's function surface.

Layout: ``csrc/`` HIP kernels + C-ABI (``include/lsm_hip.h``), ``_lib`` ctypes loader,
``frontend`` / ``snn`` host mirrors of the reference interface, ``reservoir`` wiring builder,
``dist`` clip sharding + RCCL feature gather, ``pipeline`` the overlapped audio -> features hot path,
``synth`` synthetic inputs.
"""
import os as _os

# The overlapped pipeline (pipeline.HotPath) rotates steps over six HIP streams; the runtime maps streams
# onto 4 hardware queues unless told otherwise, and kernels of streams that share a queue serialise.  The
# variable is read once, when HIP initialises, so it is set here -- the earliest point of any use of the
# package -- unless the user has chosen a value.  (pipeline.configure_hardware_queues documents the numbers.)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

__version__ = "0.2.0"
