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
def _configure_hw_queues(default: int = 12) -> int:
    """Set GPU_MAX_HW_QUEUES unless the user chose a value; returns the number of hardware queues IN FORCE for
    this process: the variable only counts if it was set before HIP initialised (the runtime's default is 4)."""
    import sys
    import warnings
    torch = sys.modules.get("torch")
    late = torch is not None and torch.cuda.is_initialized()
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        # a value present at import: in force unless HIP came up before it was set, which cannot be known here
        return int(_os.environ["GPU_MAX_HW_QUEUES"])
    if late:
        warnings.warn("HIP was initialised before lsm_speech_classifier_amd was imported: GPU_MAX_HW_QUEUES cannot "
                      "be raised any more, the overlapped pipeline's streams share 4 hardware queues", RuntimeWarning)
        return 4
    _os.environ["GPU_MAX_HW_QUEUES"] = str(int(default))
    return int(default)


EFFECTIVE_HW_QUEUES = _configure_hw_queues()

__version__ = "0.5.0"
