"""ctypes binding of liblsm_hip.so (include/lsm_hip.h).  There is no CPU fallback: if the
library is missing or a call fails, this module raises."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

_lib = None
ABI_VERSION = _build.version_number()      # lsm_version() of a library built from this tree (= __version__, one literal)

c_void = C.c_void_p
c_int = C.c_int
c_float = C.c_float

_SIGS = {
    "lsm_version": (c_int, []),
    "lsm_build_id": (C.c_char_p, []),
    "lsm_last_error": (C.c_char_p, []),
    "lsm_device_count": (c_int, []),
    "lsm_gammatone_spec_f64": (c_int, [c_void, c_int, c_int, c_void, c_int, c_int, c_int, c_int,
                                       c_void, c_void, c_int, c_void]),
    "lsm_gammatone_spikes_workspace": (C.c_long, [c_int, c_int, c_int]),
    "lsm_gammatone_spikes_f64": (c_int, [c_void, c_int, c_int, c_void, c_int, c_int, c_int, c_int, c_int,
                                         c_void, c_void, c_int, c_int, c_void, c_void, C.c_long, c_int, c_int, c_void]),
    "lsm_spec_to_spikes_f64": (c_int, [c_void, c_int, c_int, c_int, c_int, c_int, c_void, c_void,
                                       c_int, c_int, c_void, c_void, c_void]),
    "lsm_spec_to_spikes_f32": (c_int, [c_void, c_int, c_int, c_int, c_int, c_int, c_void, c_void,
                                       c_int, c_int, c_void, c_void, c_void]),
    "lsm_mel_power_f32": (c_int, [c_void, c_int, c_int, c_int, c_int, c_int, c_void, c_void, c_void,
                                  c_void, c_void, c_int, c_void, c_void]),
    "lsm_power_to_db_f32": (c_int, [c_void, c_int, c_int, c_float, c_float, c_void, c_void]),
    "lsm_mel_spikes_workspace": (C.c_long, [c_int, c_int, c_int]),
    "lsm_mel_spikes_f32": (c_int, [c_void, c_int, c_int, c_int, c_int, c_int, c_void, c_void, c_void, c_void, c_void,
                                   c_int, c_float, c_float, c_int, c_void, c_void, c_int, c_int, c_void, c_void, C.c_long,
                                   c_void]),
    "lsm_encode_hysteresis_f64": (c_int, [c_void, c_int, c_int, c_void, c_void, c_int, c_void, c_void]),
    "lsm_encode_hysteresis_f32": (c_int, [c_void, c_int, c_int, c_void, c_void, c_int, c_void, c_void]),
    "lsm_raster_pack_bits": (c_int, [c_void, C.c_long, c_int, c_void, c_void]),
    "lsm_raster_unpack_bits": (c_int, [c_void, C.c_long, c_int, c_void, c_void]),
    "lsm_reservoir_create": (c_int, [C.POINTER(c_void), c_int, c_int, c_void, c_void, c_void, c_void,
                                     c_void, c_int, c_float, c_void, c_int, c_float, c_int, c_int]),
    "lsm_reservoir_destroy": (c_int, [c_void]),
    "lsm_reservoir_set_kernel": (c_int, [c_void, c_int]),
    "lsm_reservoir_kernel_in_use": (c_int, [c_void]),
    "lsm_reservoir_run": (c_int, [c_void, c_void, c_int, c_int, c_void, c_int, c_void, c_void,
                                  c_void, c_void, c_int, c_void]),
    "lsm_reservoir_order_workspace": (C.c_long, [c_int]),
    "lsm_reservoir_run_ordered": (c_int, [c_void, c_void, c_int, c_int, c_void, c_int, c_void, c_void,
                                          c_void, c_void, c_int, c_void, C.c_long, c_void]),
    "lsm_reservoir_layout": (c_int, [c_void, c_int, c_int, c_int, C.POINTER(c_int),
                                     C.POINTER(c_int), C.POINTER(c_int)]),
    "lsm_reservoir_plan": (c_int, [c_void, c_int, c_int, c_int, C.POINTER(c_int), C.POINTER(c_int),
                                   C.POINTER(c_int), C.POINTER(c_int), C.POINTER(C.c_long)]),
    "lsm_reservoir_row_request_bytes": (c_int, [c_void, c_int, c_int, c_int, C.POINTER(C.c_double)]),
    "lsm_reservoir_input_mode": (c_int, [c_void, c_int, c_int, c_int]),
    "lsm_debug_pair_layout": (c_int, [c_int, c_void, c_void, c_void, c_int, C.c_ulonglong, C.c_ulonglong,
                                      C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(c_int), c_void, c_void, c_void]),
    "lsm_debug_lif_stamps": (c_int, [c_void, c_int]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)


class LsmHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises LsmHipError when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64 (same SONAME as /opt/rocm's).  Import it first so that this
    # library binds to the runtime torch already initialised: two HIP runtimes in one process
    # leave the second one without devices.
    import torch  # noqa: F401
    path = _build.lib_path()
    own = "LSM_HIP_LIB" not in os.environ        # a diagnostic build named by the user is loaded as it is
    # A library that is absent, of another version or built from other sources than this tree's (build.source_id, linked
    # into the binary) is rebuilt ONCE -- hipcc runs as a child process, nothing here touches the GPU -- and refused if
    # that does not help: never a silent stale binary, never a CPU fallback.
    if own and _build.needs_build(path) and os.environ.get("LSM_NO_AUTO_BUILD") != "1":
        try:
            import fcntl
            os.makedirs(os.path.join(_build.PKG_DIR, "build"), exist_ok=True)
            with open(os.path.join(_build.PKG_DIR, "build", ".lock"), "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)         # the ranks of a launcher find the same stale library together
                _build.build()                           # (a no-op for every rank but the first)
        except Exception as e:                   # no hipcc (a box that only received the binary): judged below
            import warnings
            warnings.warn(f"rebuilding {path} failed: {e}", RuntimeWarning)
    if not os.path.exists(path):
        raise LsmHipError(
            f"{path} not found: the HIP extension is not built. Run `python -c \"import "
            f"__graft_entry__ as g; g.build()\"` (needs hipcc). There is no CPU fallback.")
    lib = C.CDLL(path)
    try:
        lib.lsm_version.restype = c_int
        have = int(lib.lsm_version())
    except AttributeError:
        have = -1
    if have != ABI_VERSION:              # a stale .so: say so instead of failing on the first new symbol
        raise LsmHipError(
            f"{path} reports ABI version {have}, this package needs {ABI_VERSION}: rebuild the extension "
            f"(`python -c \"import __graft_entry__ as g; g.build()\"`).")
    if own and _build.built_id(path) != _build.source_id():
        raise LsmHipError(
            f"{path} was built from other sources than this tree's (build id {_build.built_id(path)}, sources "
            f"{_build.source_id()}): rebuild the extension (`python -c \"import __graft_entry__ as g; g.build()\"`).")
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().lsm_last_error().decode("utf-8", "replace")
        raise LsmHipError(f"{what or 'liblsm_hip call'} failed ({rc}): {msg}")


def require_gpu() -> int:
    """Number of HIP devices; raises when there is none (the product path never runs on CPU)."""
    n = load().lsm_device_count()
    if n <= 0:
        msg = load().lsm_last_error().decode("utf-8", "replace")
        raise LsmHipError(f"no HIP device visible ({n}): {msg}")
    return n
