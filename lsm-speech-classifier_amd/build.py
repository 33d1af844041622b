"""Build recipe for liblsm_hip.so (hipcc, gfx950 only, in-tree so the .so travels with gpurun)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
SOURCES = ["lsm_api.hip", "frontend.hip", "reservoir.hip"]
LIB_NAME = "liblsm_hip.so"
# -ffp-contract=off: the kernels must round every float operation exactly like the CPU oracle.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared",
         "-fvisibility=hidden", "-std=c++17"]


def lib_path() -> str:
    return os.path.join(PKG_DIR, LIB_NAME)


def needs_build() -> bool:
    out = lib_path()
    if not os.path.exists(out):
        return True
    deps = [os.path.join(PKG_DIR, "csrc", s) for s in SOURCES + ["lsm_common.h"]]
    deps.append(os.path.join(ROOT, "include", "lsm_hip.h"))
    return any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return lib_path()
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-I", os.path.join(ROOT, "include"), "-o", lib_path()]
    cmd += [os.path.join(PKG_DIR, "csrc", s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib_path()


if __name__ == "__main__":
    print(build(force=True, verbose=True))
