"""Build recipe for liblsm_hip.so (hipcc, gfx950 only, in-tree so the .so travels with gpurun)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
SOURCES = ["lsm_api.hip", "frontend.hip", "mel.hip", "reservoir.hip", "lif_variant_00.hip", "lif_variant_01.hip",
           "lif_variant_10.hip", "lif_variant_11.hip", "lif_dense_0.hip", "lif_dense_1.hip", "lif_dense_2.hip", "lif_dense_3.hip",
           "lif_ring_1.hip", "lif_ring_2.hip", "lif_ring_3.hip", "lif_ring_4.hip"]
HEADERS = ["lsm_common.h", "lif_kernel.h", "lif_dense.h", "lif_ring.h", "spikes_body.h"]
LIB_NAME = "liblsm_hip.so"
# -ffp-contract=off: the kernels must round every float operation exactly like the CPU oracle.
CFLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
          "-std=c++17"]


def lib_path() -> str:
    # LSM_HIP_LIB: load another build of the same library (diagnostic/ablation builds)
    return os.environ.get("LSM_HIP_LIB") or os.path.join(PKG_DIR, LIB_NAME)


def needs_build(out: str | None = None) -> bool:
    out = out or os.path.join(PKG_DIR, LIB_NAME)
    if not os.path.exists(out):
        return True
    deps = [os.path.join(PKG_DIR, "csrc", s) for s in SOURCES + HEADERS]
    deps.append(os.path.join(ROOT, "include", "lsm_hip.h"))
    return any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build(force: bool = False, verbose: bool = False, defines=(), out: str | None = None) -> str:
    """Compile every translation unit (in parallel) and link liblsm_hip.so in-tree."""
    out = out or os.path.join(PKG_DIR, LIB_NAME)
    if not force and not needs_build(out):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = os.path.join(PKG_DIR, "build", "obj" + ("_" + "_".join(defines) if defines else ""))
    os.makedirs(objdir, exist_ok=True)
    base = [hipcc] + CFLAGS + ["-I", os.path.join(ROOT, "include")] + [f"-D{d}" for d in defines]

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = base + ["-c", os.path.join(PKG_DIR, "csrc", src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return out


if __name__ == "__main__":
    import sys
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a for a in sys.argv[1:] if not a.startswith("-D")]
    print(build(force=True, verbose=True, defines=defs, out=outs[0] if outs else None))
