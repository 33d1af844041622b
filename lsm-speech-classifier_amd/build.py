"""Build recipe for liblsm_hip.so (hipcc, gfx950 only, in-tree so the .so travels with gpurun)."""
from __future__ import annotations

import hashlib
import os
import re
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
SOURCES = ["lsm_api.hip", "frontend.hip", "mel.hip", "reservoir.hip", "lif_variant_00.hip", "lif_variant_01.hip",
           "lif_variant_10.hip", "lif_variant_11.hip", "lif_dense_0.hip", "lif_dense_1.hip", "lif_dense_2.hip", "lif_dense_3.hip",
           "lif_ring_1.hip", "lif_ring_2.hip", "lif_ring_3.hip", "lif_ring_4.hip",
           "lif_pair_1.hip", "lif_pair_2.hip", "lif_pair_3.hip", "lif_pair_4.hip"]
HEADERS = ["lsm_common.h", "lif_kernel.h", "lif_dense.h", "lif_ring.h", "lif_pair.h", "spikes_body.h"]
LIB_NAME = "liblsm_hip.so"
# -ffp-contract=off: the kernels must round every float operation exactly like the CPU oracle.
CFLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
          "-std=c++17"]


def lib_path() -> str:
    # LSM_HIP_LIB: load another build of the same library (diagnostic/ablation builds)
    return os.environ.get("LSM_HIP_LIB") or os.path.join(PKG_DIR, LIB_NAME)


def package_version() -> str:
    """__version__ of the package, read from its text (importing the package here would import torch)."""
    with open(os.path.join(PKG_DIR, "__init__.py")) as f:
        return re.search(r'^__version__\s*=\s*"([0-9]+)\.([0-9]+)\.([0-9]+)"', f.read(), re.M).group(0).split('"')[1]


def version_number() -> int:
    """major*10000 + minor*100 + patch: what lsm_version() of a library built from this tree returns."""
    a, b, c = (int(x) for x in package_version().split("."))
    return a * 10000 + b * 100 + c


def source_id() -> str:
    """Identity of what a build of this tree compiles: every source and header under csrc/, include/lsm_hip.h, the
    compiler flags and the version, hashed.  build() links it into the library (lsm_build_id()); _lib.load() refuses a
    library that carries another one, and needs_build() compares it instead of trusting file times -- a stale binary
    that merely looks newer than the sources cannot pass for a build of them."""
    h = hashlib.sha256()
    files = sorted(os.path.join(PKG_DIR, "csrc", f) for f in SOURCES + HEADERS) + [os.path.join(ROOT, "include", "lsm_hip.h")]
    for path in files:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(" ".join(CFLAGS).encode())
    h.update(str(version_number()).encode())
    return h.hexdigest()[:24]


_ID_MARK = b"LSM_BUILD_ID="


def built_id(path: str | None = None) -> str | None:
    """The source_id a library was built from, read from the file's bytes (no dlopen); None if it has none."""
    path = path or os.path.join(PKG_DIR, LIB_NAME)
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    m = re.search(re.escape(_ID_MARK) + rb"([0-9a-f]{24})", blob)
    return m.group(1).decode() if m else None


def needs_build(out: str | None = None) -> bool:
    out = out or os.path.join(PKG_DIR, LIB_NAME)
    return built_id(out) != source_id()


def build(force: bool = False, verbose: bool = False, defines=(), out: str | None = None) -> str:
    """Compile every translation unit (in parallel) and link liblsm_hip.so in-tree."""
    out = out or os.path.join(PKG_DIR, LIB_NAME)
    if not force and not needs_build(out):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = os.path.join(PKG_DIR, "build", "obj" + ("_" + "_".join(defines) if defines else ""))
    os.makedirs(objdir, exist_ok=True)
    ident = [f'-DLSM_BUILD_ID_STRING="{_ID_MARK.decode()}{source_id()}"', f"-DLSM_VERSION_NUMBER={version_number()}"]
    base = [hipcc] + CFLAGS + ["-I", os.path.join(ROOT, "include")] + [f"-D{d}" for d in defines]

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = base + (ident if src == "lsm_api.hip" else []) + ["-c", os.path.join(PKG_DIR, "csrc", src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out + ".tmp"] + objs
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    os.replace(out + ".tmp", out)           # a reader never sees a half-written library
    return out


if __name__ == "__main__":
    import sys
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a for a in sys.argv[1:] if not a.startswith("-D")]
    print(build(force=True, verbose=True, defines=defs, out=outs[0] if outs else None))
