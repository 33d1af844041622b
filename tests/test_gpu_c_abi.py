"""The drop-in boundary without Python: a plain-C program (tests/c_abi_smoke.c) links liblsm_hip.so
through include/lsm_hip.h, drives it with raw hipMalloc buffers and checks the features against the C
oracle bit for bit."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_consumer(tmp_path, oracle_c):
    pkg = os.path.join(ROOT, "lsm-speech-classifier_amd")
    exe = str(tmp_path / "c_abi_smoke")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", os.path.join(ROOT, "tests", "c_abi_smoke.c"), "-std=c11", "-O1",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(rocm, "include"), "-D__HIP_PLATFORM_AMD__",
           "-L", pkg, "-llsm_hip", "-L", os.path.join(ROOT, "oracle"), "-llsm_oracle",
           "-L", os.path.join(rocm, "lib"), "-lamdhip64",
           f"-Wl,-rpath,{pkg}", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}", f"-Wl,-rpath,{os.path.join(rocm, 'lib')}",
           "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "C ABI OK" in run.stdout, run.stdout + run.stderr
