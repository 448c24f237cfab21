"""Builds raytracer-rpf_amd/lib/librpf_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.
hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the kernels rely on the reference's
fp64 operation order for the discrete decisions (see csrc/rpf_kernels.hip)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB = os.path.join(_HERE, "lib", "librpf_hip.so")
SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("rpf_kernels.hip", "rpf_api.hip")]
HEADERS = [os.path.join(_HERE, "csrc", "rpf_internal.h"), os.path.join(_ROOT, "include", "rpf_hip.h")]


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
           "-I" + os.path.join(_ROOT, "include"), "-I" + os.path.join(_HERE, "csrc"), "-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
