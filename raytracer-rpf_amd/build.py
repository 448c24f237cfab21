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
HEADERS = [os.path.join(_HERE, "csrc", "rpf_internal.h"), os.path.join(_HERE, "csrc", "rpf_xlane.h"),
           os.path.join(_ROOT, "include", "rpf_hip.h")]


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


HOST_LIB = os.path.join(_HERE, "lib", "librpf_host.so")
HOST_SOURCES = [os.path.join(_HERE, "host", "rpf_host.cpp")]
HOST_HEADERS = [os.path.join(_HERE, "host", "rpf_host.h"), os.path.join(_ROOT, "include", "rpf_hip.h")]


def build_host(force=False, verbose=False):
    """host-side C++ mirror of the reference interface (g++, no HIP), linked against librpf_hip.so"""
    build(force=False)
    stale = (not os.path.exists(HOST_LIB)) or any(
        os.path.getmtime(f) > os.path.getmtime(HOST_LIB) for f in HOST_SOURCES + HOST_HEADERS + [LIB])
    if not force and not stale:
        return HOST_LIB
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-I" + os.path.join(_ROOT, "include"),
           "-I" + os.path.join(_HERE, "host"), "-o", HOST_LIB] + HOST_SOURCES + [
           "-L" + os.path.dirname(LIB), "-lrpf_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HOST_LIB


if __name__ == "__main__":
    print(build_host(force=True, verbose=True))
    print(build(force=True, verbose=True))
