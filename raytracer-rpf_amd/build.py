"""Builds raytracer-rpf_amd/lib/librpf_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.
hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the kernels rely on the reference's
fp64 operation order for the discrete decisions (see csrc/rpf_kernels.hip)."""
import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB = os.path.join(_HERE, "lib", "librpf_hip.so")
# the fused per-pixel kernels compile as one translation unit per sample layout and size-class part (rpf_filter_impl.inc
# with RPF_IMPL_PART = 1 / 2 / 3), in parallel: the single kernel TU of rounds 1-2 took 3.2 minutes
KERNEL_TUS = ["rpf_impl_%s_%s.hip" % (lay, part) for lay in ("d19", "d27") for part in ("small", "mid", "large")] + ["rpf_kernels.hip"]
SOURCES = [os.path.join(_HERE, "csrc", f) for f in KERNEL_TUS + ["rpf_api.hip"]]
HEADERS = [os.path.join(_HERE, "csrc", "rpf_internal.h"), os.path.join(_HERE, "csrc", "rpf_xlane.h"),
           os.path.join(_HERE, "csrc", "rpf_filter_impl.inc"), os.path.join(_HERE, "csrc", "rpf_packed_impl.inc"),
           os.path.join(_HERE, "csrc", "rpf_device_common.h"), os.path.join(_HERE, "csrc", "rpf_reflog.h"),
           os.path.join(_ROOT, "include", "rpf_hip.h")]


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


OBJ_DIR = os.path.join(_ROOT, "build", "obj")


def _stale(target, deps):
    return (not os.path.exists(target)) or any(os.path.getmtime(f) > os.path.getmtime(target) for f in deps)


SCAN_OK = os.path.join(OBJ_DIR, "spill_scan.ok")  # written by build() after scripts/check_spills.py passed


def is_stale():
    return _stale(LIB, SOURCES + HEADERS) or not os.path.exists(SCAN_OK)


def build(force=False, verbose=False, defs=(), tag=None):
    """one object per translation unit, compiled in parallel, then the link.
    defs / tag: an experiment variant -- extra -D flags, objects under build/obj_<tag>, library lib/librpf_hip_<tag>.so
    (measured side by side with the shipped library through hip.py's RPF_HIP_LIB; never what tests or bench.py load by default)"""
    if tag:
        return _build_variant(list(defs), tag, verbose)
    if not force and not is_stale():
        return LIB
    return _build(LIB, OBJ_DIR, [], True, force, verbose)


def _build_variant(defs, tag, verbose):
    return _build(os.path.join(_HERE, "lib", "librpf_hip_%s.so" % tag), os.path.join(_ROOT, "build", "obj_" + tag), defs, False, False, verbose)


def _build(LIB, OBJ_DIR, defs, shipped, force, verbose):
    SCAN_OK = os.path.join(OBJ_DIR, "spill_scan.ok")
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
             "-I" + os.path.join(_ROOT, "include"), "-I" + os.path.join(_HERE, "csrc")] + list(defs)
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + HEADERS):
            # -save-temps=obj keeps the device assembly next to the object: scripts/check_spills.py reads it below
            cmd = [hipcc_path()] + flags + ["-save-temps=obj", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    import glob
    asm = sorted(f for f in glob.glob(os.path.join(OBJ_DIR, "*gfx950*.s")) if "rpf_api" not in os.path.basename(f))
    if asm:
        # the scan's verdict is kept next to the objects (is_stale() wants it) and, when every kernel TU was compiled in this
        # invocation, the per-kernel resource usage of the build goes to profiles/ (tracked: the figures of what ships)
        full = shipped and len(asm) == len(KERNEL_TUS)
        report = os.path.join(_ROOT, "profiles", "r03_resource_usage.txt") if full else os.path.join(OBJ_DIR, "resource_usage_partial.txt")
        chk = subprocess.run([sys.executable, os.path.join(_ROOT, "scripts", "check_spills.py"), "--report", report] + asm,
                             stdout=subprocess.PIPE, text=True)
        if chk.returncode != 0:
            if os.path.exists(SCAN_OK):
                os.remove(SCAN_OK)
            for a in asm:  # the objects of a refused build must not be linked by a later invocation
                o = os.path.join(OBJ_DIR, os.path.basename(a).split("-hip-amdgcn")[0] + ".hip.o")
                if os.path.exists(o):
                    os.remove(o)
            raise RuntimeError("miscompiled spill placement in a kernel TU (see scripts/check_spills.py):\n" + chk.stdout)
        with open(SCAN_OK, "w") as f:
            f.write("check_spills.py: no spill code in front of an EXEC restore\n" + chk.stdout)
        if verbose:
            print(chk.stdout)
    for f in glob.glob(os.path.join(OBJ_DIR, "*")):  # the -save-temps intermediates (~100 MB) have served their purpose
        if f not in objs and f != SCAN_OK and not (f.endswith(".txt") and "resolution" not in f):
            os.remove(f)
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
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
    force = "--force" in sys.argv
    if "--variant" in sys.argv:  # python build.py --variant <tag> -DNAME=VALUE ...
        print(build(verbose=True, defs=[a for a in sys.argv[1:] if a.startswith("-D")], tag=sys.argv[sys.argv.index("--variant") + 1]))
        sys.exit(0)
    print(build(force=force, verbose=True))
    print(build_host(force=force, verbose=True))
