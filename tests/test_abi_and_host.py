"""CPU tests (-m "not gpu") of the boundary: the C-ABI library loads without a GPU and exports every symbol
include/rpf_hip.h declares, struct layouts agree between the header and the ctypes mirror, argument checks
that need no device, the generator and layout helpers."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "rpf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rpf_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(hipmod):
    L = hipmod.load()
    names = header_functions()
    assert len(names) >= 13
    for n in names:
        assert hasattr(L, n), "librpf_hip.so does not export %s" % n
    assert sorted(hipmod.EXPORTS) == names
    assert b"gfx950" in L.rpf_version()
    assert hipmod.status_string(3) == "RPF_E_NONFINITE"


def test_struct_layouts_match_header(hipmod):
    # rpf_desc: 5 + 1 + 8 + 3 int32 = 17 int32 -> 68 bytes, padded to 72, + 2 doubles = 88, + 4 int32 (layout) = 104
    assert C.sizeof(hipmod.Desc) == 104 and hipmod.Desc.eps.offset == 72 and hipmod.Desc.n_random.offset == 88
    assert C.sizeof(hipmod.Debug) == 9 * C.sizeof(C.c_void_p)
    assert C.sizeof(hipmod.Counters) == 3 * 8 + 2 * 4 + 5 * 4 + 3 * 4 and hipmod.Counters.filter_kernel_ms.offset == 32
    assert hipmod.Counters.options_active.offset == 56 and hipmod.Counters.redo_pixels.offset == 60


def test_no_device_means_loud_failure_not_fallback(hipmod):
    """without a GPU the product path refuses to run: there is no CPU fallback behind the ABI"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hipmod.RpfError) as e:
        hipmod.Context(0)
    assert e.value.status == hipmod.E_NODEVICE


def test_lds_requirement_query(hipmod):
    assert 0 < hipmod.lds_bytes_required(8, 7) <= 64 * 1024
    assert hipmod.lds_bytes_required(8, 7) < hipmod.lds_bytes_required(32, 7) <= 160 * 1024
    assert hipmod.lds_bytes_required(8, 55) == -1  # 24200-sample neighbourhoods: unsupported by this kernel
    # occupancy of the one-wave kernels is an LDS budget (160 KiB per CU): twelve workgroups of the 8-spp kernel, eight of the
    # 16-spp one (its time goes as 1 / resident workgroups: 221 -> 180 ms from six to eight)
    assert hipmod.lds_bytes_required(8, 7) <= 160 * 1024 // 12
    assert hipmod.lds_bytes_required(16, 7) <= 160 * 1024 // 8


def test_product_package_never_imports_the_oracle():
    """parity claims are void if the product path can reach the oracle: grep the package for it"""
    pkg = os.path.join(ROOT, "raytracer-rpf_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "pyoracle" not in txt and "rpf_oracle" not in txt and "libref_mi" not in txt, f


def test_generator_is_deterministic_slab_consistent_and_fp32_exact():
    from raytracer_rpf_amd import feature_buffer as fb
    a = fb.synth_planes(16, 12, 4, seed=3)
    b = fb.synth_planes(16, 12, 4, seed=3)
    assert a.dtype == np.float32 and a.shape == (19, 12, 16, 4) and np.array_equal(a, b)
    assert not np.array_equal(a, fb.synth_planes(16, 12, 4, seed=4))
    # a rank generating rows [5,9) of the same image gets exactly those rows
    s = fb.synth_planes(16, 4, 4, seed=3, row0=5)
    assert np.array_equal(s, a[:, 5:9])
    # pFilm lies inside its pixel, pLens in [0,1)
    assert (np.floor(a[0]) == np.arange(16)[None, :, None]).all() and (np.floor(a[1]) == np.arange(12)[:, None, None]).all()
    assert (a[5:7] >= 0).all() and (a[5:7] < 1).all()
    # torch backend (CPU device) produces the identical buffer
    import torch
    t = fb.synth_planes(16, 12, 4, seed=3, xp=fb.torch_backend(torch.device("cpu"))).numpy()
    assert np.array_equal(t, a)


def test_aos_soa_round_trip():
    """SamplingFilm order samples[x][y][s][19] doubles (sample_film.cpp:32-42) <-> planes [19][y][x][s] fp32"""
    from raytracer_rpf_amd import feature_buffer as fb
    planes = fb.synth_planes(7, 5, 3, seed=1)
    aos = fb.planes_to_aos(planes)
    assert aos.shape == (7, 5, 3, 19) and aos.dtype == np.float64
    assert aos[2, 4, 1, 10] == planes[10, 4, 2, 1]
    assert np.array_equal(fb.aos_to_planes(aos), planes)


def test_rpfb_wire_format_round_trip(tmp_path):
    from raytracer_rpf_amd import feature_buffer as fb
    planes = fb.synth_planes(9, 6, 4, seed=2)
    rw = np.random.default_rng(0).random((6, 9, 4)).astype(np.float32)
    path = str(tmp_path / "buf.rpfb")
    fb.save_rpfb(path, planes, rw)
    assert os.path.getsize(path) == 64 + planes.nbytes + rw.nbytes
    p2, r2 = fb.load_rpfb(path)
    assert np.array_equal(p2, planes) and np.array_equal(r2, rw)
    p3, r3 = fb.load_rpfb(path, mmap=True)
    assert np.array_equal(np.asarray(p3), planes) and np.array_equal(np.asarray(r3), rw)
    fb.save_rpfb(path, planes)
    p4, r4 = fb.load_rpfb(path)
    assert r4 is None and np.array_equal(p4, planes)
    with open(path, "r+b") as f:
        f.write(b"XXXX")
    with pytest.raises(ValueError):
        fb.load_rpfb(path)


def test_oracle_feature_images_follow_visualizeSF(oracle):
    """hand check of rpf.cpp:71-85 + vis.cpp:34-51 on a 2x1 image"""
    planes = np.zeros((19, 1, 2, 2), np.float32)
    planes[7, 0, 0] = [1.0, 3.0]    # n0.x pixel 0: mean 2
    planes[7, 0, 1] = [4.0, 4.0]    # n0.x pixel 1: mean 4 -> max
    planes[8, 0, :] = -1.0          # n0.y negative everywhere: max stays 0 -> 0
    planes[0, 0, 0] = [0.25, 0.75]  # pFilm.x
    planes[0, 0, 1] = [1.25, 1.75]
    img = oracle.feature_images(planes, oracle.make_desc(2, 1, 2))
    assert img.shape == (6, 1, 2, 3)
    assert list(img[0, 0, :, 0]) == [0.5, 1.0] and (img[0, 0, :, 1] == 0).all()
    assert list(img[4, 0, :, 0]) == [0.5 / 1.5, 1.0] and (img[4, 0, :, 2] == 0).all()


def test_optional_integrator_parameters(hipmod):
    """the "rpf" integrator takes only the path tracer's keys (rpf.cpp:946-963) and hard-codes the box
    list {7} (rpf.cpp:767); the host mirror accepts optional "integer boxsizes" / "string backend" with those defaults"""
    lib = C.CDLL(os.path.join(os.path.dirname(hipmod.LIB_PATH), "librpf_host.so"))
    lib.rpf_host_parse_params.restype = C.c_int32

    def parse(boxes, backend):
        arr = (C.c_int32 * max(len(boxes), 1))(*boxes) if boxes is not None else None
        out, n, be = (C.c_int32 * 8)(), C.c_int32(0), C.c_int32(-1)
        err = C.create_string_buffer(256)
        st = lib.rpf_host_parse_params(arr, len(boxes) if boxes is not None else 0,
                                       backend.encode() if backend is not None else None, out, C.byref(n), C.byref(be), err, 256)
        return st, list(out[:n.value]), be.value, err.value.decode()

    assert parse(None, None) == (0, [7], 0, "")                       # a scene file without the keys: the reference's constants
    assert parse([55, 35, 17, 7], "hip") == (0, [55, 35, 17, 7], 0, "")  # the list commented out at rpf.cpp:767
    assert parse([7, 7, 5, 5], "reference")[:3] == (0, [7, 7, 5, 5], 1)
    st, _, _, msg = parse([7, 6], "hip")
    assert st == -1 and "6" in msg and "odd" in msg
    st, _, _, msg = parse([3] * 9, "hip")
    assert st == -1 and "at most 8" in msg
    st, _, _, msg = parse([7], "cuda")
    assert st == -1 and "cuda" in msg


def test_inline_asm_declares_the_status_registers_it_writes():
    """An asm statement whose SALU instructions write SCC (or that writes VCC / EXEC by name) must say so in its clobber list:
    without it hipcc may keep a compare alive across the statement -- round 3 traced wrong histogram sums in some kernel
    instantiations to exactly that (lds_store_u64_if: s_and_saveexec_b64 writes SCC)."""
    import glob
    import re
    csrc = os.path.join(ROOT, "raytracer-rpf_amd", "csrc")
    scc_writers = re.compile(r"\bs_(and|or|xor|andn2|orn2|nand|nor|xnor|not|add|sub|addc|subb|min|max|mul_hi|lshl|lshr|ashr|bfe|bfm|"
                             r"abs|cmp|cmpk|bitcmp|cselect|wqm|quadmask|bcnt|ff|flbit|sext|absdiff)\w*\b")
    bad = []
    for path in sorted(glob.glob(os.path.join(csrc, "*"))):
        text = open(path, errors="replace").read()
        for m in re.finditer(r"\basm\s*(volatile)?\s*\(", text):
            depth, i = 1, m.end()
            while depth and i < len(text):
                depth += {"(": 1, ")": -1}.get(text[i], 0)
                i += 1
            stmt = text[m.start():i]
            strings = " ".join(re.findall(r'"((?:[^"\\]|\\.)*)"', stmt))
            writes_scc = bool(scc_writers.search(strings)) or "saveexec" in strings
            if writes_scc and '"scc"' not in stmt:
                bad.append("%s: %s" % (os.path.basename(path), stmt[:90].replace("\n", " ")))
    assert not bad, bad
