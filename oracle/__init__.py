"""ctypes access to the CPU oracle (oracle/sdf_oracle.c).

TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing in sdfbox_amd/ imports this package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("sdf_oracle.c", "sdfgen_oracle.c")]
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return LIB_PATH


NATIVE_PATH = os.path.join(_HERE, "liboracle_native.so")


def build_native():
    """The same sources with -O3 -march=native, built on THIS host (bench.py's cpu_baseline leg on the GPU box;
    never shipped from another machine: -march=native code is not portable).  -> path, or None without a compiler."""
    srcs = [os.path.join(_HERE, f) for f in ("sdf_oracle.c", "sdfgen_oracle.c")]
    try:
        subprocess.check_call(["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-std=c11",
                               "-shared", "-o", NATIVE_PATH] + srcs + ["-lm", "-lpthread"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        return None
    return NATIVE_PATH


VARIANTS = {"unfused": "-DO_UNFUSED", "lerp_mathcs": "-DO_LERP_MATHCS", "sampler8": "-DO_SAMPLER8", "rsqrt1ulp": "-DO_RSQRT1ULP",
            # what a D3D11 GPU may really run: 8-bit bilinear weights in the texture unit AND mads that are not fused
            "sampler8_unfused": "-DO_SAMPLER8 -DO_UNFUSED"}


def build_variant(name):
    """A contract variant of sdf_oracle.c (see its header): measurement only, for tests/test_oracle_variants.py."""
    path = os.path.join(_HERE, f"liboracle_{name}.so")
    src = os.path.join(_HERE, "sdf_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-march=x86-64-v2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-std=c11",
                               *VARIANTS[name].split(), "-shared", "-o", path, src, os.path.join(_HERE, "sdfgen_oracle.c"), "-lm", "-lpthread"])
    return path


_lib = None
_native = None
_variants = {}


def lib(native=False):
    global _lib, _native
    if isinstance(native, str):                    # a contract variant
        if native not in _variants:
            _variants[native] = _bind(ctypes.CDLL(build_variant(native)))
        return _variants[native]
    if native:
        if _native is None:
            _native = _bind(ctypes.CDLL(NATIVE_PATH))
        return _native
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = _bind(ctypes.CDLL(LIB_PATH))
    return _lib


def _bind(L):
    vp, u32, c = ctypes.c_void_p, ctypes.c_uint32, ctypes
    L.oracle_render_rows.restype = c.c_int
    L.oracle_render_rows.argtypes = [vp, vp, u32, vp, u32, u32, u32, u32, vp, vp, vp, c.c_int]
    L.oracle_render_rows_pt.restype = c.c_int
    L.oracle_render_rows_pt.argtypes = [vp, vp, u32, vp, u32, u32, u32, u32, u32, u32, u32, c.c_float, vp, vp, c.c_int]
    L.oracle_bench_rows.restype = c.c_int
    L.oracle_bench_rows.argtypes = [vp, vp, u32, vp, u32, u32, u32, u32, vp, vp, c.c_int, c.c_int, c.c_int, c.POINTER(c.c_double), c.POINTER(c.c_int * 4)]
    L.oracle_pixel.restype = None
    L.oracle_pixel.argtypes = [vp, vp, u32, vp, u32, u32, vp, vp]
    L.oracle_distance_at.restype = c.c_float
    L.oracle_distance_at.argtypes = [vp, vp, u32, c.c_float, c.c_float, c.c_float,
                                     c.POINTER(u32), c.POINTER(c.c_float)]
    L.oracle_unorm_table.restype = None
    L.oracle_unorm_table.argtypes = [vp]
    L.oracle_sdfgen.restype = c.c_int
    L.oracle_sdfgen.argtypes = [vp, u32, c.c_int32, c.POINTER(vp), c.POINTER(vp), c.POINTER(vp), c.POINTER(u32),
                                c.POINTER(c.c_float), c.POINTER(c.c_float * 3)]
    L.oracle_sdfgen_free.restype = None
    L.oracle_sdfgen_free.argtypes = [vp]
    L.oracle_display.restype = None
    L.oracle_display.argtypes = [vp, c.c_uint64, c.c_int, vp]
    return L


def _info_buf(info):
    b = bytes(info)
    assert len(b) == 112
    return ctypes.create_string_buffer(b, 112)


def render(structs, values, info, width, height, row0=0, nrows=None, nthreads=1, per_pixel_nodes=False,
           row_step=1, native=False):
    """-> (rgba[nrows, W, 4] f32, counters[4] u64 (nodes, samples, steps, shadow rays)[, nodes per pixel]).
    native: True = the -march=native build (bench.py), a name from VARIANTS = that contract variant."""
    structs = np.ascontiguousarray(structs, dtype=np.int32)
    values = np.ascontiguousarray(values, dtype=np.uint8)
    n = structs.size // 2
    if nrows is None:
        nrows = (height - row0 + row_step - 1) // row_step
    out = np.zeros((nrows, width, 4), dtype=np.float32)
    cnt = np.zeros(4, dtype=np.uint64)
    pix = np.zeros((nrows, width), dtype=np.uint32) if per_pixel_nodes else None
    ib = _info_buf(info)
    rc = lib(native).oracle_render_rows(structs.ctypes.data, values.ctypes.data, n, ctypes.addressof(ib),
                                  width, row0, nrows, row_step, out.ctypes.data, cnt.ctypes.data,
                                  pix.ctypes.data if pix is not None else None, int(nthreads))
    if rc != 0:
        raise MemoryError("oracle_render_rows failed")
    return (out, cnt, pix) if per_pixel_nodes else (out, cnt)


def bench_rows(structs, values, info, width, height, row0=0, nrows=None, row_step=1, nthreads=1, numa_copies=True,
               native=False, store=True, repeat=1):
    """The timed CPU baseline (oracle_bench_rows): the same pixels as render(), by a pool of threads that are pinned, have
    their node's copy of the scene and wait at a barrier before the clock starts; pixels dealt in chunks of 64 from one counter.
    repeat: the sample rendered that many times over (counters count every pass).
    -> (rgba or None, counters[4], seconds between the barriers, {"threads", "numa_nodes", "affinity_cpus", "scene_copies"})."""
    structs = np.ascontiguousarray(structs, dtype=np.int32)
    values = np.ascontiguousarray(values, dtype=np.uint8)
    if nrows is None:
        nrows = (height - row0 + row_step - 1) // row_step
    out = np.zeros((nrows, width, 4), dtype=np.float32) if store else None
    cnt = np.zeros(4, dtype=np.uint64)
    sec = ctypes.c_double()
    topo = (ctypes.c_int * 4)()
    ib = _info_buf(info)
    rc = lib(native).oracle_bench_rows(structs.ctypes.data, values.ctypes.data, structs.size // 2, ctypes.addressof(ib), width, row0, nrows,
                                       row_step, out.ctypes.data if store else None, cnt.ctypes.data, int(nthreads),
                                       1 if numa_copies else 0, int(repeat), ctypes.byref(sec), ctypes.byref(topo))
    if rc != 0:
        raise RuntimeError(f"oracle_bench_rows failed with code {rc}")
    return out, cnt, sec.value, {"threads": topo[0], "numa_nodes": topo[1], "affinity_cpus": topo[2], "scene_copies": topo[3]}


def descent_levels(structs, values, info, width, height, nthreads=8, row_step=1):
    """Analysis (scripts/descent_levels.py): node records loaded by find()'s descents, by the octree
    level of the loaded node -> uint64[16].  Not thread-safe against other renders."""
    hist = np.zeros(16, dtype=np.uint64)
    L = lib()
    L.oracle_level_histogram.argtypes = [ctypes.c_void_p]
    L.oracle_level_histogram.restype = None
    L.oracle_level_histogram(hist.ctypes.data)
    try:
        _, cnt = render(structs, values, info, width, height, nthreads=nthreads, row_step=row_step)
    finally:
        L.oracle_level_histogram(None)
    return hist, cnt


def render_pt(structs, values, info, width, height, spp=16, max_bounces=3, seed=0x5DFB0C5, albedo=0.8,
              row0=0, nrows=None, nthreads=1, row_step=1):
    """Path-traced mode (BASELINE config 5): -> (rgba[nrows, W, 4] f32, counters[4] u64)."""
    structs = np.ascontiguousarray(structs, dtype=np.int32)
    values = np.ascontiguousarray(values, dtype=np.uint8)
    if nrows is None:
        nrows = (height - row0 + row_step - 1) // row_step
    out = np.zeros((nrows, width, 4), dtype=np.float32)
    cnt = np.zeros(4, dtype=np.uint64)
    ib = _info_buf(info)
    rc = lib().oracle_render_rows_pt(structs.ctypes.data, values.ctypes.data, structs.size // 2,
                                     ctypes.addressof(ib), width, row0, nrows, row_step, int(spp),
                                     int(max_bounces), int(seed), float(albedo), out.ctypes.data,
                                     cnt.ctypes.data, int(nthreads))
    if rc != 0:
        raise ValueError("oracle_render_rows_pt failed")
    return out, cnt


def pixel(structs, values, info, x, y):
    structs = np.ascontiguousarray(structs, dtype=np.int32)
    values = np.ascontiguousarray(values, dtype=np.uint8)
    out = np.zeros(4, dtype=np.float32)
    cnt = np.zeros(4, dtype=np.uint64)
    ib = _info_buf(info)
    lib().oracle_pixel(structs.ctypes.data, values.ctypes.data, structs.size // 2,
                       ctypes.addressof(ib), int(x), int(y), out.ctypes.data, cnt.ctypes.data)
    return out, cnt


def distance_at(structs, values, x, y, z):
    structs = np.ascontiguousarray(structs, dtype=np.int32)
    values = np.ascontiguousarray(values, dtype=np.uint8)
    idx = ctypes.c_uint32()
    sc = ctypes.c_float()
    d = lib().oracle_distance_at(structs.ctypes.data, values.ctypes.data, structs.size // 2,
                                 float(x), float(y), float(z), ctypes.byref(idx), ctypes.byref(sc))
    return float(d), idx.value, sc.value


def unorm_table():
    out = np.zeros(256, dtype=np.float32)
    lib().oracle_unorm_table(out.ctypes.data)
    return out


def display(rgba, debug=False):
    """DisplayFrag.hlsl on a float frame (H, W, 4) -> uint8 (H, W, 4), R,G,B,A."""
    rgba = np.ascontiguousarray(rgba, dtype=np.float32)
    out = np.zeros(rgba.shape, dtype=np.uint8)
    lib().oracle_display(rgba.ctypes.data, rgba.size // 4, 1 if debug else 0, out.ctypes.data)
    return out


def sdfgen(verts6, depth):
    """SdfGen (dllmain.cpp:295-319) on a point cloud: verts6 = (n, 6) float32 {position, normal}.
    -> dict(structs (N,2) int32, values (N,8) uint8, float_values (N,8) float32, scale, offset)."""
    v = np.ascontiguousarray(verts6, dtype=np.float32).reshape(-1, 6)
    s, b, f = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    n = ctypes.c_uint32()
    gs = ctypes.c_float()
    go = (ctypes.c_float * 3)()
    rc = lib().oracle_sdfgen(v.ctypes.data, len(v), int(depth), ctypes.byref(s), ctypes.byref(b), ctypes.byref(f),
                             ctypes.byref(n), ctypes.byref(gs), ctypes.byref(go))
    if rc != 0:
        raise RuntimeError(f"oracle_sdfgen failed with code {rc} (2 = empty candidate list: the reference throws)")
    N = n.value
    out = {
        "structs": np.ctypeslib.as_array(ctypes.cast(s, ctypes.POINTER(ctypes.c_int32)), (N, 2)).copy(),
        "values": np.ctypeslib.as_array(ctypes.cast(b, ctypes.POINTER(ctypes.c_uint8)), (N, 8)).copy(),
        "float_values": np.ctypeslib.as_array(ctypes.cast(f, ctypes.POINTER(ctypes.c_float)), (N, 8)).copy(),
        "scale": gs.value, "offset": tuple(go),
    }
    for p in (s, b, f):
        lib().oracle_sdfgen_free(p)
    return out
