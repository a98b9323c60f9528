"""Scene data on the host: the flattened octree and the .asdf file.

Mirrors `OctData` / `OctData.NativeOctData` of SdfBox/Program.cs:503-667: the
arrays are produced by the native library (load or generate), copied into
managed (numpy) arrays, and the native copy is freed -- the same ownership
hand-off as Logic.MakeData (Logic.cs:87-105).
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import lib, check


class _NativeArrays:
    """The malloc'ed arrays of a tree the library returned; freed (NativeOctData.Free, Logic.cs:99) with the last reference."""

    def __init__(self, raw):
        self.raw = _lib.COctData()
        self.raw.length, self.raw.structs, self.raw.values = raw.length, raw.structs, raw.values

    def __del__(self):
        try:
            lib.sdfhip_octdata_free(ctypes.byref(self.raw))
        except Exception:
            pass


class OctData:
    """`Structs` (N x {parent, children} int32) and `Values` (N x 8 uint8, corner
    k = x + 2y + 4z, not texture-swizzled).  Program.cs:503-511."""

    def __init__(self, structs, values):
        structs = np.ascontiguousarray(structs, dtype=np.int32).reshape(-1, 2)
        values = np.ascontiguousarray(values, dtype=np.uint8).reshape(-1, 8)
        if len(structs) != len(values) or len(structs) == 0:
            raise ValueError("OctData: structs and values must describe the same, non-zero, node count")
        self.Structs = structs
        self.Values = values

    @property
    def Length(self):
        return len(self.Structs)

    @property
    def nbytes(self):
        return 16 * self.Length

    # -- native hand-off ---------------------------------------------------
    @classmethod
    def _from_native(cls, raw):
        # The reference copies the native arrays out element by element and then frees them (NativeOctData.ManagedStructs /
        # ManagedValues + Free, Program.cs:590-611, Logic.cs:97-99).  Here the numpy arrays ARE the native arrays -- 30 ms of
        # copying for a 12 M-node tree -- and sdfhip_octdata_free runs when the last array (or view of one) has gone.
        n = raw.length
        if n == 0:
            lib.sdfhip_octdata_free(ctypes.byref(raw))
            raise ValueError("OctData: the library returned an empty tree")
        owner = _NativeArrays(raw)
        sbuf = (ctypes.c_int32 * (2 * n)).from_address(ctypes.addressof(raw.structs.contents))
        vbuf = (ctypes.c_uint8 * (8 * n)).from_address(ctypes.addressof(raw.values.contents))
        sbuf._owner = vbuf._owner = owner                # the arrays' .base keeps the ctypes buffer alive, and that the owner
        return cls(np.ctypeslib.as_array(sbuf).reshape(n, 2), np.ctypeslib.as_array(vbuf).reshape(n, 8))

    def _as_native(self):
        raw = _lib.COctData()
        raw.length = self.Length
        raw.structs = self.Structs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        raw.values = self.Values.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
        return raw

    # -- .asdf ------------------------------------------------------------
    @classmethod
    def LoadAsdf(cls, path):
        """NativeOctData.LoadAsdf, Program.cs:657-658 -> dllmain.cpp:250-276."""
        raw = _lib.COctData()
        check(lib.sdfhip_asdf_load(os.fsencode(path), ctypes.byref(raw)))
        return cls._from_native(raw)

    def Save(self, path):
        """NativeOctData.Save, Program.cs:660-661 -> dllmain.cpp:278-292."""
        raw = self._as_native()
        check(lib.sdfhip_asdf_save(ctypes.byref(raw), os.fsencode(path)))

    # -- analytic builder (stands in for SdfGen, which needs mesh files) ----
    @classmethod
    def Generate(cls, shape, params, max_depth, nthreads=None):
        if nthreads is None:
            nthreads = min(32, os.cpu_count() or 1)
        p = (ctypes.c_float * len(params))(*[float(x) for x in params])
        raw = _lib.COctData()
        check(lib.sdfhip_generate(int(shape), p, len(params), int(max_depth), int(nthreads),
                                  ctypes.byref(raw)))
        return cls._from_native(raw)

    # -- SdfGen: point cloud -> ASDF on the GPU ---------------------------------------
    @staticmethod
    def LoadPly(path):
        """NativeOctData.LoadPly, Program.cs:654-655 -> ply_reader.cpp: (n, 6) float32 {pos, normal}."""
        raw = _lib.CPoints()
        check(lib.sdfhip_load_ply(os.fsencode(path), ctypes.byref(raw)))
        return OctData._points(raw)

    @staticmethod
    def LoadObj(path):
        """NativeOctData.LoadObj, Program.cs:652-653 -> obj_reader.cpp."""
        raw = _lib.CPoints()
        check(lib.sdfhip_load_obj(os.fsencode(path), ctypes.byref(raw)))
        return OctData._points(raw)

    @staticmethod
    def _points(raw):
        try:
            return np.ctypeslib.as_array(raw.data, shape=(raw.count, 6)).copy() if raw.count else np.zeros((0, 6), np.float32)
        finally:
            lib.sdfhip_points_free(ctypes.byref(raw))

    @classmethod
    def SdfGen(cls, vertices, depth, device=0, want_stats=False):
        """NativeOctData.SdfGen(vertices, Model.MaxDepth), Program.cs:662-663 -> dllmain.cpp:295-319,
        built on the GPU.  vertices: (n, 6) float32 {position, normal}."""
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 6)
        raw = _lib.COctData()
        st = _lib.SdfGenStats()
        check(lib.sdfhip_sdfgen(int(device), v.ctypes.data, len(v), int(depth), ctypes.byref(raw), ctypes.byref(st)))
        od = cls._from_native(raw)
        return (od, st) if want_stats else od

    def validate(self):
        """(depth, consistent) or raises SdfHipError(ERR_BAD_TREE)."""
        depth = ctypes.c_uint32()
        cons = ctypes.c_int()
        check(lib.sdfhip_octdata_validate(self.Structs.ctypes.data, self.Length,
                                          ctypes.byref(depth), ctypes.byref(cons)))
        return depth.value, bool(cons.value)


# Named synthetic scenes.  All deterministic.
def sphere_d4():
    """cfg-1: SDF |p - 0.5| - 0.3, depth 4."""
    return OctData.Generate(_lib.SHAPE_SPHERE, [0.5, 0.5, 0.5, 0.3], 4)


def torus_d6():
    return OctData.Generate(_lib.SHAPE_TORUS, [0.5, 0.5, 0.5, 0.25, 0.09], 6)


def dragon_standin(depth=9, nthreads=None):
    """cfg-2..4: gyroid shell (frequency 12*pi, half-thickness 0.004) clipped to
    the ball r = 0.42 about the cube centre; stands in for the Stanford dragon,
    which does not ship and cannot be fetched."""
    return OctData.Generate(_lib.SHAPE_GYROID, [0.5, 0.5, 0.5, 0.42, 12.0 * np.pi, 0.004], depth,
                            nthreads)


def knot_point_cloud(n=1_000_000, seed=1):
    """A mesh-like workload for the .ply -> SdfGen -> .asdf -> render flow (Program.cs:613-650): `n` surface points
    with normals on the tube of a (2,3) torus knot, (n, 6) float32 {position, normal}.  Deterministic; stands in for
    the point cloud of a scanned mesh (the Stanford dragon's .ply does not ship and cannot be fetched)."""
    rng = np.random.default_rng(seed)
    t = rng.uniform(0, 2 * np.pi, n); a = rng.uniform(0, 2 * np.pi, n)
    p, q, R, r, tube = 2, 3, 0.28, 0.11, 0.035
    c = np.stack([(R + r * np.cos(q * t)) * np.cos(p * t), (R + r * np.cos(q * t)) * np.sin(p * t), r * np.sin(q * t)], 1)
    d = np.stack([-(R + r * np.cos(q * t)) * p * np.sin(p * t) - r * q * np.sin(q * t) * np.cos(p * t),
                  (R + r * np.cos(q * t)) * p * np.cos(p * t) - r * q * np.sin(q * t) * np.sin(p * t),
                  r * q * np.cos(q * t)], 1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    u = np.cross(d, [0, 0, 1.0]); u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = np.cross(d, u)
    nrm = np.cos(a)[:, None] * u + np.sin(a)[:, None] * v
    pos = c + tube * nrm + 0.5
    return np.concatenate([pos, nrm], 1).astype(np.float32)


def write_ply(path, vertices):
    """Binary little-endian .ply with 6 floats per vertex, vertex element first: what LoadPly reads (ply_reader.cpp:35-71)."""
    v = np.ascontiguousarray(vertices, dtype="<f4").reshape(-1, 6)
    with open(path, "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\ncomment sdfbox_amd.write_ply\nelement vertex {len(v)}\n"
                 "property float x\nproperty float y\nproperty float z\n"
                 "property float nx\nproperty float ny\nproperty float nz\nend_header\n").encode())
        f.write(v.tobytes())
