"""ctypes binding of libsdfhip.so (include/sdfhip.h).

The library is the product; there is no Python or CPU fallback.  If it has not
been built (``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C sdfbox_amd/csrc``) importing this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDFHIP_LIB: another build of the library -- sdfbox_amd/libsdfhip_lab.so is the experiments flavour (include/sdfhip_experimental.h:
# the A/B kernel forms, superseded gather formats and test hooks; sdfbox_amd.lab.load() imports this package a second time against it)
LIB_PATH = os.path.abspath(os.environ.get("SDFHIP_LIB") or os.path.join(_HERE, "libsdfhip.so"))
LAB_LIB_PATH = os.path.join(_HERE, "libsdfhip_lab.so")

# PyTorch bundles its own libamdhip64.so; load it first so that libsdfhip.so
# binds to the same HIP runtime instance torch uses (one runtime per process:
# device pointers and streams are then interchangeable).
try:  # pragma: no cover - depends on the environment
    import torch  # noqa: F401
except Exception:  # torch is plumbing, not a requirement of the C ABI
    torch = None

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
        "(hipcc --offload-arch=gfx950). sdfbox_amd has no CPU fallback.")

lib = ctypes.CDLL(LIB_PATH)

OK, ERR_ARG, ERR_IO, ERR_BAD_TREE, ERR_DEVICE, ERR_NOMEM = range(6)

KERNEL_AUTO, KERNEL_GENERIC, KERNEL_STACK = 0, 1, 2
FLAG_COMPACT, FLAG_COUNT, FLAG_DISPLAY, FLAG_DISPLAY_DEBUG = 0x10, 0x20, 0x40, 0x80
FLAG_TILE_ORDER = 0x100000    # launch the tiles in descending order of their cost in this stream's last frame (latency of one frame)
# include/sdfhip_experimental.h: the A/B knobs of the experiments build (libsdfhip.so refuses them)
FLAG_WIRE = 0x10000
TUNE_ORDER_SHIFT, TUNE_BLOCK_SHIFT = 8, 12
TUNE_ONE_KERNEL = 0x20000     # round 1's one-kernel lane state machine where the default is k_march
TUNE_LDS_TOP = 0x40000        # the one-kernel form with the top grid (level <= 3) staged in LDS per workgroup
TUNE_SHADOW_QUEUE = 0x80000   # FLAG_COMPACT's kernels with EVERY shadow ray queued for k_shadow (round 2's form; SDFHIP_SHADOW_MIN_LANES=T sets the threshold)
TUNE_PERSISTENT_WAVES = 0x400000   # with FLAG_COMPACT on a grid scene: the persistent-wave lane-refill kernel (k_compact), the flag's form until round 4
TUNE_BYTE_CELLS = 0x200000    # on a scene uploaded under SDFHIP_SAMPLE_RECORDS=1: back to the 16-byte cells every other scene reads
SHAPE_SPHERE, SHAPE_TORUS, SHAPE_GYROID = 0, 1, 2


if __name__ != "sdfbox_amd._lib":
    # the package imported a second time against another flavour of the library (sdfbox_amd.lab.load()): both flavours share ONE
    # set of ctypes classes, so that a camera, a PathTrace or a Stats object made with either package is accepted by both
    from sdfbox_amd._lib import COctData, CPoints, Info, MultiLink, MultiStats, PathTrace, SdfGenStats, SdfHipError, Stats, UploadOptions   # noqa: F401
else:
    class Info(ctypes.Structure):
        """The 112-byte `Info` cbuffer (Logic.cs:407-420)."""
        _fields_ = [
            ("heading", (ctypes.c_float * 4) * 3),
            ("position", ctypes.c_float * 3),
            ("margin", ctypes.c_float),
            ("screen_size", ctypes.c_float * 2),
            ("buffer_size", ctypes.c_uint32),
            ("limit", ctypes.c_float),
            ("light", ctypes.c_float * 3),
            ("strength", ctypes.c_float),
            ("fov", ctypes.c_float),
            ("hidef", ctypes.c_int32),
            ("pad_", ctypes.c_uint32 * 2),
        ]


    assert ctypes.sizeof(Info) == 112


    class COctData(ctypes.Structure):
        _fields_ = [
            ("length", ctypes.c_uint32),
            ("structs", ctypes.POINTER(ctypes.c_int32)),
            ("values", ctypes.POINTER(ctypes.c_uint8)),
        ]


    class Stats(ctypes.Structure):
        _fields_ = [
            ("kernel_ms", ctypes.c_float),
            ("total_ms", ctypes.c_float),
            ("n_nodes", ctypes.c_uint64),
            ("n_samples", ctypes.c_uint64),
            ("n_steps", ctypes.c_uint64),
            ("kernel_used", ctypes.c_uint32),
            ("pad_", ctypes.c_uint32),
            ("n_shadow_rays", ctypes.c_uint64),
            ("n_loads", ctypes.c_uint64),
            ("n_hits", ctypes.c_uint64),
        ]


    class PathTrace(ctypes.Structure):
        """sdfhip_pathtrace: parameters of the path-traced mode (BASELINE config 5 defaults)."""
        _fields_ = [
            ("spp", ctypes.c_uint32),
            ("max_bounces", ctypes.c_uint32),
            ("seed", ctypes.c_uint32),
            ("albedo", ctypes.c_float),
        ]

        def __init__(self, spp=16, max_bounces=3, seed=0x5DFB0C5, albedo=0.8):
            super().__init__(int(spp), int(max_bounces), int(seed), float(albedo))


    class UploadOptions(ctypes.Structure):
        """sdfhip_upload_options: the upload's own choices, overridden (None / -1 = choose)."""
        _fields_ = [("size", ctypes.c_uint32), ("top_grid_level", ctypes.c_int32), ("top_grid_split", ctypes.c_int32),
                    ("scatter_grid", ctypes.c_int32), ("scatter_order", ctypes.c_int32)]

        def __init__(self, top_grid_level=None, top_grid_split=None, scatter_grid=None, scatter_order=None):
            f = lambda v: -1 if v is None else int(v)
            super().__init__(ctypes.sizeof(type(self)), f(top_grid_level), f(top_grid_split), f(scatter_grid), f(scatter_order))


    class CPoints(ctypes.Structure):
        _fields_ = [("count", ctypes.c_uint32), ("data", ctypes.POINTER(ctypes.c_float))]


    class SdfGenStats(ctypes.Structure):
        _fields_ = [
            ("nodes", ctypes.c_uint32), ("levels", ctypes.c_uint32),
            ("candidate_entries", ctypes.c_uint64),
            ("global_scale", ctypes.c_float), ("global_offset", ctypes.c_float * 3),
            ("total_ms", ctypes.c_float),
        ]


    class MultiLink(ctypes.Structure):
        _fields_ = [("device", ctypes.c_int32), ("peer_access", ctypes.c_int32), ("ok", ctypes.c_uint32), ("push_ms", ctypes.c_float),
                    ("pci_bus_id", ctypes.c_char * 16)]


    class MultiStats(ctypes.Structure):
        _fields_ = [
            ("total_ms", ctypes.c_float), ("n_devices", ctypes.c_uint32), ("resends", ctypes.c_uint32), ("pad_", ctypes.c_uint32),
            ("gathered_bytes", ctypes.c_uint64), ("rank_ms", ctypes.c_float * 16), ("floats_used", ctypes.c_uint32 * 16),
        ]


    class SdfHipError(RuntimeError):
        def __init__(self, code, message):
            super().__init__(f"sdfhip error {code}: {message}")
            self.code = code


_c = ctypes
_vp = ctypes.c_void_p
_SIG = {
    "sdfhip_last_error": (_c.c_char_p, []),
    "sdfhip_asdf_load": (_c.c_int, [_c.c_char_p, _c.POINTER(COctData)]),
    "sdfhip_asdf_save": (_c.c_int, [_c.POINTER(COctData), _c.c_char_p]),
    "sdfhip_octdata_free": (None, [_c.POINTER(COctData)]),
    "sdfhip_generate": (_c.c_int, [_c.c_int, _c.POINTER(_c.c_float), _c.c_int, _c.c_int, _c.c_int,
                                   _c.POINTER(COctData)]),
    "sdfhip_load_ply": (_c.c_int, [_c.c_char_p, _c.POINTER(CPoints)]),
    "sdfhip_load_obj": (_c.c_int, [_c.c_char_p, _c.POINTER(CPoints)]),
    "sdfhip_points_free": (None, [_c.POINTER(CPoints)]),
    "sdfhip_sdfgen": (_c.c_int, [_c.c_int, _vp, _c.c_uint32, _c.c_int32, _c.POINTER(COctData),
                                 _c.POINTER(SdfGenStats)]),
    "sdfhip_sdfgen_scene": (_c.c_int, [_c.c_int, _vp, _c.c_uint32, _c.c_int32, _c.POINTER(_vp), _c.POINTER(COctData),
                                       _c.POINTER(SdfGenStats)]),
    "sdfhip_sdfgen_trim": (_c.c_int, []),
    "sdfhip_octdata_validate": (_c.c_int, [_vp, _c.c_uint32, _c.POINTER(_c.c_uint32),
                                           _c.POINTER(_c.c_int)]),
    "sdfhip_info_default": (None, [_c.POINTER(Info), _c.c_float, _c.c_float]),
    "sdfhip_info_set_heading": (None, [_c.POINTER(Info), _c.c_float, _c.c_float]),
    "sdfhip_info_set_position": (None, [_c.POINTER(Info), _c.c_float, _c.c_float, _c.c_float]),
    "sdfhip_device_count": (_c.c_int, [_c.POINTER(_c.c_int)]),
    "sdfhip_device_pci_bus_id": (_c.c_int, [_c.c_int, _c.c_char_p, _c.c_uint32]),
    "sdfhip_device_bandwidth": (_c.c_int, [_c.c_int, _c.c_uint64, _c.c_uint32, _c.POINTER(_c.c_double), _c.POINTER(_c.c_double), _c.POINTER(_c.c_double)]),
    "sdfhip_multi_selftest": (_c.c_int, [_vp, _c.POINTER(MultiLink)]),
    "sdfhip_scene_upload": (_c.c_int, [_c.c_int, _vp, _vp, _c.c_uint32, _c.POINTER(_vp)]),
    "sdfhip_upload_options_default": (None, [_c.POINTER(UploadOptions)]),
    "sdfhip_scene_upload_ex": (_c.c_int, [_c.c_int, _vp, _vp, _c.c_uint32, _c.POINTER(UploadOptions), _c.POINTER(_vp)]),
    "sdfhip_scene_free": (_c.c_int, [_vp]),
    "sdfhip_scene_info": (_c.c_int, [_vp, _c.POINTER(_c.c_uint32), _c.POINTER(_c.c_uint32),
                                     _c.POINTER(_c.c_int), _c.POINTER(_c.c_int)]),
    "sdfhip_render": (_c.c_int, [_vp, _c.POINTER(Info), _c.c_uint32, _c.c_uint32, _c.c_uint32, _vp,
                                 _c.POINTER(Stats)]),
    "sdfhip_render_device": (_c.c_int, [_vp, _c.POINTER(Info), _c.c_uint32, _c.c_uint32,
                                        _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32,
                                        _c.c_uint32, _vp, _vp, _c.POINTER(Stats)]),
    "sdfhip_render_batch_device": (_c.c_int, [_vp, _c.POINTER(Info), _c.c_uint32, _c.c_uint32, _c.c_uint32,
                                              _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32,
                                              _c.c_uint32, _vp, _vp, _c.POINTER(Stats)]),
    "sdfhip_scene_prepare_path": (_c.c_int, [_vp]),
    "sdfhip_render_path": (_c.c_int, [_vp, _c.POINTER(Info), _c.POINTER(PathTrace), _c.c_uint32, _c.c_uint32,
                                      _c.c_uint32, _vp, _c.POINTER(Stats)]),
    "sdfhip_render_path_device": (_c.c_int, [_vp, _c.POINTER(Info), _c.POINTER(PathTrace), _c.c_uint32,
                                             _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32,
                                             _c.c_uint32, _c.c_uint32, _vp, _vp, _c.POINTER(Stats)]),
    "sdfhip_render_display": (_c.c_int, [_vp, _c.POINTER(Info), _c.c_uint32, _c.c_uint32, _c.c_uint32,
                                         _c.c_int, _vp, _c.POINTER(Stats)]),
    "sdfhip_host_alloc": (_c.c_int, [_c.c_uint64, _c.POINTER(_vp)]),
    "sdfhip_host_register": (_c.c_int, [_vp, _c.c_uint64]),
    "sdfhip_host_release": (_c.c_int, [_vp]),
    "sdfhip_deinterleave_device": (_c.c_int, [_c.c_int, _vp, _vp, _c.c_uint32, _c.c_uint32,
                                              _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32, _vp]),
    "sdfhip_camera_update": (None, [_c.POINTER(Info), _c.POINTER(_c.c_float), _c.c_float, _c.c_uint32, _c.c_float]),
    "sdfhip_camera_mouse_move": (None, [_c.POINTER(Info), _c.POINTER(_c.c_float), _c.c_float, _c.c_float]),
    "sdfhip_camera_mouse_wheel": (_c.c_float, [_c.c_float, _c.c_float]),
    "sdfhip_scene_top_grid": (_c.c_int, [_vp, _c.POINTER(_c.c_int32), _c.POINTER(_c.c_uint64)]),
    "sdfhip_render_bands_device": (_c.c_int, [_vp, _c.POINTER(Info), _c.c_uint32, _c.POINTER(PathTrace), _c.c_uint32,
                                              _c.c_uint32, _c.c_uint32, _c.POINTER(_c.c_uint16), _c.c_uint32,
                                              _c.c_uint32, _c.c_uint32, _vp, _vp, _c.POINTER(Stats)]),
    "sdfhip_deinterleave_bands_device": (_c.c_int, [_c.c_int, _vp, _vp, _c.c_uint32, _c.c_uint32, _c.c_uint32,
                                                    _c.c_uint32, _c.c_uint32, _c.POINTER(_c.c_uint8), _c.c_uint32,
                                                    _c.c_uint32, _vp]),
    "sdfhip_sparse2_bytes": (_c.c_uint64, [_c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32]),
    "sdfhip_sparse2_floats_offset": (_c.c_uint64, [_c.c_uint32, _c.c_uint32, _c.c_uint32]),
    "sdfhip_render_sparse_device": (_c.c_int, [_vp, _c.POINTER(Info), _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32,
                                               _c.POINTER(_c.c_uint16), _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32, _vp, _vp]),
    "sdfhip_deinterleave_sparse2_device": (_c.c_int, [_c.c_int, _c.POINTER(_vp), _vp, _c.c_uint32, _c.c_uint32, _c.c_uint32,
                                                      _c.c_uint32, _c.c_uint32, _c.POINTER(_c.c_uint8), _c.c_uint32, _c.c_uint32,
                                                      _c.c_uint32, _c.c_int, _vp, _vp]),
    "sdfhip_multi_create": (_c.c_int, [_c.POINTER(_c.c_int), _c.c_uint32, _vp, _vp, _c.c_uint32, _c.POINTER(_vp)]),
    "sdfhip_multi_free": (_c.c_int, [_vp]),
    "sdfhip_multi_configure": (_c.c_int, [_vp, _c.c_uint32, _c.c_float]),
    "sdfhip_multi_info": (_c.c_int, [_vp, _c.POINTER(_c.c_uint32), _c.POINTER(_c.c_int), _c.POINTER(_c.c_uint32),
                                     _c.POINTER(_c.c_float), _c.POINTER(_c.c_int)]),
    "sdfhip_multi_render": (_c.c_int, [_vp, _c.POINTER(Info), _c.c_uint32, _c.c_uint32, _c.c_uint32, _vp, _c.POINTER(MultiStats)]),
    "sdfhip_multi_render_path": (_c.c_int, [_vp, _c.POINTER(Info), _c.POINTER(PathTrace), _c.c_uint32, _c.c_uint32, _c.c_uint32,
                                            _vp, _c.POINTER(MultiStats)]),
    "sdfhip_multi_submit": (_c.c_int, [_vp, _c.c_uint32, _c.POINTER(Info), _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32, _vp]),
    "sdfhip_multi_submit_path": (_c.c_int, [_vp, _c.c_uint32, _c.POINTER(Info), _c.POINTER(PathTrace), _c.c_uint32, _c.c_uint32,
                                            _c.c_uint32, _vp]),
    "sdfhip_multi_wait": (_c.c_int, [_vp, _c.c_uint32, _c.POINTER(_vp), _c.POINTER(MultiStats)]),
}
# include/sdfhip_experimental.h: exported by the experiments flavour only
_SIG_LAB = {
    "sdfhip_wire_sparse_bytes": (_c.c_uint64, [_c.c_uint32, _c.c_uint32, _c.c_uint32]),
    "sdfhip_wire_sparse_head_offset": (_c.c_uint64, [_c.c_uint32, _c.c_uint32, _c.c_uint32]),
    "sdfhip_deinterleave_share_device": (_c.c_int, [_c.c_int, _vp, _vp, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32,
                                                    _c.c_uint32, _c.POINTER(_c.c_uint8), _c.c_uint32, _c.c_uint32, _c.c_uint32, _vp]),
    "sdfhip_wire_compact_device": (_c.c_int, [_c.c_int, _vp, _vp, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32, _vp]),
    "sdfhip_deinterleave_sparse_device": (_c.c_int, [_c.c_int, _vp, _vp, _c.c_uint32, _c.c_uint32, _c.c_uint32, _c.c_uint32,
                                                     _c.c_uint32, _c.POINTER(_c.c_uint8), _c.c_uint32, _c.c_uint32, _vp, _vp]),
    "sdfhip_debug_tile_order": (_c.c_int, [_vp, _vp, _vp]),
    "sdfhip_debug_unorm_table": (_c.c_int, [_c.c_int, _vp]),
    "sdfhip_debug_step_classes": (_c.c_int, [_vp, _vp, _c.POINTER(_c.c_uint64)]),
    "sdfhip_multi_debug_floats_sent": (_c.c_int, [_vp, _c.c_uint32]),
    "sdfhip_debug_fail_host_allocations": (_c.c_int, [_c.c_int64, _c.POINTER(_c.c_uint64)]),
    "sdfhip_debug_touch_begin": (_c.c_int, [_vp]),
    "sdfhip_debug_touch_end": (_c.c_int, [_vp, _c.POINTER(_c.c_uint64), _c.c_uint32, _c.POINTER(_c.c_uint32), _c.POINTER(_c.c_uint64)]),
}
# every symbol include/sdfhip.h declares must be exported: fail at import otherwise
for _name, (_res, _args) in _SIG.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args
EXPERIMENTS = hasattr(lib, "sdfhip_debug_unorm_table")       # libsdfhip_lab.so: then every symbol of sdfhip_experimental.h must be there
if EXPERIMENTS:
    for _name, (_res, _args) in _SIG_LAB.items():
        _fn = getattr(lib, _name)
        _fn.restype = _res
        _fn.argtypes = _args

EXPORTED_SYMBOLS = tuple(_SIG)
EXPERIMENTAL_SYMBOLS = tuple(_SIG_LAB)


def need_lab(what):
    if not EXPERIMENTS:
        raise RuntimeError(f"{what} is part of the experiments build (include/sdfhip_experimental.h): load libsdfhip_lab.so "
                           "(sdfbox_amd.lab.load(), or SDFHIP_LIB=sdfbox_amd/libsdfhip_lab.so)")



def check(code):
    if code != OK:
        raise SdfHipError(code, lib.sdfhip_last_error().decode("utf-8", "replace"))
