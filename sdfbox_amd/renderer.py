"""Frame driver: what `Program.Draw` does for the compute pass
(SdfBox/Program.cs:79-110: UpdateBuffer(info) + DispatchSized(W, H, 1)),
on an MI355X through libsdfhip.so.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import Info, MultiStats, PathTrace, Stats, check, lib


class HostFrame:
    """A frame array in page-locked host memory (sdfhip_host_alloc), or the caller's own array page-locked where it lies
    (sdfhip_host_register): Scene.Draw / DrawDisplay into `.array` run the frame's copy to the host beside its march.
    Keep the object for as long as the array is in use; close() (or the end of a `with`) gives the memory back."""

    def __init__(self, height=None, width=None, dtype=np.float32, array=None):
        self._p = None
        self._owned = array is None
        if array is None:
            shape = (int(height), int(width), 4)
            nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
            p = ctypes.c_void_p()
            check(lib.sdfhip_host_alloc(nbytes, ctypes.byref(p)))
            self._p = p.value
            buf = (ctypes.c_uint8 * nbytes).from_address(self._p)
            self.array = np.frombuffer(buf, dtype=dtype).reshape(shape)
        else:
            if not array.flags.c_contiguous or not array.flags.writeable:
                raise ValueError("HostFrame: the array must be C-contiguous and writeable")
            check(lib.sdfhip_host_register(array.ctypes.data, array.nbytes))
            self._p = array.ctypes.data
            self.array = array

    def close(self):
        if self._p is not None:
            p, self._p = self._p, None
            self.array = None
            check(lib.sdfhip_host_release(ctypes.c_void_p(p)))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """A scene resident in one GPU's HBM (replaces the `data` / `values`
    bindings of Program.cs:147-152)."""

    def __init__(self, octdata, device=0, top_grid_level=None, top_grid_split=None, scatter_grid=None, scatter_order=None):
        """top_grid_level / top_grid_split / scatter_grid / scatter_order: sdfhip_upload_options (None = the upload chooses):
        a plain lookup grid of that level (0 = none), a split grid with that coarse level (0 = never), the levels of the
        path-traced mode's second grid's blocks (0 = none) and 0 for its blocks in x-y-z order.  Pixels never depend on them."""
        self._h = ctypes.c_void_p()
        self.device = int(device)
        if (top_grid_level, top_grid_split, scatter_grid, scatter_order) == (None, None, None, None):
            check(lib.sdfhip_scene_upload(self.device, octdata.Structs.ctypes.data,
                                          octdata.Values.ctypes.data, octdata.Length,
                                          ctypes.byref(self._h)))
        else:
            opt = _lib.UploadOptions(top_grid_level, top_grid_split, scatter_grid, scatter_order)
            check(lib.sdfhip_scene_upload_ex(self.device, octdata.Structs.ctypes.data, octdata.Values.ctypes.data, octdata.Length,
                                             ctypes.byref(opt), ctypes.byref(self._h)))
        self._describe()

    def _describe(self):
        n = ctypes.c_uint32(); d = ctypes.c_uint32(); ok = ctypes.c_int(); dev = ctypes.c_int()
        check(lib.sdfhip_scene_info(self._h, ctypes.byref(n), ctypes.byref(d), ctypes.byref(ok),
                                    ctypes.byref(dev)))
        self.Length, self.depth, self.stack_kernel_ok = n.value, d.value, bool(ok.value)
        lvl = ctypes.c_int32(); nb = ctypes.c_uint64()
        check(lib.sdfhip_scene_top_grid(self._h, ctypes.byref(lvl), ctypes.byref(nb)))
        self.top_grid_level, self.top_grid_bytes = lvl.value, nb.value

    @classmethod
    def FromPoints(cls, vertices, depth, device=0, want_octdata=False, want_stats=False):
        """The viewer's generate -> upload flow in one call (sdfhip_sdfgen_scene; Program.cs:613-650 + :147-152): point
        cloud (n, 6) float32 {position, normal} -> scene handle, the tree never leaving HBM.  want_octdata: also the host
        arrays (for the .asdf cache)."""
        from .octdata import OctData
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 6)
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p()
        self.device = int(device)
        raw = _lib.COctData()
        st = _lib.SdfGenStats()
        check(lib.sdfhip_sdfgen_scene(self.device, v.ctypes.data, len(v), int(depth), ctypes.byref(self._h),
                                      ctypes.byref(raw) if want_octdata else None, ctypes.byref(st)))
        self._describe()
        out = [self]
        if want_octdata:
            out.append(OctData._from_native(raw))
        if want_stats:
            out.append(st)
        return out[0] if len(out) == 1 else tuple(out)

    def close(self):
        if self._h:
            lib.sdfhip_scene_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- Program.Draw ----------------------------------------------------
    def Draw(self, state, width, height, flags=_lib.KERNEL_AUTO, want_stats=False, out=None):
        """Render one frame to a host array (H, W, 4) float32; alpha = step count.  out: reuse this
        array (a renderer does: a fresh 4K array per frame costs 5 ms of first-touch page faults in the copy)."""
        if out is None:
            out = np.empty((int(height), int(width), 4), dtype=np.float32)
        elif out.shape != (int(height), int(width), 4) or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError("Draw: out must be a C-contiguous float32 array of shape (height, width, 4)")
        st = Stats()
        info = state if isinstance(state, Info) else state.State
        check(lib.sdfhip_render(self._h, ctypes.byref(info), int(width), int(height), int(flags),
                                out.ctypes.data, ctypes.byref(st) if want_stats else None))
        return (out, st) if want_stats else out

    def DrawBatchDevice(self, states, width, height, out_ptr, nrows_out=None, band_rows=None,
                        band_first=0, band_stride=1, flags=_lib.KERNEL_AUTO, stream=None, stats=None):
        """Several frames (<= 8 cameras) in one launch into out_ptr[f][nrows_out][width] pixels."""
        infos = (Info * len(states))(*[s if isinstance(s, Info) else s.State for s in states])
        if nrows_out is None:
            nrows_out = height
        if band_rows is None:
            band_rows = height
        check(lib.sdfhip_render_batch_device(self._h, infos, len(states), int(width), int(height),
                                             int(band_rows), int(band_first), int(band_stride),
                                             int(nrows_out), int(flags), ctypes.c_void_p(int(out_ptr)),
                                             ctypes.c_void_p(int(stream)) if stream else None,
                                             ctypes.byref(stats) if stats is not None else None))

    def DrawBandsDevice(self, states, width, height, out_ptr, band_rows, bands, nrows_out=None, pt=None,
                        flags=_lib.KERNEL_AUTO, stream=None, stats=None):
        """The listed bands (band indices of the frame, `band_rows` rows each) of one or several
        frames into out_ptr[f][nrows_out][width] pixels: local band i = bands[i].  pt: path-traced
        mode (one frame)."""
        if not isinstance(states, (list, tuple)):
            states = [states]
        infos = (Info * len(states))(*[s if isinstance(s, Info) else s.State for s in states])
        blist = (ctypes.c_uint16 * len(bands))(*[int(b) for b in bands])
        if nrows_out is None:
            nrows_out = len(bands) * int(band_rows)
        check(lib.sdfhip_render_bands_device(self._h, infos, len(states), ctypes.byref(pt) if pt is not None else None,
                                             int(width), int(height), int(band_rows), blist, len(bands),
                                             int(nrows_out), int(flags), ctypes.c_void_p(int(out_ptr)),
                                             ctypes.c_void_p(int(stream)) if stream else None,
                                             ctypes.byref(stats) if stats is not None else None))

    def DrawPath(self, state, width, height, pt=None, flags=_lib.KERNEL_AUTO, want_stats=False):
        """Path-traced frame (BASELINE config 5; defined by the oracle's o_pixel_pt): host array
        (H, W, 4) float32, mean radiance + step count."""
        pt = pt if pt is not None else PathTrace()
        out = np.empty((int(height), int(width), 4), dtype=np.float32)
        st = Stats()
        info = state if isinstance(state, Info) else state.State
        check(lib.sdfhip_render_path(self._h, ctypes.byref(info), ctypes.byref(pt), int(width), int(height),
                                     int(flags), out.ctypes.data, ctypes.byref(st) if want_stats else None))
        return (out, st) if want_stats else out

    def DrawPathDevice(self, state, width, height, out_ptr, pt=None, nrows_out=None, band_rows=None,
                       band_first=0, band_stride=1, flags=_lib.KERNEL_AUTO, stream=None, stats=None):
        pt = pt if pt is not None else PathTrace()
        info = state if isinstance(state, Info) else state.State
        if nrows_out is None:
            nrows_out = height
        if band_rows is None:
            band_rows = height
        check(lib.sdfhip_render_path_device(self._h, ctypes.byref(info), ctypes.byref(pt), int(width),
                                            int(height), int(band_rows), int(band_first), int(band_stride),
                                            int(nrows_out), int(flags), ctypes.c_void_p(int(out_ptr)),
                                            ctypes.c_void_p(int(stream)) if stream else None,
                                            ctypes.byref(stats) if stats is not None else None))

    def DrawDisplay(self, state, width, height, debug=False, flags=_lib.KERNEL_AUTO, want_stats=False, out=None):
        """Render + display pass (DisplayFrag.hlsl) fused: host array (H, W, 4) uint8, R,G,B,A.
        debug=True gives the step-count heat map of DisplayFrag.hlsl:21-22.  out: reuse this array."""
        if out is None:
            out = np.empty((int(height), int(width), 4), dtype=np.uint8)
        elif out.shape != (int(height), int(width), 4) or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError("DrawDisplay: out must be a C-contiguous uint8 array of shape (height, width, 4)")
        st = Stats()
        info = state if isinstance(state, Info) else state.State
        check(lib.sdfhip_render_display(self._h, ctypes.byref(info), int(width), int(height), int(flags),
                                        1 if debug else 0, out.ctypes.data,
                                        ctypes.byref(st) if want_stats else None))
        return (out, st) if want_stats else out

    def DrawDevice(self, state, width, height, out_ptr, nrows_out=None, band_rows=None,
                   band_first=0, band_stride=1, flags=_lib.KERNEL_AUTO, stream=None, stats=None):
        """Render into device memory at `out_ptr` (nrows_out x width x 4 floats),
        asynchronously on `stream` (a raw hipStream_t value or None)."""
        info = state if isinstance(state, Info) else state.State
        if nrows_out is None:
            nrows_out = height
        if band_rows is None:
            band_rows = height
        check(lib.sdfhip_render_device(self._h, ctypes.byref(info), int(width), int(height),
                                       int(band_rows), int(band_first), int(band_stride),
                                       int(nrows_out), int(flags), ctypes.c_void_p(int(out_ptr)),
                                       ctypes.c_void_p(int(stream)) if stream else None,
                                       ctypes.byref(stats) if stats is not None else None))

    def step_classes(self, stream=None):
        """After a FLAG_COUNT DrawDevice on `stream`: lane-steps by the kind of cell they sampled
        (sdfhip_debug_step_classes)."""
        _lib.need_lab("Scene.step_classes")
        out = (ctypes.c_uint64 * 6)()
        check(lib.sdfhip_debug_step_classes(self._h, ctypes.c_void_p(int(stream)) if stream else None, out))
        return dict(zip(("flat_coarse", "flat_fine", "nonflat_coarse", "nonflat_full_depth", "nonflat_between", "nonflat_outside"),
                        (int(v) for v in out)))

    def touch_begin(self):
        """From here to touch_end() every FLAG_COUNT render marks the 128-byte lines of the lookup grid its find() touches, per XCD
        (sdfhip_debug_touch_begin; laboratory library)."""
        _lib.need_lab("Scene.touch_begin")
        check(lib.sdfhip_debug_touch_begin(self._h))

    def touch_end(self):
        """-> {"phases": [{"grid": "own" | "bounce", "coarse_lines", "fine_lines", "coarse_lines_xcd_sum", "fine_lines_xcd_sum"}, ...],
        "array_bytes": {"coarse", "fine", "coarse2", "fine2"}}: the distinct lines each counted launch touched (a frame of the
        default kernel is one phase; a path-traced frame is its camera segments and then one phase per bounce level), chip-wide and
        summed over the XCDs (sdfhip_debug_touch_end)."""
        _lib.need_lab("Scene.touch_end")
        out = (ctypes.c_uint64 * (16 * 8))()
        n = ctypes.c_uint32(0)
        ab = (ctypes.c_uint64 * 4)()
        check(lib.sdfhip_debug_touch_end(self._h, out, 16, ctypes.byref(n), ab))
        phases = [{"grid": "own" if int(out[8 * i + 4]) == 0 else "bounce", "coarse_lines": int(out[8 * i]), "fine_lines": int(out[8 * i + 1]),
                   "coarse_lines_xcd_sum": int(out[8 * i + 2]), "fine_lines_xcd_sum": int(out[8 * i + 3])} for i in range(n.value)]
        return {"phases": phases, "array_bytes": dict(zip(("coarse", "fine", "coarse2", "fine2"), (int(v) for v in ab)))}


class MultiScene:
    """A scene replicated on several GPUs of one node; a frame is ONE call (sdfhip_multi_*: bands dealt to the devices,
    sparse wire shares gathered into devices[0] over xGMI, assembled there).  `devices` may repeat a device
    (rehearsal of the pipeline on one GPU)."""

    def __init__(self, octdata, devices):
        self._h = ctypes.c_void_p()
        self.devices = [int(d) for d in devices]
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        check(lib.sdfhip_multi_create(arr, len(self.devices), octdata.Structs.ctypes.data, octdata.Values.ctypes.data,
                                      octdata.Length, ctypes.byref(self._h)))

    def close(self):
        if self._h:
            lib.sdfhip_multi_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def configure(self, band_rows=16, rank0_weight=1.0):
        check(lib.sdfhip_multi_configure(self._h, int(band_rows), float(rank0_weight)))

    def selftest(self):
        """First contact with the node's links (sdfhip_multi_selftest; create runs it too): per device its PCI bus id, whether it
        reaches devices[0]'s memory directly, and whether a 1 MB pattern pushed the gather's way arrived intact.  Raises
        SdfHipError naming the pair when one did not."""
        links = (_lib.MultiLink * len(self.devices))()
        check(lib.sdfhip_multi_selftest(self._h, links))
        return [{"device": l.device, "pci_bus_id": l.pci_bus_id.decode(), "peer_access": l.peer_access, "ok": bool(l.ok), "push_ms": l.push_ms}
                for l in links]

    @property
    def transport(self):
        t = ctypes.c_int()
        check(lib.sdfhip_multi_info(self._h, None, None, None, None, ctypes.byref(t)))
        return "rccl" if t.value else "peer"

    def Draw(self, state, width, height, flags=0, pt=None, want_stats=False, out=None):
        """One frame across the devices to a host array: (H, W, 4) float32, or uint8 with FLAG_DISPLAY[_DEBUG]."""
        display = bool(flags & (_lib.FLAG_DISPLAY | _lib.FLAG_DISPLAY_DEBUG))
        want = np.uint8 if display else np.float32
        if out is None:
            out = np.empty((int(height), int(width), 4), dtype=want)
        elif not isinstance(out, np.ndarray) or out.shape != (int(height), int(width), 4) or out.dtype != want or \
                not out.flags.c_contiguous or not out.flags.writeable:
            raise ValueError(f"MultiScene.Draw: out must be a writeable C-contiguous {np.dtype(want).name} array of shape (height, width, 4)")
        st = MultiStats()
        stp = ctypes.byref(st) if want_stats else None             # (statistics cost an event per rank: only when asked for)
        info = state if isinstance(state, Info) else state.State
        if pt is not None:
            check(lib.sdfhip_multi_render_path(self._h, ctypes.byref(info), ctypes.byref(pt), int(width), int(height), int(flags),
                                               out.ctypes.data, stp))
        else:
            check(lib.sdfhip_multi_render(self._h, ctypes.byref(info), int(width), int(height), int(flags), out.ctypes.data, stp))
        return (out, st) if want_stats else out

    def Submit(self, slot, states, width, height, flags=0, out_ptr=None, pt=None):
        """A group of frames (one camera block each) into slot 0..3; frames land in device memory of devices[0]."""
        if not isinstance(states, (list, tuple)):
            states = [states]
        infos = (Info * len(states))(*[s if isinstance(s, Info) else s.State for s in states])
        dst = ctypes.c_void_p(int(out_ptr)) if out_ptr else None
        if pt is not None:
            check(lib.sdfhip_multi_submit_path(self._h, int(slot), infos, ctypes.byref(pt), int(width), int(height), int(flags), dst))
        else:
            check(lib.sdfhip_multi_submit(self._h, int(slot), infos, len(states), int(width), int(height), int(flags), dst))

    def Wait(self, slot, want_stats=False):
        """Block until the slot's frames are complete; returns the device pointer of its frames (and the stats)."""
        p = ctypes.c_void_p()
        st = MultiStats()
        check(lib.sdfhip_multi_wait(self._h, int(slot), ctypes.byref(p), ctypes.byref(st)))
        return (p.value, st) if want_stats else p.value

    def debug_floats_sent(self, floats):
        _lib.need_lab("MultiScene.debug_floats_sent")
        check(lib.sdfhip_multi_debug_floats_sent(self._h, int(floats)))


def device_pci_bus_id(device=0):
    out = ctypes.create_string_buffer(32)
    check(lib.sdfhip_device_pci_bus_id(int(device), out, 32))
    return out.value.decode()


def device_bandwidth(device=0, nbytes=2 << 30, reps=10):
    """(copy, triad, read) GB/s of the device's memory for a streaming kernel (sdfhip_device_bandwidth): arrays of nbytes each."""
    c, t, r = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    check(lib.sdfhip_device_bandwidth(int(device), int(nbytes), int(reps), ctypes.byref(c), ctypes.byref(t), ctypes.byref(r)))
    return c.value, t.value, r.value


def sdfgen_trim():
    """Give back the device memory the point-cloud builder keeps between builds (sdfhip_sdfgen_trim)."""
    check(lib.sdfhip_sdfgen_trim())


def device_count():
    n = ctypes.c_int()
    check(lib.sdfhip_device_count(ctypes.byref(n)))
    return n.value


def unorm_table(device=0):
    _lib.need_lab("unorm_table")
    out = np.empty(256, dtype=np.float32)
    check(lib.sdfhip_debug_unorm_table(int(device), out.ctypes.data))
    return out
