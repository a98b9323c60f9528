"""sdfbox_amd -- MI355X (gfx950) sphere tracing of adaptively sampled distance
fields behind SdfBox's frame boundary.  The compute lives in libsdfhip.so
(hand-written HIP, C ABI in include/sdfhip.h); this package is the host-side
mirror of the reference's interface for that path: OctData (.asdf), Logic
(camera -> Info), Scene.Draw (Program.Draw's compute pass).
"""
from . import _lib, tiles
from ._lib import (FLAG_COMPACT, FLAG_COUNT, FLAG_DISPLAY, FLAG_DISPLAY_DEBUG, FLAG_TILE_ORDER, FLAG_WIRE, KERNEL_AUTO, KERNEL_GENERIC, KERNEL_STACK, TUNE_ONE_KERNEL, TUNE_SHADOW_QUEUE, Info,
                   MultiStats, PathTrace, SdfHipError, Stats)
from .logic import Logic
from .octdata import OctData, dragon_standin, knot_point_cloud, sphere_d4, torus_d6, write_ply
from .renderer import HostFrame, MultiScene, Scene, device_bandwidth, device_count, device_pci_bus_id, sdfgen_trim, unorm_table

__all__ = [
    "FLAG_COMPACT", "FLAG_COUNT", "FLAG_DISPLAY", "FLAG_DISPLAY_DEBUG", "FLAG_TILE_ORDER", "FLAG_WIRE", "KERNEL_AUTO", "KERNEL_GENERIC", "KERNEL_STACK", "TUNE_ONE_KERNEL", "TUNE_SHADOW_QUEUE", "Info",
    "PathTrace", "SdfHipError", "Stats", "Logic", "OctData", "dragon_standin", "knot_point_cloud", "write_ply", "sphere_d4", "torus_d6",
    "Scene", "HostFrame", "MultiScene", "MultiStats", "device_bandwidth", "device_count", "device_pci_bus_id", "sdfgen_trim", "unorm_table",
]
