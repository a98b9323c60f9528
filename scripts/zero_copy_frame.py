"""Experiment: the march kernel storing its pixels straight into page-locked host memory (no device frame, no copy) against
sdfhip_render's march + copy.  GPU: gpurun -- python scripts/zero_copy_frame.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import sdfbox_amd as sb

od = sb.dragon_standin(9, nthreads=32)
sc = sb.Scene(od)
for W, H in ((1920, 1080), (3840, 2160)):
    cams = bench.orbit_cameras(sb, W, H, 90)
    for dtype, flags, name in ((np.float32, 0, "RGBA32F"), (np.uint8, sb.FLAG_DISPLAY, "RGBA8 display")):
        hf = sb.HostFrame(H, W, dtype)
        ref = np.empty((H, W, 4), dtype=dtype)
        sb._lib.check(sb._lib.lib.sdfhip_render(sc._h, __import__("ctypes").byref(cams[0].State), W, H, flags, ref.ctypes.data, None))
        for order in (0, sb.FLAG_TILE_ORDER):
            ts = []
            for k in range(60):
                t = time.perf_counter()
                sc.DrawDevice(cams[0], W, H, hf.array.ctypes.data, flags=flags | order)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t)
            same = bool(np.array_equal(hf.array.view(np.uint8), ref.view(np.uint8)))
            print(f"{W}x{H} {name:14s} stores into host memory{' (tile order)' if order else '':13s}: {1e3 * float(np.median(ts[10:])):.3f} ms per frame, identical {same}", flush=True)
        hf.close()
