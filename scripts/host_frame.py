"""The viewer's call, sdfhip_render / sdfhip_render_display: wall time per frame into a reused host array, camera at rest and
moving, banded (default) and as one launch + one copy (SDFHIP_HOST_BANDS=0).  GPU: gpurun -- python scripts/host_frame.py"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np

    import bench
    import sdfbox_amd as sb
    od = sb.dragon_standin(9, nthreads=32)
    sc = sb.Scene(od)
    for W, H in ((1920, 1080), (3840, 2160)):
        cams = bench.orbit_cameras(sb, W, H, 90)
        if os.environ.get("HOST_FRAME_LOCKED") == "alloc":
            keep = (sb.HostFrame(H, W, np.float32), sb.HostFrame(H, W, np.uint8))
            out, out8 = keep[0].array, keep[1].array
        else:
            out = np.empty((H, W, 4), dtype=np.float32)
            out8 = np.empty((H, W, 4), dtype=np.uint8)
            if os.environ.get("HOST_FRAME_LOCKED") == "register":
                keep = (sb.HostFrame(array=out), sb.HostFrame(array=out8))
        for what, fn in (("RGBA32F", lambda c: sc.Draw(c, W, H, out=out)), ("RGBA8 display", lambda c: sb._lib.check(sb._lib.lib.sdfhip_render_display(sc._h, __import__("ctypes").byref(c.State), W, H, 0, 0, out8.ctypes.data, None)))):
            for moving in (False, True):
                ts = []
                for k in range(60):
                    c = cams[k % 90] if moving else cams[0]
                    t = time.perf_counter(); fn(c); ts.append(time.perf_counter() - t)
                print(f"  {W}x{H} {what:14s} camera {'moving' if moving else 'at rest'}: {1e3 * float(np.median(ts[10:])):.3f} ms per frame")
    sys.exit(0)

for locked in ("alloc", "register") if "--locked" in sys.argv else ():
    for bands in ("", "1", "4"):
        print(f"page-locked destination ({locked}), " + (f"SDFHIP_HOST_BANDS={bands} (band copies)" if bands else "default"))
        sys.stdout.flush()
        env = {k: v for k, v in os.environ.items() if k != "SDFHIP_HOST_BANDS"}
        env["HOST_FRAME_LOCKED"] = locked
        if bands:
            env["SDFHIP_HOST_BANDS"] = bands
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
if "--locked" in sys.argv:
    sys.exit(0)
for bands in ("", "1", "2", "4", "0"):
    print(f"SDFHIP_HOST_BANDS={bands or chr(34)+chr(34)}" + (" (default: one band per 16 MB of frame, copies beside the march, tile order)" if not bands else " (one launch, one copy, no tile order)" if bands == "0" else ""))
    sys.stdout.flush()
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=(dict(os.environ, SDFHIP_HOST_BANDS=bands) if bands else {k: v for k, v in os.environ.items() if k != "SDFHIP_HOST_BANDS"}))
