"""Dev script: PCIe-inclusive frame time through sdfhip_render (host buffer)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import sdfbox_amd as sb
od = sb.dragon_standin(9); sc = sb.Scene(od)
for (W, H) in [(1920, 1080), (3840, 2160)]:
    cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
    buf = np.empty((H, W, 4), dtype=np.float32)
    for what, kw in (("a fresh pageable array per frame", {}), ("one pageable array reused", {"out": buf})):
        ts = []
        for i in range(12):
            img, st = sc.Draw(cam, W, H, want_stats=True, **kw)
            ts.append((st.kernel_ms, st.total_ms))
        k = np.median([t[0] for t in ts[2:]]); t = np.median([t[1] for t in ts[2:]])
        print(f"{W}x{H}, {what}: kernel {k:.3f} ms, kernel + D2H {t:.3f} ms -> {W*H/t/1e3:.0f} Mray/s "
              f"({W*H*16/1e6:.1f} MB frame, {W*H*16/((t-k)*1e-3)/1e9:.1f} GB/s effective copy)")
