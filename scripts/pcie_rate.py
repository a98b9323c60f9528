"""Dev script: PCIe-inclusive frame time through sdfhip_render (host buffer)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import sdfbox_amd as sb
od = sb.dragon_standin(9); sc = sb.Scene(od)
for (W, H) in [(1920, 1080), (3840, 2160)]:
    cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
    ts = []
    for i in range(12):
        img, st = sc.Draw(cam, W, H, want_stats=True)
        ts.append((st.kernel_ms, st.total_ms))
    k = np.median([t[0] for t in ts[2:]]); t = np.median([t[1] for t in ts[2:]])
    print(f"{W}x{H}: kernel {k:.3f} ms, kernel + D2H into pageable host memory {t:.3f} ms -> {W*H/t/1e3:.0f} Mray/s ({W*H*16/1e6:.1f} MB frame, {W*H*16/((t-k)*1e-3)/1e9:.1f} GB/s effective copy)")
