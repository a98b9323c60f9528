import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import sdfbox_amd as sb
od = sb.dragon_standin(9); sc = sb.Scene(od)
for (W, H) in [(1920, 1080), (3840, 2160)]:
    cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
    for what, buf in (("pageable, reused", np.empty((H, W, 4), dtype=np.float32)), ("pinned (torch), reused", torch.empty((H, W, 4), dtype=torch.float32).pin_memory().numpy())):
        ts = []
        for i in range(14):
            img, st = sc.Draw(cam, W, H, want_stats=True, out=buf)
            ts.append((st.kernel_ms, st.total_ms))
        k = np.median([t[0] for t in ts[3:]]); t = np.median([t[1] for t in ts[3:]])
        print(f"{W}x{H}, {what}: kernel {k:.3f} ms, kernel + D2H {t:.3f} ms ({W*H*16/1e6:.1f} MB, {W*H*16/((t-k)*1e-3)/1e9:.1f} GB/s effective copy)")
