"""Dev script (GPU): wave-iterations of k_march on the bench frame = sum over the 8x8 tiles of the iterations their wave ran
(the longest primary march of the tile, low byte; the longest shadow march of the tile, high byte), from the kernel's own
per-tile cost output.  PMC instruction counts per launch divided by this give VALU / SALU per wave-iteration
(profiles/r05_isa_k_march.txt)."""
import ctypes, sys
sys.path.insert(0, ".")
import numpy as np, torch
import sdfbox_amd as sb
from sdfbox_amd._lib import lib, check
for size in ("1920x1080", "3840x2160"):
    W, H = (int(v) for v in size.split("x"))
    od = sb.dragon_standin(9, nthreads=32); sc = sb.Scene(od)
    cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
    tx, ty = (W + 7) // 8, (H + 7) // 8
    cost = torch.zeros(tx * ty, dtype=torch.int16, device="cuda")
    buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    check(lib.sdfhip_debug_tile_order(sc._h, None, ctypes.c_void_p(cost.data_ptr())))
    st = sb.Stats()
    sc.DrawDevice(cam, W, H, buf.data_ptr(), flags=sb.FLAG_COUNT, stats=st)
    torch.cuda.synchronize()
    raw = cost.cpu().numpy().astype(np.int64) & 0xFFFF
    c, sh = raw & 0xFF, raw >> 8
    steps = buf[..., 3].sum().item()
    print(f"{size}: {tx * ty} waves, k_march wave-iterations: primary loop {c.sum()} (mean {c.mean():.1f}, max {c.max()}), shadow loop {sh.sum()} "
          f"in {int((sh > 0).sum())} waves (mean {sh[sh > 0].mean():.1f}, max {sh.max()}); all march steps {int(steps)}, "
          f"shadow rays {st.n_shadow_rays}, cell loads {st.n_loads}")
    check(lib.sdfhip_debug_tile_order(sc._h, None, None))
    sc.close()
