import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import sdfbox_amd as sb
from sdfbox_amd.tiles import BandLayout, deinterleave, render_bands
from conftest import make_camera
scene = sb.Scene(sb.torus_d6())
def fl(v):
    f = {"generic": sb.KERNEL_GENERIC, "stack": sb.KERNEL_STACK}[v.split("+")[0]]
    return f | (sb.FLAG_COMPACT if v.endswith("compact") else 0)
for (W, H, world, band_rows, variant) in [(160, 100, 3, 16, "stack"), (90, 77, 4, 8, "stack+compact"), (64, 64, 8, 16, "generic"), (70, 50, 2, 24, "generic+compact")]:
    cam = make_camera("rotated", W, H)
    full = torch.from_numpy(scene.Draw(cam, W, H, fl(variant))).cuda()
    lay = BandLayout(H, world, band_rows)
    gathered = torch.zeros((world, lay.rows_per_rank, W, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for r in range(world):
        render_bands(scene, cam, W, lay, r, gathered[r].data_ptr(), flags=fl(variant), stream=stream)
    frame = torch.full((H, W, 4), -1.0, dtype=torch.float32, device="cuda")
    deinterleave(0, gathered.data_ptr(), frame.data_ptr(), W, lay, stream=stream)
    torch.cuda.synchronize()
    bad = (frame.view(torch.int32) != full.view(torch.int32)).any(-1).cpu().numpy()
    print(W, H, world, band_rows, variant, "bad pixels", bad.sum(), "bad rows", np.nonzero(bad.any(1))[0][:20], "bad cols", np.nonzero(bad.any(0))[0][:20])
    if bad.any():
        y, x = np.argwhere(bad)[0]
        print("  first", x, y, frame[y, x].cpu().numpy(), full[y, x].cpu().numpy(), "src", lay.source_of(int(y)), gathered[lay.source_of(int(y))[0], lay.source_of(int(y))[1], x].cpu().numpy())
