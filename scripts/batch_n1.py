"""Dev script: whole 1080p / 4K frames, G frames per launch (grid.y), 2 launches in flight."""
import sys, time
sys.path.insert(0, ".")
import torch
import sdfbox_amd as sb
od = sb.dragon_standin(9); sc = sb.Scene(od)
for (W, H) in ((1920, 1080), (3840, 2160)):
    cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
    for G in (1, 2, 4, 8):
        streams = [torch.cuda.Stream() for _ in range(2)]
        bufs = [torch.zeros((G, H, W, 4), device="cuda") for _ in range(2)]
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(24):
                sc.DrawBatchDevice([cam] * G, W, H, bufs[k % 2].data_ptr(), stream=streams[k % 2].cuda_stream)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / (24 * G) * 1e3)
        print(f"{W}x{H}: {G} frames per launch, 2 launches in flight: {best:.4f} ms per frame -> {W*H/best/1e3:.0f} Mray/s", flush=True)
