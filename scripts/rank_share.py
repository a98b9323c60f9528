"""Dev script: the render throughput of ONE rank's share of a 1080p frame sharded over
`world` GPUs (no gather), against the number of streams kept busy."""
import os, sys, time
sys.path.insert(0, ".")
import torch
import sdfbox_amd as sb
from sdfbox_amd.tiles import BandLayout, render_bands, render_bands_batch
W, H = 1920, 1080
od = sb.dragon_standin(9); sc = sb.Scene(od)
cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
for world in (2, 4, 8):
    lay = BandLayout(H, world, 16)
    for rank in (0, world - 1):
        for S in (1, 2, 4, 8):
            streams = [torch.cuda.Stream() for _ in range(S)]
            bufs = [torch.zeros((lay.rows_per_rank, W, 4), device="cuda") for _ in range(S)]
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for k in range(64):
                    render_bands(sc, cam, W, lay, rank, bufs[k % S].data_ptr(), stream=streams[k % S].cuda_stream)
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 64 * 1e3)
            print(f"world {world} rank {rank} streams {S}: {best:.4f} ms per frame-share (HWQ={os.environ.get('GPU_MAX_HW_QUEUES','default')})", flush=True)

# the same share, G frames per launch (grid.y = frame), 2 launches in flight
for world in (2, 4, 8):
    lay = BandLayout(H, world, 16)
    for G in (4, 8):
        streams = [torch.cuda.Stream() for _ in range(2)]
        bufs = [torch.zeros((G, lay.rows_per_rank, W, 4), device="cuda") for _ in range(2)]
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(32):
                render_bands_batch(sc, [cam] * G, W, lay, 0, bufs[k % 2].data_ptr(), stream=streams[k % 2].cuda_stream)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / (32 * G) * 1e3)
        print(f"world {world} rank 0, {G} frames per launch, 2 launches in flight: {best:.4f} ms per frame-share", flush=True)
