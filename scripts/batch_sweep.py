"""Dev script: frames per launch (G) x launches in flight (S) for one rank's share of a 1080p frame
(world = 1, 2, 4, 8; render only, wire pixels), ms per frame."""
import sys, time
sys.path.insert(0, ".")
import torch
import sdfbox_amd as sb
from sdfbox_amd.tiles import BandLayout, render_bands_batch, wire_shape
W, H = 1920, 1080
od = sb.dragon_standin(9); sc = sb.Scene(od)
cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
for world in (1, 2, 4, 8):
    lay = BandLayout(H, world, 16)
    for G in (1, 2, 4, 8):
        row = []
        for S in (2, 3, 4, 8):
            streams = [torch.cuda.Stream() for _ in range(S)]
            bufs = [torch.zeros((G,) + wire_shape(lay.rows_per_rank, W), dtype=torch.uint8, device="cuda") for _ in range(S)]
            n = max(8, 256 // G)
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for k in range(n):
                    render_bands_batch(sc, [cam] * G, W, lay, 0, bufs[k % S].data_ptr(), flags=sb.FLAG_WIRE, stream=streams[k % S].cuda_stream)
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / (n * G) * 1e3)
            row.append(f"S={S}: {best:.4f}")
        print(f"world {world} G={G}: " + "  ".join(row), flush=True)
