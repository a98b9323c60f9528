"""Dev script: what each rank of a sharded 1080p run does per frame -- peers render their share,
rank 0 renders its share and assembles the frame (de-interleave + wire expansion of every rank's
rows) -- for even and weighted band layouts.  No gather: rank 0's gathered buffer is just memory.
The slowest rank sets the frame time."""
import sys, time
sys.path.insert(0, ".")
import torch
import sdfbox_amd as sb
from sdfbox_amd.tiles import BandLayout, deinterleave, render_bands_batch, wire_shape
W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1920x1080").split("x"))
od = sb.dragon_standin(9); sc = sb.Scene(od)
cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
pb, fl = 5, sb.FLAG_WIRE
for world, weights in ((2, (1.0, 0.9, 0.8)), (4, (1.0, 0.8, 0.6, 0.5)), (8, (1.0, 0.5, 0.4, 0.3, 0.2))):
    G, S = 4, 4                      # bench.py's defaults: 4 frames per launch, 4 launches in flight
    for band_rows in (16,):
        for w0 in weights:
            lay = BandLayout(H, world, band_rows, w0)
            streams = [torch.cuda.Stream() for _ in range(S)]
            local = [torch.zeros((G,) + wire_shape(lay.rows_per_rank, W), dtype=torch.uint8, device="cuda") for _ in range(S)]
            gathered = [torch.zeros((world, G) + wire_shape(lay.rows_per_rank, W), dtype=torch.uint8, device="cuda") for _ in range(S)]
            frames = [torch.zeros((G, H, W, 4), device="cuda") for _ in range(S)]
            res = []
            for rank in sorted({0, 1, world - 1}):
                best = 1e9
                for rep in range(3):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for k in range(24):
                        s = streams[k % S].cuda_stream
                        render_bands_batch(sc, [cam] * G, W, lay, rank, local[k % S].data_ptr(), flags=fl, stream=s)
                        if rank == 0:
                            deinterleave(0, gathered[k % S].data_ptr(), frames[k % S].data_ptr(), W, lay, stream=s, pixel_bytes=pb, frames=G)
                    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / (24 * G) * 1e3)
                res.append((rank, best))
            print(f"{W}x{H} world {world} bands of {band_rows} weight {w0}: shares {[len(lay.bands_of(r)) for r in range(world)]} "
                  + " ".join(f"rank{r} {t:.4f}" for r, t in res) + f" -> frame {max(t for _, t in res):.4f} ms", flush=True)
