"""Dev script (GPU): rank 0's expansion of sparse shares into frames, alone on the GPU: the grid-shaped kernel (k_expand_sparse2, frames
of at most 512 bands) against the per-pixel form it replaced (k_deinterleave_sparse2: still taken by frames of more bands, here forced
with 4-row bands).  usage: python scripts/expand_time.py"""
import sys, time
sys.path.insert(0, ".")
import torch
import sdfbox_amd as sb
from sdfbox_amd.tiles import BandLayout, deinterleave_sparse2, render_sparse2, sparse2_bytes

od = sb.dragon_standin(9); sc = sb.Scene(od)
for (W, H) in ((1920, 1080), (3840, 2160)):
    cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
    for world in (1, 8):
        for G in (1, 4):
            for band_rows, form in ((16, ""), (8, ""), (64, "")):
                lay = BandLayout(H, world, band_rows)
                full = lay.rows_per_rank * W * G
                share = torch.zeros(sparse2_bytes(lay.rows_per_rank, W, G, full), dtype=torch.uint8, device="cuda")
                frames = torch.zeros((G, H, W, 4), device="cuda")
                render_sparse2(sc, [cam] * G, W, lay, 0, share.data_ptr(), full, 0)
                torch.cuda.synchronize()
                best = 1e9
                for rep in range(5):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for k in range(20):
                        deinterleave_sparse2(0, [share.data_ptr()] * world, frames.data_ptr(), W, lay, full, frames=G)
                    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20)
                mb = G * H * W * 16 / 1e6
                nb = (H + band_rows - 1) // band_rows
                print(f"{W}x{H} world {world} G={G} bands of {band_rows} rows ({nb} bands: {'grid-shaped' if nb <= 512 else 'per-pixel'}): "
                      f"{best * 1e6 / G:.1f} us per frame, {mb / best / 1e6:.2f} TB/s written", flush=True)
