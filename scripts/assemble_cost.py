"""Dev script: what the gather formats cost around the gather, 1080p shared by 8 ranks: a peer's
compaction of its shares (sparse only) and rank 0's assembly of the frame (us per frame)."""
import sys, time
sys.path.insert(0, ".")
import torch
import sdfbox_amd as sb
from sdfbox_amd.tiles import (BandLayout, deinterleave, deinterleave_sparse, render_bands_batch, sparse_count,
                              sparse_share_bytes, wire_compact, wire_shape)
W, H, G = 1920, 1080, 8
od = sb.dragon_standin(9); sc = sb.Scene(od)
cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
st = torch.cuda.current_stream().cuda_stream
for world in (2, 4, 8):
    lay = BandLayout(H, world, 16)
    R = lay.rows_per_rank
    dense = torch.zeros((world, G) + wire_shape(R, W), dtype=torch.uint8, device="cuda")
    for r in range(world):
        render_bands_batch(sc, [cam] * G, W, lay, r, dense[r].data_ptr(), flags=sb.FLAG_WIRE, stream=st)
    full = R * W
    probe = torch.zeros((world, G, sparse_share_bytes(R, W, full)), dtype=torch.uint8, device="cuda")
    for r in range(world):
        wire_compact(0, dense[r].data_ptr(), probe[r].data_ptr(), W, R, G, full, stream=st)
    counts, _ = sparse_count(probe, R, W, full)
    cap = (max(counts) * 5 // 4 + 1023) // 1024 * 1024
    nb = sparse_share_bytes(R, W, cap)
    sparse = torch.zeros((world, G, nb), dtype=torch.uint8, device="cuda")
    frames = torch.zeros((G, H, W, 4), device="cuda")

    def timed(fn, n=20):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n / G * 1e6
    t_compact = timed(lambda: wire_compact(0, dense[1].data_ptr(), sparse[1].data_ptr(), W, R, G, cap, stream=st))
    for r in range(world):
        wire_compact(0, dense[r].data_ptr(), sparse[r].data_ptr(), W, R, G, cap, stream=st)
    t_sparse = timed(lambda: deinterleave_sparse(0, sparse.data_ptr(), frames.data_ptr(), W, lay, cap, stream=st, frames=G))
    t_dense = timed(lambda: deinterleave(0, dense.data_ptr(), frames.data_ptr(), W, lay, stream=st, pixel_bytes=5, frames=G))
    print(f"world {world}: lit pixels per share max {max(counts)} of {R * W} -> capacity {cap}, {nb / (R * W):.2f} B/pixel on the wire (dense wire 5); "
          f"peer compaction {t_compact:.1f} us/frame; rank-0 assembly sparse {t_sparse:.1f} us/frame, dense wire {t_dense:.1f}", flush=True)
