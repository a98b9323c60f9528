"""Dev script (GPU): what ONE rank of an N-rank sharded run does on its GPU, timed alone on this one -- its bands of G frames per
launch into a sparse share (sdfhip_render_sparse_device), four launches in flight as bench.py keeps them -- in the steady state
and as the short burst the driver's scaling run times (20 steps after a synchronisation).  No gather, no expansion: an upper
bound of the scaling the march itself allows, t(1 rank, whole frame) / t(rank's share).
usage: python scripts/rank_emulation.py [WxH]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ".")
import torch
import sdfbox_amd as sb
from sdfbox_amd.tiles import BandLayout, deinterleave_sparse2, render_sparse2, sparse2_bytes

W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1920x1080").split("x"))
od = sb.dragon_standin(9); sc = sb.Scene(od)
cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
NBUF = 4
streams = [torch.cuda.Stream() for _ in range(NBUF)]

# the whole frame on one GPU, four frames in flight (the N = 1 line's steady state and its 20-step burst)
bufs = [torch.zeros((H, W, 4), device="cuda") for _ in range(NBUF)]
def whole(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        sc.DrawDevice(cam, W, H, bufs[k % NBUF].data_ptr(), stream=streams[k % NBUF].cuda_stream)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
whole(40)
t1 = min(whole(400) for _ in range(3)); t1_20 = min(whole(20) for _ in range(5))
print(f"{W}x{H} whole frame, 4 in flight: {t1:.4f} ms steady, {t1_20:.4f} ms per frame in a 20-step burst", flush=True)
del bufs

for world in (2, 4, 8):
    for G in ((4, 8) if world == 8 else (4,)):
        lay = BandLayout(H, world, 16)
        full = lay.rows_per_rank * W * G
        shares = [torch.zeros(sparse2_bytes(lay.rows_per_rank, W, G, full), dtype=torch.uint8, device="cuda") for _ in range(NBUF)]
        base = [0] * NBUF
        for rank in (0, world - 1):
            def run(nframes):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                k = 0
                while k < nframes:
                    g = min(G, nframes - k); slot = (k // G) % NBUF
                    render_sparse2(sc, [cam] * g, W, lay, rank, shares[slot].data_ptr(), full, base[slot], stream=streams[slot].cuda_stream)
                    k += g
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / nframes * 1e3
                for s in range(NBUF):
                    base[s] = int(shares[s][:4].view(torch.int32).item()) & 0xFFFFFFFF
                return dt
            run(40)
            steady = min(run(400) for _ in range(3)); burst = min(run(20) for _ in range(5))
            print(f"world {world} G={G} rank {rank}: share {steady:.4f} ms steady ({t1 / steady:.2f}x of {world}), "
                  f"{burst:.4f} ms per frame in a 20-step burst ({t1_20 / burst:.2f}x)", flush=True)
        del shares

# ... and rank 0's whole job: its (smaller, weighted) share AND the expansion of all N shares into the frames (its own share stands in
# for the peers': the expansion reads as many bytes) -- against a peer's share, by the weight of rank 0's share.  The larger of the
# two is what a frame costs the pipeline before any byte travels.
for world in (2, 4, 8):
    G = 8 if world == 8 else 4
    frames = torch.zeros((G, H, W, 4), device="cuda")
    rows = []
    for weight in (1.0, 0.9, 0.8, 0.7, 0.6, 0.5):
        lay = BandLayout(H, world, 16, weight)
        full = lay.rows_per_rank * W * G
        shares = [torch.zeros(sparse2_bytes(lay.rows_per_rank, W, G, full), dtype=torch.uint8, device="cuda") for _ in range(NBUF)]
        base = [0] * NBUF

        def job(rank, nframes, expand):
            for s_ in range(NBUF):
                shares[s_][:4].zero_(); base[s_] = 0
            torch.cuda.synchronize(); t0 = time.perf_counter()
            k = 0
            while k < nframes:
                g = min(G, nframes - k); slot = (k // G) % NBUF
                if k >= G * NBUF:            # the slot's counter runs on: read it back as bench.py does (its previous group is complete by now)
                    streams[slot].synchronize(); base[slot] = int(shares[slot][:4].view(torch.int32).item()) & 0xFFFFFFFF
                render_sparse2(sc, [cam] * g, W, lay, rank, shares[slot].data_ptr(), full, base[slot], stream=streams[slot].cuda_stream)
                if expand:
                    deinterleave_sparse2(0, [shares[slot].data_ptr()] * world, frames.data_ptr(), W, lay, full, frames=g, stream=streams[slot].cuda_stream)
                k += g
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / nframes * 1e3
        job(0, 40, True)
        r0 = min(job(0, 400, True) for _ in range(3)); r0b = min(job(0, 20, True) for _ in range(5))
        p1 = min(job(1, 400, False) for _ in range(3)); p1b = min(job(1, 20, False) for _ in range(5))
        rows.append((max(r0, p1), weight, r0, p1, r0b, p1b))
        del shares
        if r0 <= p1:
            break
    best, weight, r0, p1, r0b, p1b = min(rows)
    print(f"world {world} G={G}: rank 0 renders {weight:.1f} of a peer's share and expands: {r0:.4f} ms per frame, a peer {p1:.4f} -> {t1 / best:.2f}x of {world} steady; "
          f"20-step burst {r0b:.4f} / {p1b:.4f} -> {t1_20 / max(r0b, p1b):.2f}x", flush=True)
