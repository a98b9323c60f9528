"""Dev script (GPU): what the ranks of an N-rank sharded run do on their GPUs, timed alone on this one.

A peer renders its bands of G frames per launch into a sparse share (sdfhip_render_sparse_device), `nbuf` launches in flight as
bench.py keeps them; rank 0 renders a smaller share (the weight bench.py searches at start-up) AND expands all N shares into the
frames (its own share stands in for the peers': the expansion reads as many bytes).  The larger of the two is what a frame costs the
pipeline before any byte travels.  Timed in the steady state (400 steps) and as the short burst the driver's scaling run times
(`--steps`, default 20, after a synchronisation), against the whole frame on one GPU under the same clock: t(1 GPU) / t(N ranks).
No gather: the links are what this cannot emulate.

usage: python scripts/rank_emulation.py [WxH] [--steps 20] [--sweep] [--order]
  --sweep   every (G, nbuf) of a small table instead of bench.py's own choice (tiles.group_plan)
  --order   the shares' tiles launched in the order of their cost in the last launch (SDFHIP_FLAG_TILE_ORDER on a batch)
  --deal    cost (default): the bands dealt by measured cost, rank 0 charged for the assembly; weight: rounds 2-4's credit deal"""
import argparse, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ".")
import torch
import sdfbox_amd as sb
from sdfbox_amd.tiles import BandLayout, SparseExpandCall, SparseShareCall, balanced_owner, band_costs, group_plan, sparse2_bytes

ap = argparse.ArgumentParser()
ap.add_argument("size", nargs="?", default="1920x1080")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--sweep", action="store_true")
ap.add_argument("--order", action="store_true")
ap.add_argument("--worlds", default="2,4,8")
ap.add_argument("--deal", default="cost", choices=["cost", "weight"])
ap.add_argument("--verbose", action="store_true")
ap.add_argument("--no-readback", action="store_true", help="experiment: without the share counter's pinned copy + event per group")
ap.add_argument("--even", action="store_true", help="experiment: the even round-robin deal only (rank 0 a whole share)")
ap.add_argument("--fixed", type=float, default=4.0, help="band_costs: a tile's fixed part, in iterations")
ap.add_argument("--band-order", action="store_true", help="experiment: a share's bands in descending order of their cost (the expansion scrambles rows: timing only)")
args = ap.parse_args()
W, H = (int(v) for v in args.size.split("x"))
STEPS = args.steps
FLAGS = sb.FLAG_TILE_ORDER if args.order else 0
od = sb.dragon_standin(9); sc = sb.Scene(od)
cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
MAXBUF = 8
streams = [torch.cuda.Stream() for _ in range(MAXBUF)]


def best(fn, n, reps):
    return min(fn(n) for _ in range(reps))


# the whole frame on one GPU, four frames in flight (the N = 1 line's steady state and its burst)
bufs = [torch.zeros((H, W, 4), device="cuda") for _ in range(4)]
def whole(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        sc.DrawDevice(cam, W, H, bufs[k % 4].data_ptr(), stream=streams[k % 4].cuda_stream)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
whole(40)
t1 = best(whole, 400, 3); t1_b = best(whole, STEPS, 7)
print(f"{W}x{H} whole frame, 4 in flight: {t1:.4f} ms steady, {t1_b:.4f} ms per frame in a {STEPS}-step burst", flush=True)
del bufs


# the cost of every band, as bench.py prices them before its deal (tiles.band_costs on the frame's step counts)
_whole = torch.zeros((H, W, 4), device="cuda")
sc.DrawDevice(cam, W, H, _whole.data_ptr(), stream=torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
COSTS = band_costs(_whole[..., 3], 16, fixed=args.fixed); TOTAL = sum(COSTS)
del _whole


def measure(world, G, nbuf, lay):
    """(rank 0 steady, slowest peer steady, rank 0 burst, slowest peer burst) in ms per frame for this layout; rank 0 renders its
    share and expands `world` shares, EVERY peer's share is timed (the run is as slow as its slowest rank)"""
    full = lay.rows_per_rank * W * G
    shares = [torch.zeros(sparse2_bytes(lay.rows_per_rank, W, G, full), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
    frames = torch.zeros((G, H, W, 4), device="cuda")
    base = [0] * nbuf
    own = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(nbuf)]      # the share's counter on its way back, as bench.py
    evs = [torch.cuda.Event() for _ in range(nbuf)]                                  # keeps it: pinned copy + event on the slot's stream

    expand_call = SparseExpandCall(0, W, lay, full)
    calls = {r: SparseShareCall(sc, W, lay, r, full, max_frames=G, flags=FLAGS,
                                bands=sorted(lay.bands_of(r), key=lambda b: (-COSTS[b], b)) if args.band_order else None) for r in range(world)}
    ptr = [sh.data_ptr() for sh in shares]
    ptrs = [[p] * world for p in ptr]
    own_src = [sh[:4].view(torch.int32) for sh in shares]
    groups = {g: [cam] * g for g in range(1, G + 1)}
    frames_ptr = frames.data_ptr()
    host = [0.0, 0]

    def job(rank, nframes, expand):
        call = calls[rank]
        for s_ in range(nbuf):
            shares[s_][:4].zero_(); base[s_] = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        k = 0
        while k < nframes:
            g = min(G, nframes - k); slot = (k // G) % nbuf
            if k >= G * nbuf and not args.no_readback:   # the slot's previous group is complete by now: its counter has arrived (finish() in bench.py)
                evs[slot].synchronize(); base[slot] = int(own[slot].item()) & 0xFFFFFFFF
            h0 = time.perf_counter()
            call(groups[g], ptr[slot], base[slot], stream=streams[slot].cuda_stream)
            if expand:
                expand_call(slot, ptrs[slot], frames_ptr, frames=g, stream=streams[slot].cuda_stream)
            host[0] += time.perf_counter() - h0; host[1] += 1
            if not args.no_readback:
                with torch.cuda.stream(streams[slot]):
                    own[slot].copy_(own_src[slot], non_blocking=True)
                    evs[slot].record(streams[slot])
            k += g
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / nframes * 1e3
    warm = 2 * G * nbuf
    job(0, warm, True)
    r0, r0b = best(lambda n: job(0, n, True), 400, 2), best(lambda n: job(0, n, True), STEPS, 7)
    p, pb = 0.0, 0.0
    per = []
    for r in range(1, world):
        job(r, warm, False)                       # (and, with --order, this rank's tile order on every stream)
        a, b = best(lambda n: job(r, n, False), 400, 2), best(lambda n: job(r, n, False), STEPS, 7)
        per.append((r, a, b, sum(COSTS[q] for q in lay.bands_of(r)) / TOTAL, len(lay.bands_of(r))))
        p = max(p, a); pb = max(pb, b)
    measure.per = "; ".join(f"rank {r}: {a * 1e3:.1f} / {b * 1e3:.1f} us, {c * 100:.1f} % of the cost in {n} bands" for r, a, b, c, n in per)
    del shares, frames
    measure.host_us = host[0] / max(1, host[1]) * 1e6          # host time of a group's launch call(s)
    return r0, p, r0b, pb


for world in (int(w) for w in args.worlds.split(",")):
    plans = [(4, 4), (8, 4), (5, 4), (7, 3), (4, 5), (4, 8), (2, 8), (8, 3), (6, 4), (3, 7)] if args.sweep else [group_plan(world, STEPS)[:2]]
    for G, nbuf in plans:
        if G > 8 or nbuf > MAXBUF:       # (MAX_BATCH frames per launch)
            continue
        rows = []
        if args.deal == "cost":
            for frac in (0.0, 0.04, 0.07, 0.10, 0.13, 0.16, 0.20):
                lay = BandLayout(H, world, 16, owner=balanced_owner(COSTS, world, extra0=frac * TOTAL))
                r0, p1, r0b, p1b = measure(world, G, nbuf, lay)
                rows.append((max(r0b, p1b), f"assembly charged {frac:.2f} of the frame ({len(lay.bands_of(0))}/{lay.n_bands} bands)", r0, p1, r0b, p1b))
                if r0b <= p1b and r0 <= p1 and frac > 0:
                    break
        else:
            for weight in ((1.0,) if args.even else (1.0, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4)):
                r0, p1, r0b, p1b = measure(world, G, nbuf, BandLayout(H, world, 16, weight))
                rows.append((max(r0b, p1b), f"{weight:.1f} of a peer's share", r0, p1, r0b, p1b))
                if r0b <= p1b and r0 <= p1:
                    break
        _, what, r0, p1, r0b, p1b = min(rows)
        st = min(max(r[2], r[3]) for r in rows)
        print(f"world {world} G={G} nbuf={nbuf}{' ordered' if args.order else ''} deal={args.deal}{' bands by cost' if args.band_order else ''}: rank 0 {what}, expands: steady {r0:.4f} / slowest peer {p1:.4f} ms per frame "
              f"-> {t1 / max(r0, p1):.2f}x of {world} (best steady over the deals tried {t1 / st:.2f}x); {STEPS}-step burst {r0b:.4f} / {p1b:.4f} -> {t1_b / max(r0b, p1b):.2f}x  [{measure.host_us:.1f} us of host time per launch call]", flush=True)
        if args.verbose:
            print("    peers (steady / burst): " + measure.per, flush=True)
