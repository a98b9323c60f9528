"""Dev script (GPU): what does launching a frame's expensive tiles first buy for the latency of ONE frame?
k_march writes every tile's march iterations; the next frame's workgroups take their tiles from a permutation that
keeps each XCD's set of tiles (blocks b, b+8, ... share an L2) but sorts it by descending cost of the previous frame.
    python scripts/ab_tile_order.py [WxH] [orbit step in degrees]"""
import ctypes, math, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import sdfbox_amd as sb
from sdfbox_amd._lib import lib, check
W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1920x1080").split("x"))
deg = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
od = sb.dragon_standin(9, nthreads=32); sc = sb.Scene(od)
def camera(k):
    phi = math.radians(k * deg); r = 0.85
    c = sb.Logic(W, H); c.Position = (0.5 - r * math.sin(phi), 0.5, 0.5 - r * math.cos(phi)); c.Heading = (-0.2, 0.35 + phi)
    return c
tx, ty = (W + 7) // 8, (H + 7) // 8
nblk = 8 * ((ty + 7) // 8) * tx
def default_perm():
    b = np.arange(nblk); xcd = b & 7; j = b >> 3; r = j // tx; cx = j - r * tx; row = r * 8 + xcd
    return np.where(row < ty, row * tx + cx, 0xFFFFFFFF).astype(np.uint32)
base = default_perm()
def pack(p):                      # what k_march reads: tile row << 16 | tile column (all ones: an idle workgroup), [XCD label][slot]
    e = np.where(p != 0xFFFFFFFF, (p // tx) << 16 | (p % tx), 0xFFFFFFFF).astype(np.uint32)
    return np.ascontiguousarray(e.reshape(-1, 8).T).reshape(-1)
buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
cost = torch.zeros(tx * ty, dtype=torch.int16, device="cuda")
perm = torch.from_numpy(base.astype(np.int64)).to(torch.int32).cuda() if False else torch.from_numpy(base.view(np.int32)).cuda()
def frame(cam):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sc.DrawDevice(cam, W, H, buf.data_ptr())
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
def sorted_perm(c):
    out = base.copy()
    for x in range(8):
        idx = np.nonzero((np.arange(nblk) & 7) == x)[0]
        tiles = base[idx]; real = tiles != 0xFFFFFFFF
        t = tiles[real]; order = np.argsort(-c[t].astype(np.int64), kind="stable")
        out[idx] = np.concatenate([t[order], tiles[~real]])
    return out
def long_first_perm(c, threshold):
    # only the few long tiles move: per XCD, the tiles whose wave ran >= threshold iterations in the previous frame are
    # launched first (in their old order), all others keep their order behind them
    out = base.copy()
    for x in range(8):
        idx = np.nonzero((np.arange(nblk) & 7) == x)[0]
        tiles = base[idx]; real = tiles != 0xFFFFFFFF
        t = tiles[real]; long = c[t] >= threshold
        out[idx] = np.concatenate([t[long], t[~long], tiles[~real]])
    return out
def mixed_perm(c, threshold, ratio):
    # the long tiles (>= threshold iterations), in descending order of cost, spread over the FRONT of the launch with `ratio` other
    # tiles (in their default order) between two of them; the rest behind: the long waves start early without all running at once
    out = base.copy()
    for x in range(8):
        idx = np.nonzero((np.arange(nblk) & 7) == x)[0]
        tiles = base[idx]; real = tiles != 0xFFFFFFFF
        t = tiles[real]; long = c[t] >= threshold
        H = t[long][np.argsort(-c[t[long]].astype(np.int64), kind="stable")]; L = t[~long]
        seq = []; li = 0
        for h in H:
            seq.append(h); seq.extend(L[li:li + ratio]); li += ratio
        seq.extend(L[li:])
        out[idx] = np.concatenate([np.array(seq, dtype=np.uint32), tiles[~real]])
    return out
def row_perm(c, key):
    # whole tile rows stay together (neighbouring tiles share cells); each XCD's rows k, k+8, ... are launched in
    # descending order of the previous frame's row cost
    out = base.copy()
    rows = c.reshape(ty, tx).astype(np.int64)
    rc = rows.max(1) if key == "max" else rows.sum(1)
    for x in range(8):
        myrows = np.arange(x, ty, 8)
        order = myrows[np.argsort(-rc[myrows], kind="stable")]
        idx = np.nonzero((np.arange(nblk) & 7) == x)[0]
        tiles = np.concatenate([r * tx + np.arange(tx) for r in order]).astype(np.uint32)
        out[idx] = np.concatenate([tiles, np.full(len(idx) - len(tiles), 0xFFFFFFFF, np.uint32)])
    return out
def dilate(c, R):
    from scipy.ndimage import maximum_filter
    return maximum_filter(c.reshape(ty, tx), size=2 * R + 1, mode="nearest").reshape(-1)
for mode in ("default order", "previous frame's cost, descending per XCD", "previous cost dilated 1", "previous cost dilated 2", "previous cost dilated 4",
             "same frame's cost (oracle for the idea)",
             "rows by previous frame's summed cost", "rows by previous frame's max cost",
             "long tiles first, threshold 60", "long tiles first, threshold 90", "long tiles first, threshold 110",
             "mixed 1:1 from 40", "mixed 1:2 from 40", "mixed 1:3 from 40", "mixed 1:2 from 60", "mixed 1:4 from 30"):
    times = []
    check(lib.sdfhip_debug_tile_order(sc._h, None, ctypes.c_void_p(cost.data_ptr())))
    frame(camera(0))
    for k in range(1, 61):
        cam = camera(k)
        if mode != "default order":
            if mode.startswith("same"):                 # render once to learn this frame's own cost
                check(lib.sdfhip_debug_tile_order(sc._h, None, ctypes.c_void_p(cost.data_ptr()))); frame(cam)
            c = cost.cpu().numpy().view(np.uint16)
            c = ((c & 0xFF) + (c >> 8)).astype(np.uint16)       # primary + shadow loop iterations of the tile's wave
            if "dilated" in mode:
                c = dilate(c, int(mode.split()[-1]))
            p = row_perm(c, "sum") if "summed" in mode else row_perm(c, "max") if "rows" in mode else \
                long_first_perm(c, int(mode.split()[-1])) if mode.startswith("long") else \
                mixed_perm(c, int(mode.split()[-1]), int(mode.split()[1].split(":")[1])) if mode.startswith("mixed") else sorted_perm(c)
            perm.copy_(torch.from_numpy(pack(p).view(np.int32)))
            check(lib.sdfhip_debug_tile_order(sc._h, ctypes.c_void_p(perm.data_ptr()), ctypes.c_void_p(cost.data_ptr())))
        times.append(frame(cam))
    ref_ok = True
    print(f"{W}x{H}, camera moving {deg} deg/frame, {mode:48s}: median {np.median(times[5:]):.4f} ms, min {min(times[5:]):.4f}")
# and the steady state with two frames in flight (bench.py's timed region), fixed camera: default order against the
# order by this camera's own cost
cam = camera(0)
check(lib.sdfhip_debug_tile_order(sc._h, None, ctypes.c_void_p(cost.data_ptr()))); frame(cam)
c = cost.cpu().numpy().view(np.uint16); c = ((c & 0xFF) + (c >> 8)).astype(np.uint16)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in streams]
for name, p in (("default order", None), ("by cost, descending per XCD", sorted_perm(c)), ("long tiles first, threshold 60", long_first_perm(c, 60)),
                ("mixed 1:2 from 40", mixed_perm(c, 40, 2))):
    if p is not None:
        perm.copy_(torch.from_numpy(pack(p).view(np.int32)))
    check(lib.sdfhip_debug_tile_order(sc._h, ctypes.c_void_p(perm.data_ptr()) if p is not None else None, None))
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 400
        for k in range(n):
            sc.DrawDevice(cam, W, H, bufs[k & 1].data_ptr(), stream=streams[k & 1].cuda_stream)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    print(f"{W}x{H}, two frames in flight, fixed camera, {name:40s}: {dt:.4f} ms per frame")
check(lib.sdfhip_debug_tile_order(sc._h, None, None))
