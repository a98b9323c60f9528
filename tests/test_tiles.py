"""Multi-GPU host logic on CPU: band layout, the de-interleave index math, and
the N>1 exchange (gloo, world_size 2) with the oracle standing in for the
kernel so that it runs without a GPU."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import REPO, assert_frames_identical, make_camera


def test_band_layout_partitions_every_row_once():
    from sdfbox_amd.tiles import BandLayout
    for (H, G, B, w0) in [(1080, 8, 16, 1.0), (1080, 3, 16, 1.0), (2160, 8, 16, 1.0), (37, 4, 8, 1.0), (16, 8, 16, 1.0),
                          (100, 1, 16, 1.0), (64, 2, 24, 1.0),
                          (1080, 8, 16, 0.775), (1080, 8, 8, 0.775), (2160, 4, 8, 0.86), (1080, 2, 16, 0.93),
                          (37, 4, 8, 0.5), (16, 8, 16, 0.3), (100, 1, 16, 0.5)]:
        lay = BandLayout(H, G, B, w0)
        seen = np.zeros(H, dtype=np.int32)
        for r in range(G):
            rows = lay.rows_of(r)
            assert all(l < lay.rows_per_rank for l, _ in rows)
            assert len({l for l, _ in rows}) == len(rows)
            for l, y in rows:
                seen[y] += 1
                assert lay.source_of(y) == (r, l)
        assert (seen == 1).all()
        assert lay.rows_per_rank * G >= H


def test_weighted_layout_gives_rank0_less_and_spreads_every_share():
    from sdfbox_amd.tiles import BandLayout
    even = BandLayout(1080, 8, 8)
    assert not even.weighted and even.owner == [b % 8 for b in range(even.n_bands)]
    lay = BandLayout(1080, 8, 8, 0.775)
    n = [len(lay.bands_of(r)) for r in range(8)]
    assert lay.weighted and sum(n) == lay.n_bands == 135
    assert max(n[1:]) - min(n[1:]) <= 1                       # peers equal within one band
    assert abs(n[0] / (sum(n[1:]) / 7) - 0.775) < 0.06        # rank 0: the requested fraction of a peer's share
    assert lay.rows_per_rank == max(n) * 8
    for r in range(8):                                         # no share is a contiguous slab: gaps stay near world bands
        b = lay.bands_of(r)
        assert max(np.diff(b)) <= 2 * 8 + 2, (r, b)
    assert BandLayout(1080, 8, 8, 0.775).owner == lay.owner    # deterministic: every rank computes the same deal
    with pytest.raises(ValueError):
        BandLayout(1080, 8, 8, 0.0)
    with pytest.raises(ValueError):
        BandLayout(8192, 8, 8, 0.5)                            # 1024 bands > 512 in a weighted layout


def test_bands_dealt_by_cost():
    # tiles.balanced_owner: longest processing time first, rank 0 charged for the assembly, every rank at least one band and none
    # more than the cap; tiles.band_costs: a tile costs its longest pixel's steps + a fixed part, a band the sum of its tiles
    import torch
    from sdfbox_amd.tiles import BandLayout, balanced_owner, band_costs
    rng = np.random.default_rng(5)
    costs = [float(c) for c in rng.choice([12, 14, 40, 90, 300, 700], size=68)]
    total = sum(costs)
    own = balanced_owner(costs, 8, extra0=0.08 * total)
    lay = BandLayout(1080, 8, 16, owner=own)
    assert lay.weighted and lay.dealt_by_cost and lay.owner == own and sum(len(lay.bands_of(r)) for r in range(8)) == 68
    load = [sum(costs[b] for b in lay.bands_of(r)) for r in range(8)]
    assert max(load[1:]) - min(load[1:]) <= 0.03 * max(load[1:])          # the peers within 3 % of each other
    assert abs(load[0] + 0.08 * total - np.mean(load[1:])) <= 0.05 * np.mean(load[1:])     # rank 0's bands + its assembly = a peer's
    assert balanced_owner(costs, 8, extra0=0.08 * total) == own            # deterministic
    assert max(len(lay.bands_of(r)) for r in range(8)) <= 9 + 2 and lay.rows_per_rank == 16 * max(len(lay.bands_of(r)) for r in range(8))
    for y in range(1080):                                                  # the inverse the de-interleave uses
        r, l = lay.source_of(y)
        assert (l, y) in set(lay.rows_of(r)) if y % 97 == 0 else True
    # a rank 0 charged more than a share is worth still gets one band (its share may not be empty), and the cap binds
    own = balanced_owner([1.0] * 16 + [100.0] * 4, 4, extra0=1e6)
    assert own.count(0) == 1 and max(own.count(r) for r in range(4)) <= 5 + 2
    assert balanced_owner([5, 1, 1, 1], 4, extra0=100) == [1, 2, 3, 0]
    with pytest.raises(ValueError):
        balanced_owner([1.0] * 3, 4)
    with pytest.raises(ValueError):
        BandLayout(64, 2, 16, owner=[0, 0, 0, 0])                          # rank 1 would have no band
    with pytest.raises(ValueError):
        BandLayout(64, 2, 16, owner=[0, 1, 2, 0])
    # band_costs: alpha = steps; 16 x 24 frame = 2 x 3 tiles, band_rows 8 -> two bands
    a = torch.zeros(16, 24)
    a[0, 0] = 100.0; a[3, 9] = 7.0; a[9, 20] = 30.0; a[15, 23] = float("nan")
    assert band_costs(a, 8, fixed=4.0) == [100 + 7 + 0 + 12.0, 0 + 0 + 30 + 12.0]
    assert band_costs(a, 16, fixed=0.0) == [137.0]
    assert len(band_costs(torch.zeros(1080, 1920), 16)) == 68 and len(band_costs(torch.zeros(37, 50), 16)) == 3


def wire_pack(img):
    """numpy model of the kernel's wire buffer (FrameSink mode 3, raymarch_kernels.h): every pixel is
    (a, a, a, n), n <= 140, or the sky constant (0.005, 0.01, 0.2, n), n <= 100 -> a plane of floats
    (the bits of a) followed by a plane of bytes (n, or 255 - n for sky): uint8 [5][rows][W]."""
    rows, W = img.shape[:2]
    bits = np.ascontiguousarray(img).view(np.uint32)
    sky = bits[..., 0] != bits[..., 1]
    expect_sky = np.array([0.005, 0.01, 0.2], dtype=np.float32).view(np.uint32)
    assert (bits[sky][:, :3] == expect_sky).all(), "a non-grey pixel that is not the sky constant"
    assert (bits[~sky][:, 0] == bits[~sky][:, 2]).all()
    steps = img[..., 3].astype(np.uint32)
    assert (steps.astype(np.float32) == img[..., 3]).all()
    assert steps[sky].max(initial=0) <= 100 and steps[~sky].max(initial=0) <= 140
    out = np.empty((5, rows, W), dtype=np.uint8)
    out[:4].reshape(-1).view(np.uint32)[:] = np.where(sky, 0, bits[..., 0]).reshape(-1)
    out[4] = np.where(sky, 255 - steps, steps).astype(np.uint8)
    return out


def wire_expand(w):
    """[..., 5, rows, W] uint8 -> [..., rows, W, 4] float32"""
    lead, (rows, W) = w.shape[:-3], w.shape[-2:]
    flat = np.ascontiguousarray(w).reshape((-1, 5, rows, W))
    out = np.empty((flat.shape[0], rows, W, 4), dtype=np.uint32)
    for k in range(flat.shape[0]):
        a = flat[k, :4].reshape(-1).view(np.uint32).reshape(rows, W)
        code = flat[k, 4].astype(np.uint32)
        sky = code > 140
        out[k, ..., 0] = out[k, ..., 1] = out[k, ..., 2] = a
        out[k][sky, :3] = np.array([0.005, 0.01, 0.2], dtype=np.float32).view(np.uint32)
        out[k, ..., 3] = np.where(sky, 255 - code, code).astype(np.float32).view(np.uint32)
    return out.view(np.float32).reshape(lead + (rows, W, 4))


def test_wire_pixels_are_lossless_on_oracle_frames(oracle_mod, scenes):
    # the premise of the 5-byte gather format, checked on the oracle's own frames (NaN greys included)
    for name in ("sphere_d4", "torus_d6"):
        od = scenes[name]
        for camname in ("default", "rotated", "closeup"):
            cam = make_camera(camname, 96, 64)
            cam.State.light[0] = 0.9
            img, _ = oracle_mod.render(od.Structs, od.Values, cam.State, 96, 64)
            w = wire_pack(img)
            assert w.dtype == np.uint8 and w.shape == (5, 64, 96)
            assert (wire_expand(w).view(np.uint32) == img.view(np.uint32)).all(), (name, camname)


def numpy_deinterleave(gathered, layout):
    """Mirror of k_deinterleave (gather_kernels.h) in numpy."""
    W = gathered.shape[2]
    frame = np.zeros((layout.height, W, 4), dtype=np.float32)
    for y in range(layout.height):
        r, l = layout.source_of(y)
        frame[y] = gathered[r, l]
    return frame


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_main(rank, world, port, W, H, band_rows, q, rank0_weight=1.0, wire=False):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch
    import torch.distributed as dist
    import oracle
    import sdfbox_amd as sb
    from sdfbox_amd.tiles import BandLayout
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    od = sb.torus_d6()
    cam = make_camera("default", W, H)
    if rank0_weight == "cost":
        # the deal bench.py makes (measure_band_deal): rank 0 prices the bands from a rendered frame's step counts, charges itself
        # for the assembly, and every rank receives the owner list in a broadcast
        from sdfbox_amd.tiles import balanced_owner, band_costs
        owner = torch.zeros((H + band_rows - 1) // band_rows, dtype=torch.uint8)
        if rank == 0:
            full0, _ = oracle.render(od.Structs, od.Values, cam.State, W, H)
            costs = band_costs(torch.from_numpy(full0[..., 3].copy()), band_rows)
            owner = torch.tensor(balanced_owner(costs, world, extra0=0.2 * sum(costs)), dtype=torch.uint8)
        dist.broadcast(owner, src=0)
        lay = BandLayout(H, world, band_rows, owner=owner.tolist())
    else:
        lay = BandLayout(H, world, band_rows, rank0_weight)
    local = np.zeros((lay.rows_per_rank, W, 4), dtype=np.float32)
    # the oracle renders this rank's bands (the GPU kernel's stand-in on CPU)
    for lb, b in enumerate(lay.bands_of(rank)):
        y0 = b * band_rows
        n = min(band_rows, H - y0)
        img, _ = oracle.render(od.Structs, od.Values, cam.State, W, H, row0=y0, nrows=n)
        local[lb * band_rows: lb * band_rows + n] = img
    t = torch.from_numpy(wire_pack(local) if wire else local)
    glist = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, glist, dst=0)
    if rank == 0:
        got = torch.stack(glist).numpy()
        frame = numpy_deinterleave(wire_expand(got) if wire else got, lay)
        full, _ = oracle.render(od.Structs, od.Values, cam.State, W, H)
        q.put(bool(((frame.view(np.uint32) == full.view(np.uint32)) | (np.isnan(frame) & np.isnan(full))).all()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,band_rows,rank0_weight,wire", [(2, 40, 8, 1.0, False), (2, 37, 16, 1.0, False),
                                                                 (2, 56, 8, 0.6, True), (2, 56, 8, "cost", False),
                                                                 (3, 61, 8, "cost", True)])
def test_two_rank_gather_reassembles_the_frame(world, H, band_rows, rank0_weight, wire):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, 48, H, band_rows, q, rank0_weight, wire)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_sparse_share_size_arithmetic():
    # the byte layout of a sparse wire share (codes, per-tile masks and slot indices, header, floats), from the
    # host side of the library: sizes are 16-byte multiples, grow by 4 bytes per slot, and beat the dense 5 bytes
    # per pixel whenever at most ~90 % of the pixels are lit
    import sdfbox_amd.lab                           # round 2's format: the experiments build (include/sdfhip_experimental.h)
    sparse_share_bytes = sdfbox_amd.lab.load().tiles.sparse_share_bytes
    for rows, W in [(8, 8), (16, 61), (144, 1920), (1088, 3840)]:
        tiles = ((W + 7) // 8) * (rows // 8)
        up = lambda v: (v + 15) & ~15
        for cap in (0, 1, 1000, rows * W // 4, rows * W):
            nb = sparse_share_bytes(rows, W, cap)
            want = up(up(up(up(rows * W) + tiles * 8) + tiles * 4) + 16 + cap * 4)
            assert nb == want and nb % 16 == 0, (rows, W, cap, nb, want)
        assert sparse_share_bytes(rows, W, rows * W // 4) < 5 * rows * W / 2 or rows * W < 1024
        assert sparse_share_bytes(rows, W, int(rows * W * 0.9)) < 5 * rows * W + 64
