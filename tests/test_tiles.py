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
    for (H, G, B) in [(1080, 8, 16), (1080, 3, 16), (2160, 8, 16), (37, 4, 8), (16, 8, 16), (100, 1, 16), (64, 2, 24)]:
        lay = BandLayout(H, G, B)
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


def numpy_deinterleave(gathered, layout):
    """Mirror of k_deinterleave (sdfhip_device.hip) in numpy."""
    W = gathered.shape[2]
    frame = np.zeros((layout.height, W, 4), dtype=np.float32)
    for y in range(layout.height):
        r, l = layout.source_of(y)
        frame[y] = gathered[r, l]
    return frame


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_main(rank, world, port, W, H, band_rows, q):
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
    lay = BandLayout(H, world, band_rows)
    local = np.zeros((lay.rows_per_rank, W, 4), dtype=np.float32)
    # the oracle renders this rank's bands (the GPU kernel's stand-in on CPU)
    for lb, b in enumerate(lay.bands_of(rank)):
        y0 = b * band_rows
        n = min(band_rows, H - y0)
        img, _ = oracle.render(od.Structs, od.Values, cam.State, W, H, row0=y0, nrows=n)
        local[lb * band_rows: lb * band_rows + n] = img
    t = torch.from_numpy(local)
    glist = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, glist, dst=0)
    if rank == 0:
        frame = numpy_deinterleave(torch.stack(glist).numpy(), lay)
        full, _ = oracle.render(od.Structs, od.Values, cam.State, W, H)
        q.put(bool(((frame.view(np.uint32) == full.view(np.uint32)) | (np.isnan(frame) & np.isnan(full))).all()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,band_rows", [(2, 40, 8), (2, 37, 16)])
def test_two_rank_gather_reassembles_the_frame(world, H, band_rows):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, 48, H, band_rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
