"""The laboratory's line-touch count (sdfhip_debug_touch_begin / _end: what scripts/compulsory_bytes.py and bench.py's
roofline.compulsory_bytes rest on) on the GPU: it changes no pixel, it is deterministic, and its numbers obey what distinct
lines must obey."""
import pytest

from conftest import assert_frames_identical

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lab():
    import sdfbox_amd.lab
    return sdfbox_amd.lab.load()


def _count(lab, scene, cam, W, H, flags=0, pt=None, rows=None):
    import torch
    buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    st = lab.Stats()
    s = torch.cuda.current_stream().cuda_stream
    if pt is not None:
        scene.DrawPathDevice(cam, W, H, buf.data_ptr(), pt=pt, flags=flags | lab.FLAG_COUNT, stream=s, stats=st)
    else:
        scene.DrawDevice(cam, W, H, buf.data_ptr(), flags=flags | lab.FLAG_COUNT, stream=s, stats=st)
    torch.cuda.synchronize()
    return buf.cpu().numpy(), st


def _cam(lab, W, H):
    cam = lab.Logic(W, H)
    cam.Position = (0.5, 0.5, -0.35)
    cam.Heading = (-0.2, 0.35)
    return cam


@pytest.mark.parametrize("split", [None, 5])
def test_touched_lines_of_a_frame(lab, split):
    # split None: the dense grid at the tree's depth (one array); 5: a split grid, coarse level 5 + blocks of two finer levels
    W, H = 640, 360
    od = lab.dragon_standin(7, nthreads=8)
    cam = _cam(lab, W, H)
    with lab.Scene(od, device=0, top_grid_split=split) as sc:
        plain, st0 = _count(lab, sc, cam, W, H)
        sc.touch_begin()
        a, st1 = _count(lab, sc, cam, W, H)
        b, _ = _count(lab, sc, cam, W, H)                       # a second frame: a second phase, the bitmaps were cleared in between
        c, _ = _count(lab, sc, cam, W, H, flags=lab.FLAG_COMPACT)
        t = sc.touch_end()
        assert_frames_identical(a, plain, "counting render with the touch bitmaps on")
        assert_frames_identical(b, plain, "second counted frame")
        assert_frames_identical(c, plain, "compact form")
        assert (st1.n_nodes, st1.n_samples, st1.n_loads) == (st0.n_nodes, st0.n_samples, st0.n_loads)
        ph = t["phases"]
        assert len(ph) == 3 and all(p["grid"] == "own" for p in ph)
        # deterministic chip-wide, and the same lookups whichever kernel form makes them
        assert (ph[0]["coarse_lines"], ph[0]["fine_lines"]) == (ph[1]["coarse_lines"], ph[1]["fine_lines"]) == (ph[2]["coarse_lines"], ph[2]["fine_lines"])
        ab = t["array_bytes"]
        for p in ph:
            for kind in ("coarse", "fine"):
                n, nx = p[f"{kind}_lines"], p[f"{kind}_lines_xcd_sum"]
                assert 0 <= n <= (ab[kind] + 127) // 128                         # no more lines than the array has
                assert n <= nx <= 8 * n                                          # every line is some XCD's; at most all eight's
            lines = p["coarse_lines"] + p["fine_lines"]
            assert 0 < lines <= st1.n_loads                                      # a line is touched by at least one lookup
        assert ab["coarse"] == 16 << (3 * sc.top_grid_level) and (ab["fine"] > 0) == (split is not None)
        assert (ph[0]["fine_lines"] > 0) == (split is not None)
        # outside begin / end nothing is counted and a second end is an error, not a crash
        with pytest.raises(lab.SdfHipError):
            sc.touch_end()
        # a sub-frame touches a subset: fewer (or as many) lines than the whole frame
        sc.touch_begin()
        import torch
        half = torch.zeros((120, W, 4), dtype=torch.float32, device="cuda")          # the first of three 120-row bands
        sc.DrawDevice(cam, W, H, half.data_ptr(), nrows_out=120, band_rows=120, band_first=0, band_stride=3, flags=lab.FLAG_COUNT,
                      stream=torch.cuda.current_stream().cuda_stream, stats=lab.Stats())
        torch.cuda.synchronize()
        th = sc.touch_end()["phases"]
        assert len(th) == 1 and th[0]["coarse_lines"] <= ph[0]["coarse_lines"] and th[0]["fine_lines"] <= ph[0]["fine_lines"]
        assert 0 < th[0]["coarse_lines"] + th[0]["fine_lines"]


def test_touched_lines_of_a_path_traced_frame_come_per_launch(lab):
    W, H = 160, 96
    od = lab.dragon_standin(7, nthreads=8)
    cam = _cam(lab, W, H)
    pt = lab.PathTrace(spp=4)
    with lab.Scene(od, device=0) as sc:
        plain, st0 = _count(lab, sc, cam, W, H, pt=pt)
        sc.touch_begin()
        a, st1 = _count(lab, sc, cam, W, H, pt=pt)
        t = sc.touch_end()
        assert_frames_identical(a, plain, "path-traced counting render with the touch bitmaps on")
        ph = t["phases"]
        # the camera segments, then one phase per bounce level (max_bounces + 1 launches of k_pt_bounce)
        assert len(ph) == 1 + pt.max_bounces + 1
        assert ph[0]["grid"] == "own" and ph[0]["coarse_lines"] > 0
        second = t["array_bytes"]["coarse2"] > 0
        assert all(p["grid"] == ("bounce" if second else "own") for p in ph[1:])
        assert ph[1]["coarse_lines"] + ph[1]["fine_lines"] > 0
        assert sum(p["coarse_lines"] + p["fine_lines"] for p in ph) <= st1.n_loads
        assert st1.n_loads == st0.n_loads and st1.n_hits == st0.n_hits
        # n_hits: the entries that went through the hit queues, all levels -- at least the vertices that cast a shadow ray, at most
        # every path once per level
        assert st1.n_shadow_rays <= st1.n_hits <= W * H * pt.spp * (pt.max_bounces + 1)


def test_the_product_has_no_touch_hook(sb):
    assert not hasattr(sb._lib.lib, "sdfhip_debug_touch_begin")
    with pytest.raises(RuntimeError):
        import sdfbox_amd
        od = sdfbox_amd.sphere_d4()
        with sdfbox_amd.Scene(od, device=0) as sc:
            sc.touch_begin()
