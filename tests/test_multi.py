"""One frame over several devices behind one call (sdfhip_multi_*): the sparse shares the march kernel
writes itself, the band deal, the gather into devices[0] (every code path but the link: the device list names one GPU
several times), the float tail that is sent again, groups of frames in flight, the display pass at assembly, the
path-traced mode -- every assembled frame bit for bit the frame one device renders (and, on the small scenes, the
oracle's)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, assert_frames_identical, make_camera


def test_multi_create_fails_loudly_without_a_gpu(sb):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(sb.SdfHipError) as e:
        sb.MultiScene(sb.sphere_d4(), [0, 0])
    assert e.value.code == sb._lib.ERR_DEVICE


def test_multi_argument_errors_are_codes(sb):
    L = sb._lib
    h = ctypes.c_void_p()
    od = sb.sphere_d4()
    assert L.lib.sdfhip_multi_create(None, 0, od.Structs.ctypes.data, od.Values.ctypes.data, od.Length, ctypes.byref(h)) == L.ERR_ARG
    assert L.lib.sdfhip_multi_submit(None, 0, None, 1, 8, 8, 0, None) == L.ERR_ARG
    assert L.lib.sdfhip_multi_wait(None, 0, None, None) == L.ERR_ARG
    assert L.lib.sdfhip_multi_free(None) == L.OK


def test_band_deal_of_the_library_is_the_python_layout(sb):
    # the library deals the bands itself (sdfhip_multi.hip deal_bands); tiles.BandLayout is the same rule, used by the
    # torch.distributed path: sizes of the sparse shares agree for the layouts both can produce
    BandLayout = sb.tiles.BandLayout
    lay = BandLayout(2160, 8, 16, 1.0)
    assert lay.rows_per_rank == 272 and sorted(lay.bands_of(3)) == list(range(3, 135, 8))
    L = sb._lib.lib
    full = lay.rows_per_rank * 3840 * 4
    assert L.sdfhip_sparse2_bytes(3840, lay.rows_per_rank, 4, full) > L.sdfhip_sparse2_floats_offset(3840, lay.rows_per_rank, 4) > 0
    # fixed part: 64 + frames * tiles * (8 + 4 + 64) bytes, rounded to 64
    tiles = (3840 // 8) * (272 // 8)
    assert L.sdfhip_sparse2_floats_offset(3840, 272, 4) == ((64 + 4 * tiles * 12 + 63) // 64) * 64 + 4 * tiles * 64


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module", params=["product", "lab"])
def sb(request):
    # every test of this file runs on both flavours of the library (the experiments build adds one test hook here:
    # sdfhip_multi_debug_floats_sent)
    import sdfbox_amd
    if request.param == "product":
        return sdfbox_amd
    import sdfbox_amd.lab
    return sdfbox_amd.lab.load()


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    return torch


def whole_frame(sb, torch, scene, cam, W, H, flags=0):
    display = bool(flags & (sb.FLAG_DISPLAY | sb.FLAG_DISPLAY_DEBUG))
    buf = torch.zeros((H, W) if display else (H, W, 4), dtype=torch.int32 if display else torch.float32, device="cuda")
    scene.DrawDevice(cam, W, H, buf.data_ptr(), flags=flags, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return buf


@pytest.mark.gpu
@pytest.mark.parametrize("world,weight", [(1, 1.0), (3, 1.0), (4, 0.6)])
def test_sparse_share_written_by_the_march_kernel_expands_to_the_frame(sb, torch_mod, scenes, world, weight):
    # sdfhip_render_sparse_device + sdfhip_deinterleave_sparse2_device, the ranks played by one device: three cameras in one
    # launch (a group), ragged frame size, even and weighted band deals; RGBA32F and both display modes
    torch = torch_mod
    BandLayout = sb.tiles.BandLayout
    L = sb._lib.lib
    W, H, G = 333, 211, 3
    od = scenes["torus_d6"]
    cams = [make_camera(n, W, H) for n in ("rotated", "closeup", "default")]
    infos = (sb.Info * G)(*[c.State for c in cams])
    lay = BandLayout(H, world, 16, weight)
    cap = lay.rows_per_rank * W * G
    nbytes = L.sdfhip_sparse2_bytes(W, lay.rows_per_rank, G, cap)
    st = torch.cuda.current_stream().cuda_stream
    with sb.Scene(od) as scene:
        shares = [torch.full((nbytes,), 0xA5, dtype=torch.uint8, device="cuda") for _ in range(world)]      # garbage: the render must define all it reads
        for r in range(world):
            bands = (ctypes.c_uint16 * len(lay.bands_of(r)))(*lay.bands_of(r))
            # (the share's counter is never zeroed by the library: it counts on from the value it is told -- here the garbage's)
            sb._lib.check(L.sdfhip_render_sparse_device(scene._h, infos, G, W, H, lay.band_rows, bands, len(bands), lay.rows_per_rank, cap,
                                                        0xA5A5A5A5, 0, ctypes.c_void_p(shares[r].data_ptr()), ctypes.c_void_p(st)))
        ptrs = (ctypes.c_void_p * world)(*[s.data_ptr() for s in shares])
        owner = (ctypes.c_uint8 * lay.n_bands)(*lay.owner) if lay.weighted else None
        for flags in (0, sb.FLAG_DISPLAY, sb.FLAG_DISPLAY_DEBUG):
            out = torch.zeros((G, H, W, 4) if flags == 0 else (G, H, W), dtype=torch.float32 if flags == 0 else torch.int32, device="cuda")
            counts = torch.zeros(world, dtype=torch.int32, device="cuda")
            sb._lib.check(L.sdfhip_deinterleave_sparse2_device(0, ptrs, ctypes.c_void_p(out.data_ptr()), W, H, lay.band_rows, world,
                                                               lay.rows_per_rank, owner, cap, G, flags, -1, ctypes.c_void_p(counts.data_ptr()),
                                                               ctypes.c_void_p(st)))
            torch.cuda.synchronize()
            for f in range(G):
                ref = whole_frame(sb, torch, scene, cams[f], W, H, flags)
                assert torch.equal(out[f].view(torch.int32), ref.view(torch.int32)), f"world {world} flags {flags:#x} frame {f}"
        # the header counts the lit pixels of the share (all frames)
        lit = 0
        for f in range(G):
            fr = whole_frame(sb, torch, scene, cams[f], W, H)
            sky = (fr[..., 0] == 0.005) & (fr[..., 1] == 0.01) & (fr[..., 2] == 0.2)
            lit += int(((fr[..., 0].view(torch.int32) != 0) & ~sky).sum())
        used = sum((int(s[:4].view(torch.int32).item()) - 0xA5A5A5A5) % (1 << 32) for s in shares)
        assert used == lit                    # float slots = pixels whose grey level has any bit set
        assert [int(c) % (1 << 32) for c in counts.tolist()] == [int(s[:4].view(torch.int32).item()) % (1 << 32) for s in shares]


@pytest.mark.gpu
@pytest.mark.parametrize("world,weight", [(2, 1.0), (4, 0.6)])
def test_sparse_shares_of_a_batch_launched_in_tile_order(sb, torch_mod, scenes, world, weight):
    # SDFHIP_FLAG_TILE_ORDER on the batched launch of a gather group (round 5: a rank's short burst ends with its longest waves, so
    # the group's expensive tiles -- of ALL its frames -- go first): launch after launch on one stream, the order made from the last
    # frame of the launch before -- same cameras again, other cameras (a stale order), a partial last group (fewer frames: another
    # share layout), a second stream with its own order.  Every expanded frame must equal the whole-frame render; the shares start
    # as garbage, so a tile no workgroup took would show.
    torch = torch_mod
    BandLayout = sb.tiles.BandLayout
    T = sb.tiles
    W, H, G = 333, 211, 4
    od = scenes["torus_d6"]
    names = ("rotated", "closeup", "default", "rotated")
    lay = BandLayout(H, world, 16, weight)
    cap = lay.rows_per_rank * W * G
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    with sb.Scene(od) as scene:
        refs = {n: whole_frame(sb, torch, scene, make_camera(n, W, H), W, H) for n in set(names)}
        groups = (names, names, names[::-1], names[:3], names[1:2], names, names[:2], names)
        stream_of = lambda k: streams[k % 3 == 2]              # (launches 2 and 5 on the second stream)
        shares = {(k, r): torch.full((T.sparse2_bytes(lay.rows_per_rank, W, len(group), cap),), 0xA5, dtype=torch.uint8, device="cuda")
                  for k, group in enumerate(groups) for r in range(world)}
        torch.cuda.synchronize()                               # (the fills run on torch's stream, the renders on their own)
        for r in range(world):                                 # a rank's launches one after the other, as on its own GPU: the order
            for k, group in enumerate(groups):                 # a launch uses is the one made behind the same rank's launch before it
                cams = [make_camera(n, W, H) for n in group]
                T.render_sparse2(scene, cams, W, lay, r, shares[k, r].data_ptr(), cap, 0xA5A5A5A5, flags=sb.FLAG_TILE_ORDER,
                                 stream=stream_of(k).cuda_stream)
        torch.cuda.synchronize()
        for k, group in enumerate(groups):
            n = len(group)
            out = torch.full((n, H, W, 4), float("nan"), dtype=torch.float32, device="cuda")
            T.deinterleave_sparse2(0, [shares[k, r].data_ptr() for r in range(world)], out.data_ptr(), W, lay, cap, frames=n)
            torch.cuda.synchronize()
            for f in range(n):
                assert torch.equal(out[f].view(torch.int32), refs[group[f]].view(torch.int32)), f"world {world} launch {k} frame {f}"


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0, 0], [0] * 7])
def test_multi_frame_equals_the_oracle_and_one_device(sb, oracle_mod, scenes, devices):
    for sname, cname, (W, H) in (("sphere_d4", "default", (256, 256)), ("torus_d6", "rotated", (200, 120)), ("torus_d6", "closeup", (129, 65))):
        od = scenes[sname]
        cam = make_camera(cname, W, H)
        ref, _ = oracle_mod.render(od.Structs, od.Values, cam.State, W, H, nthreads=8)
        with sb.MultiScene(od, devices) as ms:
            img, st = ms.Draw(cam, W, H, want_stats=True)
            assert_frames_identical(img, ref, f"{sname}/{cname} over {devices}")
            assert st.n_devices == len(devices) and st.resends == 0
            assert ms.transport == "peer"
            # again (the float estimate is in use now), and with the tile-order flag
            assert_frames_identical(ms.Draw(cam, W, H), ref, "second frame")
            assert_frames_identical(ms.Draw(cam, W, H, flags=sb.FLAG_TILE_ORDER), ref, "tile order")
            assert_frames_identical(ms.Draw(cam, W, H, flags=sb.FLAG_TILE_ORDER), ref, "tile order, second frame")
            # other band heights and a lighter rank 0
            ms.configure(band_rows=8, rank0_weight=0.5)
            assert_frames_identical(ms.Draw(cam, W, H), ref, "8-row bands, rank 0 at half a share")
            ms.configure(band_rows=64, rank0_weight=1.0)
            assert_frames_identical(ms.Draw(cam, W, H), ref, "64-row bands")


@pytest.mark.gpu
def test_multi_selftest_reports_every_link(sb, scenes):
    # sdfhip_multi_selftest (create runs it; callable again): per device its PCI bus id, peer access into devices[0], and a 1 MB
    # pattern pushed the gather's way and read back from devices[0]
    with sb.MultiScene(scenes["sphere_d4"], [0, 0, 0]) as ms:
        links = ms.selftest()
        assert [l["device"] for l in links] == [0, 0, 0] and all(l["ok"] for l in links)
        assert len({l["pci_bus_id"] for l in links}) == 1 and links[0]["pci_bus_id"] == sb.device_pci_bus_id(0) and ":" in links[0]["pci_bus_id"]
        assert all(l["peer_access"] == -1 for l in links)                  # the same device: nothing to reach
        assert links[1]["push_ms"] > 0 and links[0]["push_ms"] == 0
        ms.Submit(0, make_camera("default", 64, 64), 64, 64)
        with pytest.raises(sb.SdfHipError):                                # not while a slot is in flight
            ms.selftest()
        ms.Wait(0)
        assert all(l["ok"] for l in ms.selftest())


@pytest.mark.gpu
def test_multi_display_pass_and_path_traced_mode(sb, oracle_mod, scenes):
    od = scenes["torus_d6"]
    W, H = 160, 96
    cam = make_camera("rotated", W, H)
    with sb.Scene(od) as one, sb.MultiScene(od, [0, 0, 0]) as ms:
        for dbg in (False, True):
            ref = one.DrawDisplay(cam, W, H, debug=dbg)
            got = ms.Draw(cam, W, H, flags=sb.FLAG_DISPLAY_DEBUG if dbg else sb.FLAG_DISPLAY)
            assert got.dtype == np.uint8 and (got == ref).all(), "display pass at assembly == display pass in the kernel"
        pt = sb.PathTrace(spp=4)
        ref, _ = oracle_mod.render_pt(od.Structs, od.Values, cam.State, W, H, spp=pt.spp, max_bounces=pt.max_bounces, seed=pt.seed,
                                      albedo=pt.albedo, nthreads=8)
        assert_frames_identical(ms.Draw(cam, W, H, pt=pt), ref, "path-traced frame over three ranks against the oracle")
        ms.configure(band_rows=8, rank0_weight=0.7)
        assert_frames_identical(ms.Draw(cam, W, H, pt=pt), ref, "path-traced frame, weighted deal")


@pytest.mark.gpu
def test_multi_path_traced_frame_then_ordinary_frames_on_one_handle(sb, oracle_mod, scenes):
    # A viewer that toggles path tracing: sdfhip_multi_render_path, then sdfhip_multi_render on the same slot.  The dense bands of the
    # path-traced frame once went into the ranks' sparse-share buffers and overwrote the shares' running slot counters: the next
    # ordinary frame lost every lit pixel of the ranks > 0.  Both kinds of share have their own buffers now.
    od = scenes["torus_d6"]
    W, H = 160, 96
    cam = make_camera("rotated", W, H)
    ref, _ = oracle_mod.render(od.Structs, od.Values, cam.State, W, H, nthreads=8)
    pt = sb.PathTrace(spp=2)
    ref_pt, _ = oracle_mod.render_pt(od.Structs, od.Values, cam.State, W, H, spp=pt.spp, max_bounces=pt.max_bounces, seed=pt.seed,
                                     albedo=pt.albedo, nthreads=8)
    for devices in ([0, 0], [0, 0, 0]):
        with sb.MultiScene(od, devices) as ms:
            assert_frames_identical(ms.Draw(cam, W, H), ref, "ordinary frame first")            # (the shares' counters have run on)
            for k in range(2):
                assert_frames_identical(ms.Draw(cam, W, H, pt=pt), ref_pt, f"path-traced frame {k}")
                img, st = ms.Draw(cam, W, H, want_stats=True)
                assert_frames_identical(img, ref, f"ordinary frame right after a path-traced one ({k})")
                assert st.resends == 0
                assert_frames_identical(ms.Draw(cam, W, H), ref, f"the frame after that ({k})")
            # the same through the slots: a path-traced submission between two groups on one slot
            ms.Submit(1, [cam, cam], W, H); ms.Wait(1)
            ms.Submit(1, cam, W, H, pt=pt); ms.Wait(1)
            ms.Submit(1, [cam, cam], W, H)
            ptr, st = ms.Wait(1, want_stats=True)
            assert st.resends == 0
            assert_frames_identical(ms.Draw(cam, W, H), ref, "after the slots")
        with pytest.raises(ValueError):
            with sb.MultiScene(od, devices) as ms:
                ms.Draw(cam, W, H, out=np.empty((H, W, 3), dtype=np.float32))                    # a wrong array is refused, not overrun


@pytest.mark.gpu
def test_multi_scene_without_a_full_depth_grid_gathers_dense_bands(sb, oracle_mod, scenes):
    # sparse shares come from the default kernel, which needs a grid as deep as the tree.  A tree with inconsistent parent
    # links (legal input: the shader just follows them) or deeper than 12 levels is rendered by the generic kernel on one
    # device -- and across devices as dense bands, bit for bit the same frame.
    od = scenes["sphere_d4"]
    s = od.Structs.copy()
    s[9:17, 0] = 2                                     # the block of node 1 claims node 2 as its parent
    bad = sb.OctData(s, od.Values)
    W, H = 120, 88
    cam = make_camera("default", W, H)
    ref, _ = oracle_mod.render(s, od.Values, cam.State, W, H)
    with sb.Scene(bad) as one, sb.MultiScene(bad, [0, 0, 0]) as ms:
        assert not one.stack_kernel_ok
        assert_frames_identical(ms.Draw(cam, W, H), ref, "inconsistent tree over three ranks")
        ms.configure(band_rows=8, rank0_weight=0.5)
        img, st = ms.Draw(cam, W, H, want_stats=True)
        assert_frames_identical(img, ref, "inconsistent tree, weighted deal")
        assert st.gathered_bytes > 0
        for dbg in (False, True):
            assert (ms.Draw(cam, W, H, flags=sb.FLAG_DISPLAY_DEBUG if dbg else sb.FLAG_DISPLAY) == one.DrawDisplay(cam, W, H, debug=dbg)).all()
        ms.Submit(2, [cam, make_camera("rotated", W, H)], W, H)          # a group of two frames, dense
        ptr = ms.Wait(2)
        import torch
        frames = torch.empty((2, H, W, 4), dtype=torch.float32, device="cuda")
        hip = ctypes.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(ctypes.c_void_p(frames.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(frames.numel() * 4), 3) == 0
        assert_frames_identical(frames[0].cpu().numpy(), ref, "group frame 0")
        ref2, _ = oracle_mod.render(s, od.Values, make_camera("rotated", W, H).State, W, H)
        assert_frames_identical(frames[1].cpu().numpy(), ref2, "group frame 1")


@pytest.fixture(scope="module")
def dragon(sb):
    od = sb.dragon_standin(9)
    sc = sb.Scene(od)
    yield od, sc
    sc.close()


@pytest.mark.gpu
def test_multi_4k_moving_camera_forced_resend_and_groups(sb, torch_mod, dragon):
    # BASELINE cfg-4's frame, 3840x2160 of the depth-9 stand-in, through the device list [0, 0, 0, 0]: one frame per call with
    # a camera that moves every frame; float tails forced to be sent again; then groups of four frames in four slots
    torch = torch_mod
    sys.path.insert(0, REPO)
    import bench
    od, one = dragon
    W, H = 3840, 2160
    cams = bench.orbit_cameras(sb, W, H, 6)
    refs = [whole_frame(sb, torch, one, c, W, H) for c in cams]
    with sb.MultiScene(od, [0, 0, 0, 0]) as ms:
        host = np.empty((H, W, 4), dtype=np.float32)
        for k, c in enumerate(cams):
            forced = k == 3 and sb._lib.EXPERIMENTS
            if forced:
                ms.debug_floats_sent(1024)                    # far too few floats travel with the next shares (a hook of the experiments build)
            img, st = ms.Draw(c, W, H, want_stats=True, out=host)
            assert np.array_equal(img.view(np.uint32), refs[k].cpu().numpy().view(np.uint32)), f"frame {k}"
            assert (st.resends > 0) == forced, (k, st.resends)
            assert st.gathered_bytes > 0 and all(st.rank_ms[r] > 0 for r in range(4))
        # the estimate recovered: the frame after the forced one needs no resend, and sends less than the full float arrays
        _, st = ms.Draw(cams[0], W, H, want_stats=True, out=host)
        assert st.resends == 0 and st.gathered_bytes < 3 * (544 * W * 4)
        # groups: 4 slots x 4 frames in flight, into the caller's buffers and into the slots' own
        mine = [torch.zeros((4, H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        for slot in range(4):
            group = [cams[(slot + i) % len(cams)] for i in range(4)]
            ms.Submit(slot, group, W, H, out_ptr=mine[slot].data_ptr() if slot < 2 else None)
        for slot in range(4):
            ptr, st = ms.Wait(slot, want_stats=True)
            frames = mine[slot] if slot < 2 else None
            if frames is None:                     # the slot's own buffer lives in the library: copy it out
                frames = torch.empty((4, H, W, 4), dtype=torch.float32, device="cuda")
                hip = ctypes.CDLL("libamdhip64.so")
                assert hip.hipMemcpy(ctypes.c_void_p(frames.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(frames.numel() * 4), 3) == 0
            else:
                assert ptr == mine[slot].data_ptr()
            for i in range(4):
                assert torch.equal(frames[i].view(torch.int32), refs[(slot + i) % len(cams)].view(torch.int32)), (slot, i)
        # a slot in flight cannot be submitted to again, and a geometry change needs idle slots
        ms.Submit(0, cams[0], W, H)
        with pytest.raises(sb.SdfHipError):
            ms.Submit(0, cams[0], W, H)
        with pytest.raises(sb.SdfHipError):
            ms.Submit(1, cams[0], 640, 360)
        ms.Wait(0)
        assert_frames_identical(ms.Draw(make_camera("rotated", 640, 360), 640, 360), one.Draw(make_camera("rotated", 640, 360), 640, 360), "after a geometry change")


@pytest.mark.gpu
def test_multi_rccl_transport_on_one_device(sb, oracle_mod, scenes):
    # SDFHIP_MULTI_TRANSPORT=rccl with one device and SDFHIP_MULTI_RCCL_SELF=1: the device's share goes through ncclSend /
    # ncclRecv to itself (RCCL loaded with dlopen, ncclCommInitAll, one group per share) before it is expanded -- all of the
    # RCCL code path that one GPU can run.  In a child process: the environment is read at create.
    code = (
        "import os, sys, numpy as np\n"
        f"sys.path.insert(0, {REPO!r}); sys.path.insert(0, os.path.join({REPO!r}, 'tests'))\n"
        "import sdfbox_amd as sb, oracle\n"
        "from conftest import make_camera, bits_equal\n"
        "oracle.build()\n"
        "od = sb.torus_d6(); W, H = 200, 120; cam = make_camera('rotated', W, H)\n"
        "ref, _ = oracle.render(od.Structs, od.Values, cam.State, W, H, nthreads=8)\n"
        "with sb.MultiScene(od, [0]) as ms:\n"
        "    assert ms.transport == 'rccl', ms.transport\n"
        "    links = ms.selftest()\n"
        "    assert links[0]['ok'] and links[0]['push_ms'] > 0, links        # the pattern went through ncclSend / ncclRecv\n"
        "    for k in range(3):\n"
        "        img = ms.Draw(cam, W, H)\n"
        "        assert bits_equal(img, ref).all(), k\n"
        "print('rccl self ok')\n")
    env = dict(os.environ, SDFHIP_MULTI_TRANSPORT="rccl", SDFHIP_MULTI_RCCL_SELF="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert out.returncode == 0 and "rccl self ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
