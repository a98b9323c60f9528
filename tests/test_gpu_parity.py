"""Parity of the HIP path with the oracle, through the C ABI, on a real MI355X.
Bit-exact: the kernels and the oracle share one arithmetic contract (fp32,
contraction off, same operation order), so every channel of every pixel,
and the algorithmic counters, must be identical."""
import hashlib
import os
import threading

import numpy as np
import pytest

from conftest import CAMERAS, GOLDEN, assert_frames_identical, bits_equal, make_camera

pytestmark = pytest.mark.gpu

# Every test of this file runs on BOTH flavours of the library: the product (libsdfhip.so, include/sdfhip.h) and the experiments
# build (libsdfhip_lab.so, include/sdfhip_experimental.h: the same sources + the A/B kernel forms and superseded gather formats).
# "stack" = the default: k_march (primary march, shading, shadow march as wave-converged loops) wherever the scene has a full-depth or split
# grid (every test scene of depth <= 12 does); "generic" = the shader's own traversal; "+compact" = BASELINE cfg-3's wavefront ray compaction
# (SDFHIP_FLAG_COMPACT: on "stack" the shadow rays of sparse waves queued for k_shadow, on "generic" persistent waves with lane refill).
# Experiments build only: "stack+one" = round 1's one-kernel form of the same traversal; "stack+queue" = EVERY shadow ray queued for
# k_shadow (SDFHIP_TUNE_SHADOW_QUEUE); "stack+persistent" = the persistent-wave kernel on a grid scene too (SDFHIP_TUNE_PERSISTENT_WAVES,
# the flag's form until round 4); "stack+bytes" = SDFHIP_TUNE_BYTE_CELLS (a no-op unless SDFHIP_SAMPLE_RECORDS=1 built the grid's second form)
PRODUCT_VARIANTS = ["generic", "stack", "generic+compact", "stack+compact"]
LAB_VARIANTS = ["stack+one", "stack+queue", "stack+persistent", "stack+bytes"]
ALL_VARIANTS = PRODUCT_VARIANTS + LAB_VARIANTS


@pytest.fixture(scope="module", params=["product", "lab"])
def sb(request):
    import sdfbox_amd
    if request.param == "product":
        return sdfbox_amd
    import sdfbox_amd.lab
    return sdfbox_amd.lab.load()


def variants_of(sb):
    return ALL_VARIANTS if sb._lib.EXPERIMENTS else PRODUCT_VARIANTS


def skip_unless_offered(sb, variant):
    if variant in LAB_VARIANTS and not sb._lib.EXPERIMENTS:
        pytest.skip("an A/B form of the experiments build")


def ab(sb, *flags):
    """the A/B forms' flags where the library offers them"""
    return tuple(flags) if sb._lib.EXPERIMENTS else ()


def lab_only(sb):
    if not sb._lib.EXPERIMENTS:
        pytest.skip("include/sdfhip_experimental.h: the experiments build only")


def flags_of(sb, name):
    f = {"generic": sb.KERNEL_GENERIC, "stack": sb.KERNEL_STACK}[name.split("+")[0]]
    return f | (sb.FLAG_COMPACT if name.endswith("compact") else 0) | (sb.TUNE_ONE_KERNEL if name.endswith("+one") else 0) | \
        (sb._lib.TUNE_SHADOW_QUEUE if name.endswith("+queue") else 0) | (sb._lib.TUNE_BYTE_CELLS if name.endswith("+bytes") else 0) | \
        ((sb.FLAG_COMPACT | sb._lib.TUNE_PERSISTENT_WAVES) if name.endswith("+persistent") else 0)


@pytest.fixture(scope="module")
def gpu_scenes(sb, scenes):
    out = {k: sb.Scene(v, device=0) for k, v in scenes.items()}
    yield out
    for s in out.values():
        s.close()


def test_native_library_is_the_one_in_tree(sb):
    # the driver records which .so the process loaded: it must be sdfbox_amd/libsdfhip.so (and, for the A/B forms, libsdfhip_lab.so)
    with open("/proc/self/maps") as f:
        maps = f.read()
    assert sb._lib.LIB_PATH in maps and os.path.dirname(sb._lib.LIB_PATH).endswith("sdfbox_amd")
    assert os.path.basename(sb._lib.LIB_PATH) == ("libsdfhip_lab.so" if sb._lib.EXPERIMENTS else "libsdfhip.so")


def test_product_refuses_the_experiments_flags(sb, gpu_scenes):
    # the A/B knobs exist in the experiments build only: the product says so instead of silently rendering the default
    if sb._lib.EXPERIMENTS:
        pytest.skip("the product flavour's behaviour")
    cam = make_camera("default", 32, 32)
    for fl in (sb.TUNE_ONE_KERNEL, sb.TUNE_SHADOW_QUEUE, sb.FLAG_WIRE, sb._lib.TUNE_LDS_TOP, sb._lib.TUNE_BYTE_CELLS, 1 << 8, 2 << 12):
        with pytest.raises(sb.SdfHipError) as e:
            gpu_scenes["sphere_d4"].Draw(cam, 32, 32, fl)
        assert e.value.code == sb._lib.ERR_ARG and "experimental" in str(e.value)
    for name in sb._lib.EXPERIMENTAL_SYMBOLS:
        assert not hasattr(sb._lib.lib, name), name


def test_unorm_table(sb, oracle_mod):
    lab_only(sb)
    # the kernel's reciprocal + fma fix-up decode == byte / 255.0f for all 256 bytes
    assert bits_equal(sb.unorm_table(0), oracle_mod.unorm_table()).all()


@pytest.mark.parametrize("variant", ALL_VARIANTS)
def test_golden_frames(sb, gpu_scenes, variant):
    skip_unless_offered(sb, variant)
    g = np.load(os.path.join(GOLDEN, "frames.npz"))
    for sname, scene in gpu_scenes.items():
        for cname in CAMERAS:
            cam = make_camera(cname, 64, 64)
            img, st = scene.Draw(cam, 64, 64, flags_of(sb, variant) | sb.FLAG_COUNT, want_stats=True)
            assert_frames_identical(img, g[f"{sname}/{cname}/rgba"], f"{variant} {sname}/{cname}")
            assert [st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays] == g[f"{sname}/{cname}/counters"].tolist()
            img2 = scene.Draw(cam, 64, 64, flags_of(sb, variant))          # non-counting build of the kernel
            assert_frames_identical(img2, img, f"{variant} count vs no-count")


@pytest.mark.parametrize("variant", ALL_VARIANTS)
@pytest.mark.parametrize("size", [(256, 256), (200, 120), (1, 1), (7, 3), (17, 33), (129, 65)])
def test_against_oracle_cfg1_and_ragged_sizes(sb, oracle_mod, scenes, gpu_scenes, variant, size):
    skip_unless_offered(sb, variant)
    # cfg-1 (256x256 sphere_d4, default camera) plus frames that do not fill whole
    # 8x8 / 16x16 tiles; the kernels must neither skip nor write outside W x H
    W, H = size
    for sname in ("sphere_d4", "torus_d6"):
        cam = make_camera("default" if sname == "sphere_d4" else "rotated", W, H)
        ref, cnt = oracle_mod.render(scenes[sname].Structs, scenes[sname].Values, cam.State, W, H, nthreads=8)
        img, st = gpu_scenes[sname].Draw(cam, W, H, flags_of(sb, variant) | sb.FLAG_COUNT, want_stats=True)
        assert_frames_identical(img, ref, f"{variant} {sname} {W}x{H}")
        assert (st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays) == tuple(int(c) for c in cnt)


def test_tuning_knobs_never_change_results(sb, gpu_scenes):
    lab_only(sb)
    # blockIdx -> tile order (bits 8..11) and workgroup size (bits 12..15) are pure speed knobs
    scene = gpu_scenes["torus_d6"]
    cam = make_camera("rotated", 333, 211)
    base = scene.Draw(cam, 333, 211, sb.KERNEL_STACK)
    for kern in (sb.KERNEL_STACK, sb.KERNEL_GENERIC):
        for order in (0, 1, 2):
            for block in (0, 1, 2, 3):
                img = scene.Draw(cam, 333, 211, kern | (order << 8) | (block << 12))
                assert_frames_identical(img, base, f"kernel {kern} order {order} block {block}")


def test_camera_edge_cases(sb, oracle_mod, scenes, gpu_scenes):
    W, H = 96, 64
    od, scene = scenes["torus_d6"], gpu_scenes["torus_d6"]
    cams = []
    c = sb.Logic(W, H); c.Position = (0.5, 0.5, 0.5); cams.append(("inside the hole", c))
    c = sb.Logic(W, H); c.Position = (0.5 + 0.25, 0.5, 0.5); cams.append(("inside the solid: negative prox", c))
    c = sb.Logic(W, H); c.Position = (3.0, -2.0, -4.0); c.Heading = (0.4, -0.6); cams.append(("far outside the cube", c))
    c = sb.Logic(W, H); c.Position = (0.0, 0.0, 0.0); cams.append(("camera on the light: normalize(0)", c))
    c = sb.Logic(W, H); c.State.margin = 0.05; cams.append(("huge margin", c))
    # 2 * margin >= 1 = the initial prox: the primary march takes no step and the gradient is taken before any
    # find(), in the root box with node 0's values (Compute.hlsl:194,207)
    c = sb.Logic(W, H); c.State.margin = 0.5; cams.append(("margin 0.5: shading before any find", c))
    c = sb.Logic(W, H); c.State.margin = 0.75; c.Position = (0.4, 0.6, 0.3); cams.append(("margin 0.75 from inside the cube", c))
    c = sb.Logic(W, H); c.State.margin = 0.0; cams.append(("zero margin: 100-step grazing", c))
    c = sb.Logic(W, H); c.State.fov = 6.0; c.State.strength = 3.0; cams.append(("wide fov, strong light", c))
    c = sb.Logic(W, H); c.State.light[0] = 0.5; c.State.light[1] = 0.5; c.State.light[2] = 0.5; cams.append(("light inside the object", c))
    c = sb.Logic(W, H); c.State.limit = 0.0; cams.append(("limit 0: everything is sky at step 0", c))
    c = sb.Logic(W, H); c.State.position[0] = float("nan"); cams.append(("NaN position", c))
    c = sb.Logic(W, H); c.State.screen_size[0] = 640.0; c.State.screen_size[1] = 480.0; cams.append(("screen_size != frame size", c))
    for what, cam in cams:
        ref, cnt = oracle_mod.render(od.Structs, od.Values, cam.State, W, H, nthreads=8)
        for variant in variants_of(sb):
            img, st = scene.Draw(cam, W, H, flags_of(sb, variant) | sb.FLAG_COUNT, want_stats=True)
            assert_frames_identical(img, ref, f"{what} / {variant}")
            assert (st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays) == tuple(int(c) for c in cnt), what
            # the build that does not count (the one that is timed) is its own set of kernels
            assert_frames_identical(scene.Draw(cam, W, H, flags_of(sb, variant)), ref, f"{what} / {variant}, not counting")


def _chain_tree(depth):
    """A consistent tree that is `depth` levels deep along the (0,0,0) corner."""
    rng = np.random.default_rng(depth)
    n = 1 + 8 * depth
    s = np.full((n, 2), -1, dtype=np.int32)
    for lvl in range(depth):
        node = 0 if lvl == 0 else 1 + 8 * (lvl - 1)       # child 0 of the previous block
        s[node, 1] = 1 + 8 * lvl
        s[1 + 8 * lvl: 9 + 8 * lvl, 0] = node
    v = rng.integers(40, 255, size=(n, 8), dtype=np.uint8)
    return s, v


def test_degenerate_and_deep_trees(sb, oracle_mod):
    W, H = 48, 40
    cam = sb.Logic(W, H); cam.Position = (0.3, 0.4, -0.2)
    # a single leaf: the root has no children
    od = sb.OctData(np.array([[-1, -1]], dtype=np.int32), np.array([[60, 80, 90, 120, 70, 200, 40, 255]], dtype=np.uint8))
    ref, cnt = oracle_mod.render(od.Structs, od.Values, cam.State, W, H)
    with sb.Scene(od) as sc:
        assert sc.stack_kernel_ok and sc.depth == 0
        for variant in variants_of(sb):
            assert_frames_identical(sc.Draw(cam, W, H, flags_of(sb, variant)), ref, f"single leaf / {variant}")
    # depth 12: still within the shader's 12-descent limit -> stack kernel allowed
    s, v = _chain_tree(12)
    od = sb.OctData(s, v)
    cam2 = sb.Logic(W, H); cam2.Position = (0.0001, 0.0002, -0.1); cam2.State.fov = 0.02
    ref, cnt = oracle_mod.render(s, v, cam2.State, W, H)
    with sb.Scene(od) as sc:
        assert sc.stack_kernel_ok and sc.depth == 12
        for variant in variants_of(sb):
            img, st = sc.Draw(cam2, W, H, flags_of(sb, variant) | sb.FLAG_COUNT, want_stats=True)
            assert_frames_identical(img, ref, f"depth 12 / {variant}")
            assert (st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays) == tuple(int(c) for c in cnt)
    # depth 14: the reference's `iterations < 12` cap binds; only the generic kernel
    # reproduces that, AUTO must pick it and an explicit STACK request is refused
    s, v = _chain_tree(14)
    od = sb.OctData(s, v)
    ref, cnt = oracle_mod.render(s, v, cam2.State, W, H)
    with sb.Scene(od) as sc:
        assert not sc.stack_kernel_ok and sc.depth == 14
        for fl in (sb.KERNEL_AUTO, sb.KERNEL_GENERIC, sb.KERNEL_GENERIC | sb.FLAG_COMPACT):
            img, st = sc.Draw(cam2, W, H, fl | sb.FLAG_COUNT, want_stats=True)
            assert (st.kernel_used & 0xF) == sb.KERNEL_GENERIC
            assert_frames_identical(img, ref, "depth 14 / generic")
            assert (st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays) == tuple(int(c) for c in cnt)
        with pytest.raises(sb.SdfHipError):
            sc.Draw(cam2, W, H, sb.KERNEL_STACK)


def test_inconsistent_parent_links_use_the_generic_kernel(sb, oracle_mod, scenes):
    # parent links that do not mirror the children links are legal input for the
    # shader (it just follows them); the stack kernel must not be used for them
    od = scenes["sphere_d4"]
    s = od.Structs.copy()
    s[9:17, 0] = 2                                     # block of node 1 claims node 2 as its parent
    bad = sb.OctData(s, od.Values)
    cam = make_camera("default", 80, 80)
    ref, cnt = oracle_mod.render(s, od.Values, cam.State, 80, 80)
    with sb.Scene(bad) as sc:
        assert not sc.stack_kernel_ok
        img, st = sc.Draw(cam, 80, 80, sb.KERNEL_AUTO | sb.FLAG_COUNT, want_stats=True)
        assert_frames_identical(img, ref, "inconsistent tree")
        assert (st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays) == tuple(int(c) for c in cnt)
    s = od.Structs.copy(); s[3, 1] = od.Length          # out of range -> refused at upload, nothing reaches the GPU
    with pytest.raises(sb.SdfHipError) as e:
        sb.Scene(sb.OctData(s, od.Values))
    assert e.value.code == sb._lib.ERR_BAD_TREE


def test_upload_validates_on_the_device_like_the_host_function(sb):
    # sdfhip_scene_upload checks the tree on the device (k_validate: 14 ms of upload for the 28 M-node scene instead of 128);
    # its verdicts must be sdfhip_octdata_validate's: error code, consistency, depth -- on good trees, inconsistent ones,
    # orphan subtrees deeper than anything the root reaches, and every kind of bad link
    rng = np.random.default_rng(11)
    od = sb.sphere_d4()
    N = od.Length
    cases = [("sphere_d4", od.Structs.copy())]
    for seed, depth in ((1, 5), (2, 9), (3, 12)):
        s, _ = _random_tree(np.random.default_rng(seed), depth, p_split=0.5)
        cases.append((f"random tree depth {depth}", s))
    s = od.Structs.copy(); s[0, 0] = 3; cases.append(("root with a parent", s))
    s = od.Structs.copy(); s[9:17, 0] = 2; cases.append(("a block that names another node as its parent", s))
    s = od.Structs.copy(); s[5, 1] = 0 if s[5, 1] < 0 else s[5, 1]; s[5, 1] = 0; cases.append(("children block 0", s))
    # an orphan chain hanging under a leaf that does not list it: deeper than the tree, but nothing reaches it
    s = np.concatenate([od.Structs, np.full((40, 2), -1, np.int32)]); leaf = int(np.nonzero(od.Structs[:, 1] < 0)[0][-1])
    s[N, 0] = leaf
    for k in range(1, 40):
        s[N + k, 0] = N + k - 1
    cases.append(("orphan chain of 40 under a leaf", s))
    s = od.Structs.copy(); s[5, 1] = N - 3; cases.append(("children block past the end", s))
    s = od.Structs.copy(); s[9, 0] = N + 10; cases.append(("parent out of range", s))
    s = od.Structs.copy(); s[9, 0] = 17; s[17, 0] = 9; cases.append(("parent cycle", s))
    s = od.Structs.copy(); s[1:200, 0] = np.arange(0, 199); cases.append(("199-link parent chain", s))
    s = np.concatenate([od.Structs, np.full((70, 2), -1, np.int32)])
    for k in range(70):
        s[N + k, 0] = (N + k - 1) if k else 0
    cases.append(("a 70-link chain", s))
    s = np.concatenate([od.Structs, np.full((60, 2), -1, np.int32)])
    for k in range(60):
        s[N + k, 0] = (N + k - 1) if k else 0
    cases.append(("a 60-link chain (allowed; nothing reaches it)", s))
    for links in (16, 17, 63, 64, 65, 128, 129):        # round the limits of the pointer jumping's two stages (16 links, 128) and of the rule (64)
        s = np.concatenate([od.Structs, np.full((links, 2), -1, np.int32)])
        for k in range(links):
            s[N + k, 0] = (N + k - 1) if k else 0
        cases.append((f"a {links}-link chain", s))
    for depth in (13, 16, 17, 40, 64):                  # consistent trees deeper than the first four rounds resolve: the depth must come out
        cases.append((f"a consistent chain tree of depth {depth}", _chain_tree(depth)[0]))
    s = _chain_tree(30)[0]; s[1 + 8 * 20, 0] = 5        # ... and one whose chain is broken in the middle: everything below is unattached
    cases.append(("a deep chain tree with a false parent link half-way", s))
    assert rng is not None
    for name, s in cases:
        s = np.ascontiguousarray(s, dtype=np.int32)
        v = np.zeros((len(s), 8), dtype=np.uint8); v[:] = 128
        tree = sb.OctData(s, v)
        try:
            depth, consistent = tree.validate()
            expect = ("ok", depth, consistent)
        except sb.SdfHipError as e:
            expect = ("error", e.code, None)
        try:
            with sb.Scene(tree) as sc:
                got = ("ok", sc.depth, sc.stack_kernel_ok)
        except sb.SdfHipError as e:
            got = ("error", e.code, None)
        if expect[0] == "ok":
            depth, consistent = expect[1], expect[2]
            assert got == ("ok", depth if consistent else 0xFFFFFFFF, bool(consistent and depth <= 12)), (name, expect, got)
        else:
            assert got == expect, (name, expect, got)


def test_band_rendering_reassembles_the_frame(sb, gpu_scenes):
    # the multi-GPU sharding, all ranks played by one GPU: each rank's bands into a compact
    # buffer, "gathered" side by side, de-interleaved by the rank-0 kernel
    import torch
    BandLayout, deinterleave, render_bands = (sb.tiles.BandLayout, sb.tiles.deinterleave, sb.tiles.render_bands)
    scene = gpu_scenes["torus_d6"]
    for (W, H, world, band_rows, variant) in [(160, 100, 3, 16, "stack"), (90, 77, 4, 8, "stack+compact"),
                                             (64, 64, 8, 16, "generic"), (70, 50, 2, 24, "generic+compact")]:
        cam = make_camera("rotated", W, H)
        full = torch.from_numpy(scene.Draw(cam, W, H, flags_of(sb, variant))).cuda()
        lay = BandLayout(H, world, band_rows)
        gathered = torch.zeros((world, lay.rows_per_rank, W, 4), dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        for r in range(world):
            render_bands(scene, cam, W, lay, r, gathered[r].data_ptr(), flags=flags_of(sb, variant), stream=stream)
        frame = torch.full((H, W, 4), -1.0, dtype=torch.float32, device="cuda")
        deinterleave(0, gathered.data_ptr(), frame.data_ptr(), W, lay, stream=stream)
        torch.cuda.synchronize()
        assert torch.equal(frame.view(torch.int32), full.view(torch.int32)), (W, H, world, band_rows, variant)


@pytest.mark.parametrize("kernel", ["generic", "stack"])
def test_path_traced_mode(sb, oracle_mod, scenes, gpu_scenes, kernel):
    # BASELINE config 5 in small: the mode is defined by the oracle (no reference semantics);
    # no transcendental function in it, so parity is bit-exact, counters included
    g = np.load(os.path.join(GOLDEN, "frames.npz"))
    fl = flags_of(sb, kernel)
    for sname in ("sphere_d4", "torus_d6"):
        cam = make_camera("default", 48, 32)
        img, st = gpu_scenes[sname].DrawPath(cam, 48, 32, sb.PathTrace(spp=4), flags=fl | sb.FLAG_COUNT, want_stats=True)
        assert_frames_identical(img, g[f"{sname}/path4/rgba"], f"{kernel} {sname} golden path4")
        assert [st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays] == g[f"{sname}/path4/counters"].tolist()
        for cname, (W, H), pt in (("rotated", (61, 37), sb.PathTrace(spp=16, max_bounces=3)),
                                  ("closeup", (40, 40), sb.PathTrace(spp=3, max_bounces=5, seed=7, albedo=0.5)),
                                  ("default", (33, 20), sb.PathTrace(spp=2, max_bounces=0))):
            cam = make_camera(cname, W, H)
            ref, cnt = oracle_mod.render_pt(scenes[sname].Structs, scenes[sname].Values, cam.State, W, H, spp=pt.spp,
                                            max_bounces=pt.max_bounces, seed=pt.seed, albedo=pt.albedo, nthreads=8)
            img, st = gpu_scenes[sname].DrawPath(cam, W, H, pt, flags=fl | sb.FLAG_COUNT, want_stats=True)
            assert_frames_identical(img, ref, f"{kernel} {sname}/{cname} path")
            assert (st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays) == tuple(int(c) for c in cnt)
            assert_frames_identical(gpu_scenes[sname].DrawPath(cam, W, H, pt, flags=fl), img, "count vs no-count")
    with pytest.raises(sb.SdfHipError):
        gpu_scenes["sphere_d4"].DrawPath(make_camera("default", 8, 8), 8, 8, sb.PathTrace(spp=0))
    with pytest.raises(sb.SdfHipError):
        gpu_scenes["sphere_d4"].DrawPath(make_camera("default", 8, 8), 8, 8, flags=sb.FLAG_COMPACT)


@pytest.mark.parametrize("blocks,order", [("0", "1"), ("1", "1"), ("2", "1"), ("2", "0"), ("3", "1"), ("4", "1"), ("4", "0")])
def test_path_traced_mode_through_every_scatter_grid(sb, oracle_mod, scenes, blocks, order):
    # The bounce levels of the path-traced pipeline read a split grid of their own (sdfhip_upload_options.scatter_grid: the levels inside
    # a block, 0 = the scene's grid; scatter_order: blocks stored sub-cube by sub-cube or in x-y-z order).  Whatever the grid, the frame
    # is the oracle's.
    od = scenes["torus_d6"]
    W, H = 61, 37
    cam = make_camera("rotated", W, H)
    pt = sb.PathTrace(spp=8, max_bounces=3)
    ref, cnt = oracle_mod.render_pt(od.Structs, od.Values, cam.State, W, H, spp=pt.spp, max_bounces=pt.max_bounces, seed=pt.seed,
                                    albedo=pt.albedo, nthreads=8)
    with sb.Scene(od, scatter_grid=int(blocks), scatter_order=int(order)) as sc:
        img, st = sc.DrawPath(cam, W, H, pt, flags=sb.FLAG_COUNT, want_stats=True)
        assert_frames_identical(img, ref, f"scatter grid {blocks} order {order}, counting")
        assert (st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays) == tuple(int(c) for c in cnt)
        assert_frames_identical(sc.DrawPath(cam, W, H, pt), ref, f"scatter grid {blocks} order {order}")


def test_path_traced_bounce_levels_ordered_by_key(sb, oracle_mod, scenes):
    # A/B of the experiments build (SDFHIP_PT_SORT=R, profiles/r04_cfg5_sort_ab.txt in the history (commit 53ee955); profiles/r06_cfg5_xcd_order_ab.txt: measured without gain): before every bounce
    # level its queue entries are ordered by (region of the hit, octant of the outgoing direction) and read through a permutation.
    # A path's results do not depend on where its entry sits in a queue: every ordering must give the oracle's frame, counters included
    lab_only(sb)
    W, H = 96, 72
    # round 6: + every XCD walking a contiguous eighth of the order (SDFHIP_PT_SORT_XCD=1), and a level with lane refill
    # (SDFHIP_PT_REFILL=1: k_pt_bounce_refill, persistent waves) -- both measured and dropped, both bit-identical
    prev = {k: os.environ.get(k) for k in ("SDFHIP_PT_SORT", "SDFHIP_PT_SORT_FROM", "SDFHIP_PT_SORT_XCD", "SDFHIP_PT_REFILL")}
    try:
        for sname, cname in (("torus_d6", "rotated"), ("sphere_d4", "closeup")):
            od = scenes[sname]
            cam = make_camera(cname, W, H)
            pt = sb.PathTrace(spp=5, max_bounces=3)
            ref, cnt = oracle_mod.render_pt(od.Structs, od.Values, cam.State, W, H, spp=pt.spp, max_bounces=pt.max_bounces, seed=pt.seed,
                                            albedo=pt.albedo, nthreads=8)
            with sb.Scene(od) as sc:
                hits = None
                for bits, first, xcd, refill in (("1", "0", "0", "0"), ("2", "0", "0", "0"), ("3", "0", "0", "0"), ("3", "1", "0", "0"), ("2", "2", "0", "0"),
                                                 ("9", "0", "0", "0"),                                     # (9: out of range = off)
                                                 ("3", "0", "1", "0"), ("2", "1", "1", "0"), ("0", "0", "0", "1")):
                    os.environ["SDFHIP_PT_SORT"], os.environ["SDFHIP_PT_SORT_FROM"] = bits, first
                    os.environ["SDFHIP_PT_SORT_XCD"], os.environ["SDFHIP_PT_REFILL"] = xcd, refill
                    what = f"{sname}: bounce levels ordered with {bits} region bits from level {first}, XCD walk {xcd}, lane refill {refill}"
                    img, st = sc.DrawPath(cam, W, H, pt, flags=sb.FLAG_COUNT, want_stats=True)
                    assert_frames_identical(img, ref, what)
                    assert (st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays) == tuple(int(c) for c in cnt), what
                    hits = st.n_hits if hits is None else hits
                    assert st.n_hits == hits and st.n_shadow_rays <= st.n_hits <= W * H * pt.spp * (pt.max_bounces + 1), what    # the queues' entries
                    assert_frames_identical(sc.DrawPath(cam, W, H, pt), ref, what + ", not counting")
    finally:
        for k, v in prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_path_traced_config5_full_size(sb, oracle_mod, dragon):
    # BASELINE config 5: 3840x2160, 16 spp, 3 bounces, seed 0x5DFB0C5 on the dragon stand-in.
    # Size-independent properties + the oracle on sampled rows.
    od, sc = dragon
    W, H = 3840, 2160
    cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
    img, st = sc.DrawPath(cam, W, H, sb.PathTrace(), flags=sb.FLAG_COUNT, want_stats=True)
    a = img[..., 3].astype(np.float64)
    assert (a == np.floor(a)).all() and a.min() >= 16 * 1 and a.max() <= 16 * 4 * 140
    assert int(a.sum()) == st.n_steps == st.n_samples
    assert np.isfinite(img[..., :3]).mean() > 0.99 and np.nanmin(img[..., :3]) >= 0
    gen = sc.DrawPath(cam, W, H, sb.PathTrace(), flags=sb.KERNEL_GENERIC)
    assert_frames_identical(gen, img, "generic vs stack, config 5")
    for y0 in (300, 700, 1100, H - 4):
        ref, _ = oracle_mod.render_pt(od.Structs, od.Values, cam.State, W, H, row0=y0, nrows=4, nthreads=16)
        assert_frames_identical(img[y0:y0 + 4], ref, f"config 5 rows {y0}..{y0 + 3}")


def assert_display_close(got, ref, what):
    # pow() is the one operation the GPU and the host libm do not share bit for bit; after the
    # 8-bit quantiser a 1-ulp difference shows only on a rounding boundary.  Tolerance: no byte
    # off by more than 1 LSB, at least 99.9 % of the bytes equal.
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
    assert d.max() <= 1, f"{what}: max byte difference {d.max()}"
    assert (d == 0).mean() >= 0.999, f"{what}: only {(d == 0).mean():.5f} of the bytes equal"


@pytest.mark.parametrize("variant", ALL_VARIANTS)
def test_fused_display_pass(sb, oracle_mod, scenes, gpu_scenes, variant):
    skip_unless_offered(sb, variant)
    # N2: DisplayFrag.hlsl fused into the epilogue -- gamma RGBA8 and the debug heat map
    for sname in ("sphere_d4", "torus_d6"):
        for cname, (W, H) in (("default", (200, 120)), ("rotated", (97, 61)), ("closeup", (64, 64))):
            cam = make_camera(cname, W, H)
            frame, _ = oracle_mod.render(scenes[sname].Structs, scenes[sname].Values, cam.State, W, H, nthreads=8)
            got = gpu_scenes[sname].DrawDisplay(cam, W, H, flags=flags_of(sb, variant))
            assert_display_close(got, oracle_mod.display(frame), f"{variant} {sname}/{cname} gamma")
            heat = gpu_scenes[sname].DrawDisplay(cam, W, H, debug=True, flags=flags_of(sb, variant))
            assert (heat == oracle_mod.display(frame, debug=True)).all(), f"{variant} {sname}/{cname} heat map"
    # NaN colours (zero gradient) and step count 0 (limit 0) go through the UNORM rules
    cam = sb.Logic(64, 48); cam.State.limit = 0.0
    got = gpu_scenes["torus_d6"].DrawDisplay(cam, 64, 48, flags=flags_of(sb, variant))
    frame, _ = oracle_mod.render(scenes["torus_d6"].Structs, scenes["torus_d6"].Values, cam.State, 64, 48)
    assert (got == oracle_mod.display(frame)).all() and (got[..., 3] == 0).all()


def test_display_bands_reassemble(sb, gpu_scenes):
    import torch
    BandLayout, deinterleave, render_bands = (sb.tiles.BandLayout, sb.tiles.deinterleave, sb.tiles.render_bands)
    scene = gpu_scenes["torus_d6"]
    W, H, world = 150, 90, 3
    cam = make_camera("rotated", W, H)
    full = torch.from_numpy(scene.DrawDisplay(cam, W, H).view(np.uint32).reshape(H, W).astype(np.int64)).cuda()
    lay = BandLayout(H, world, 16)
    gathered = torch.zeros((world, lay.rows_per_rank, W), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for r in range(world):
        render_bands(scene, cam, W, lay, r, gathered[r].data_ptr(), flags=sb.FLAG_DISPLAY, stream=stream)
    frame = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    deinterleave(0, gathered.data_ptr(), frame.data_ptr(), W, lay, stream=stream, pixel_bytes=4)
    torch.cuda.synchronize()
    assert torch.equal(frame.to(torch.int64) & 0xFFFFFFFF, full)


def test_batched_launch_renders_each_camera(sb, gpu_scenes):
    # several frames in one launch (grid.y = frame): each with its own camera block, bands included
    import torch
    BandLayout = sb.tiles.BandLayout
    scene = gpu_scenes["torus_d6"]
    W, H = 150, 90
    cams = [make_camera(n, W, H) for n in ("default", "rotated", "closeup")]
    cams[1].State.margin = 0.002; cams[2].State.strength = 1.5; cams[2].State.fov = 0.7
    stream = torch.cuda.current_stream().cuda_stream
    for kern in (sb.KERNEL_STACK, sb.KERNEL_GENERIC):
        full = [torch.from_numpy(scene.Draw(c, W, H, kern)).cuda() for c in cams]
        out = torch.zeros((3, H, W, 4), dtype=torch.float32, device="cuda")
        scene.DrawBatchDevice(cams, W, H, out.data_ptr(), flags=kern, stream=stream)
        torch.cuda.synchronize()
        for f in range(3):
            assert torch.equal(out[f].view(torch.int32), full[f].view(torch.int32)), (kern, f)
        # a rank's bands of all three frames in one launch
        lay = BandLayout(H, 4, 16)
        for rank in (0, 3):
            part = torch.zeros((3, lay.rows_per_rank, W, 4), dtype=torch.float32, device="cuda")
            scene.DrawBatchDevice(cams, W, H, part.data_ptr(), nrows_out=lay.rows_per_rank, band_rows=16,
                                  band_first=rank, band_stride=4, flags=kern, stream=stream)
            torch.cuda.synchronize()
            for f in range(3):
                for lrow, y in lay.rows_of(rank):
                    assert torch.equal(part[f, lrow].view(torch.int32), full[f][y].view(torch.int32)), (kern, rank, f, y)
    disp = torch.zeros((3, H, W), dtype=torch.int32, device="cuda")
    scene.DrawBatchDevice(cams, W, H, disp.data_ptr(), flags=sb.FLAG_DISPLAY, stream=stream)
    torch.cuda.synchronize()
    for f in range(3):
        ref = scene.DrawDisplay(cams[f], W, H).view(np.uint32).reshape(H, W)
        assert (disp[f].cpu().numpy().view(np.uint32) == ref).all()
    with pytest.raises(sb.SdfHipError):
        scene.DrawBatchDevice(cams * 3, W, H, out.data_ptr(), stream=stream)           # 9 > 8 frames
    with pytest.raises(sb.SdfHipError):
        scene.DrawBatchDevice(cams, W, H, out.data_ptr(), flags=sb.FLAG_COMPACT, stream=stream)


def test_grouped_gather_deinterleave(sb, gpu_scenes):
    # one gather may carry several frames: gathered [world][frames][rows][W] -> [frames][H][W]
    import torch
    BandLayout, deinterleave, render_bands = (sb.tiles.BandLayout, sb.tiles.deinterleave, sb.tiles.render_bands)
    scene = gpu_scenes["sphere_d4"]
    W, H, world, G = 120, 70, 3, 3
    lay = BandLayout(H, world, 16)
    cams = [make_camera(n, W, H) for n in ("default", "rotated", "closeup")]
    full = [torch.from_numpy(scene.Draw(c, W, H)).cuda() for c in cams]
    gathered = torch.zeros((world, G, lay.rows_per_rank, W, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for r in range(world):
        for f, c in enumerate(cams):
            render_bands(scene, c, W, lay, r, gathered[r, f].data_ptr(), stream=stream)
    frames = torch.zeros((G, H, W, 4), dtype=torch.float32, device="cuda")
    deinterleave(0, gathered.data_ptr(), frames.data_ptr(), W, lay, stream=stream, frames=G)
    torch.cuda.synchronize()
    for f in range(G):
        assert torch.equal(frames[f].view(torch.int32), full[f].view(torch.int32)), f


def test_wire_pixels_expand_to_the_same_frame(sb, gpu_scenes):
    lab_only(sb)
    # the gather's 5-byte wire format (FLAG_WIRE) is lossless: ranks render wire pixels, rank 0's
    # de-interleave expands them, and the frame is the direct RGBA32F render bit for bit --
    # sky, lit, shadowed and back-facing pixels, single and batched launches, both kernels
    import torch
    BandLayout, deinterleave, render_bands, render_bands_batch, wire_shape = (sb.tiles.BandLayout, sb.tiles.deinterleave, sb.tiles.render_bands, sb.tiles.render_bands_batch, sb.tiles.wire_shape)
    stream = torch.cuda.current_stream().cuda_stream
    for name, (W, H, world, band_rows) in [("sphere_d4", (128, 96, 1, 96)), ("torus_d6", (150, 90, 3, 16)),
                                           ("torus_d6", (97, 61, 4, 8))]:
        scene = gpu_scenes[name]
        cams = [make_camera(n, W, H) for n in ("default", "rotated", "closeup")]
        for i, v in enumerate((0.9, 0.1, 0.2)):                          # a light that leaves shadowed and back-facing pixels
            cams[1].State.light[i] = v
        lay = BandLayout(H, world, band_rows)
        for kern in (sb.KERNEL_STACK, sb.KERNEL_GENERIC):
            full = [torch.from_numpy(scene.Draw(c, W, H, kern)).cuda() for c in cams]
            kinds = set()
            for f in full:
                rgb = f[..., :3]
                kinds |= {"sky"} if bool((rgb[..., 0] != rgb[..., 1]).any()) else set()
                kinds |= {"black"} if bool((rgb == 0).all(-1).any()) else set()
                kinds |= {"lit"} if bool(((rgb[..., 0] == rgb[..., 1]) & (rgb[..., 0] > 0)).any()) else set()
            assert kinds == {"sky", "black", "lit"}, kinds
            # one launch per frame
            gathered = torch.zeros((world, 3) + wire_shape(lay.rows_per_rank, W), dtype=torch.uint8, device="cuda")
            for r in range(world):
                for f, c in enumerate(cams):
                    render_bands(scene, c, W, lay, r, gathered[r, f].data_ptr(), flags=kern | sb.FLAG_WIRE, stream=stream)
            frames = torch.full((3, H, W, 4), -1.0, dtype=torch.float32, device="cuda")
            deinterleave(0, gathered.data_ptr(), frames.data_ptr(), W, lay, stream=stream, pixel_bytes=5, frames=3)
            torch.cuda.synchronize()
            for f in range(3):
                assert torch.equal(frames[f].view(torch.int32), full[f].view(torch.int32)), (name, kern, f)
            # the three frames in one launch per rank
            gathered2 = torch.zeros_like(gathered)
            for r in range(world):
                render_bands_batch(scene, cams, W, lay, r, gathered2[r].data_ptr(), flags=kern | sb.FLAG_WIRE, stream=stream)
            torch.cuda.synchronize()
            assert torch.equal(gathered2, gathered), (name, kern)
    # wire pixels are for device-resident gathers of the plain kernel only
    scene = gpu_scenes["sphere_d4"]
    cam = make_camera("default", 64, 64)
    buf = torch.zeros((64, 64, 4), dtype=torch.float32, device="cuda")
    with pytest.raises(sb.SdfHipError):
        scene.Draw(cam, 64, 64, sb.FLAG_WIRE)
    with pytest.raises(sb.SdfHipError):
        scene.DrawDevice(cam, 64, 64, buf.data_ptr(), flags=sb.FLAG_WIRE | sb.FLAG_COMPACT, stream=stream)
    with pytest.raises(sb.SdfHipError):
        scene.DrawDevice(cam, 64, 64, buf.data_ptr(), flags=sb.FLAG_WIRE | sb.FLAG_DISPLAY, stream=stream)
    with pytest.raises(sb.SdfHipError):
        scene.DrawPathDevice(cam, 64, 64, buf.data_ptr(), pt=sb.PathTrace(spp=1), flags=sb.FLAG_WIRE, stream=stream)


def test_weighted_band_layouts_reassemble_the_frame(sb, gpu_scenes):
    # layouts that give rank 0 a smaller share (it also assembles the frame): explicit band lists
    # on the render side, an owner table on the de-interleave side; all pixel formats, batched
    # launches and the path-traced mode
    import torch
    BandLayout, deinterleave, render_bands, render_bands_batch, wire_shape = (sb.tiles.BandLayout, sb.tiles.deinterleave, sb.tiles.render_bands, sb.tiles.render_bands_batch, sb.tiles.wire_shape)
    scene = gpu_scenes["torus_d6"]
    stream = torch.cuda.current_stream().cuda_stream
    for (W, H, world, band_rows, w0) in [(160, 100, 3, 8, 0.5), (90, 77, 4, 16, 0.8), (64, 200, 8, 8, 0.775), (70, 50, 2, 24, 0.9)]:
        lay = BandLayout(H, world, band_rows, w0)
        assert lay.weighted and len(lay.bands_of(0)) < len(lay.bands_of(1))
        cams = [make_camera(n, W, H) for n in ("rotated", "closeup")]
        full = [torch.from_numpy(scene.Draw(c, W, H)).cuda() for c in cams]
        # RGBA32F and wire pixels, one launch per frame and one per group
        for pb, fl in ((16, 0), (5, sb.FLAG_WIRE)) if sb._lib.EXPERIMENTS else ((16, 0),):
            if pb == 16:
                gathered = torch.zeros((world, 2, lay.rows_per_rank, W, 4), dtype=torch.float32, device="cuda")
            else:
                gathered = torch.zeros((world, 2) + wire_shape(lay.rows_per_rank, W), dtype=torch.uint8, device="cuda")
            for r in range(world):
                for f, c in enumerate(cams):
                    render_bands(scene, c, W, lay, r, gathered[r, f].data_ptr(), flags=fl, stream=stream)
            frames = torch.full((2, H, W, 4), -1.0, dtype=torch.float32, device="cuda")
            deinterleave(0, gathered.data_ptr(), frames.data_ptr(), W, lay, stream=stream, pixel_bytes=pb, frames=2)
            batched = torch.zeros_like(gathered)
            for r in range(world):
                render_bands_batch(scene, cams, W, lay, r, batched[r].data_ptr(), flags=fl, stream=stream)
            torch.cuda.synchronize()
            for f in range(2):
                assert torch.equal(frames[f].view(torch.int32), full[f].view(torch.int32)), (W, H, world, pb, f)
            assert torch.equal(batched, gathered), (W, H, world, pb)
        # RGBA8 frames of the fused display pass
        ref8 = torch.from_numpy(scene.DrawDisplay(cams[0], W, H).view(np.int32).reshape(H, W)).cuda()
        g8 = torch.zeros((world, lay.rows_per_rank, W), dtype=torch.int32, device="cuda")
        for r in range(world):
            render_bands(scene, cams[0], W, lay, r, g8[r].data_ptr(), flags=sb.FLAG_DISPLAY, stream=stream)
        f8 = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        deinterleave(0, g8.data_ptr(), f8.data_ptr(), W, lay, stream=stream, pixel_bytes=4)
        # path-traced mode
        pt = sb.PathTrace(spp=2)
        refp = torch.from_numpy(scene.DrawPath(cams[0], W, H, pt=pt)).cuda()
        gp = torch.zeros((world, lay.rows_per_rank, W, 4), dtype=torch.float32, device="cuda")
        for r in range(world):
            render_bands(scene, cams[0], W, lay, r, gp[r].data_ptr(), stream=stream, pt=pt)
        fp = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        deinterleave(0, gp.data_ptr(), fp.data_ptr(), W, lay, stream=stream)
        torch.cuda.synchronize()
        assert torch.equal(f8, ref8), (W, H, world)
        assert torch.equal(fp.view(torch.int32), refp.view(torch.int32)), (W, H, world)
    # argument checks of the band-list entry points
    cam = make_camera("default", 64, 64)
    buf = torch.zeros((64, 64, 4), dtype=torch.float32, device="cuda")
    with pytest.raises(sb.SdfHipError):      # band 8 of a frame with 8 bands
        scene.DrawBandsDevice([cam], 64, 64, buf.data_ptr(), 8, [0, 8], stream=stream)
    with pytest.raises(sb.SdfHipError):      # the list does not fit the output rows
        scene.DrawBandsDevice([cam], 64, 64, buf.data_ptr(), 8, [0, 1, 2], nrows_out=16, stream=stream)
    with pytest.raises(sb.SdfHipError):      # path-traced mode: one frame per launch
        scene.DrawBandsDevice([cam, cam], 64, 64, buf.data_ptr(), 8, [0], pt=sb.PathTrace(spp=1), stream=stream)


def test_top_grid_levels_never_change_results(sb, oracle_mod, scenes):
    # the top grid (one load instead of a chain of dependent ones when a descent restarts near the
    # root) at every level, none included: frames and algorithmic counters stay the oracle's
    import os
    rng = np.random.default_rng(7)
    od6 = scenes["torus_d6"]
    cases = [(scenes["sphere_d4"], "default"), (od6, "rotated"), (od6, "closeup")]
    W, H = 96, 80
    for od, camname in cases:
        cam = make_camera(camname, W, H)
        ongrid = sb.Logic(W, H); ongrid.Position = (0.25, 0.5, -0.125); ongrid.Heading = (0.0, 0.0)   # rays along cell faces
        refs = [oracle_mod.render(od.Structs, od.Values, c.State, W, H) for c in (cam, ongrid)]
        levels_seen = set()
        for lv in (0, 1, 2, 3, 4, 5, 6, None):
            with sb.Scene(od, top_grid_level=lv) as scene:
                levels_seen.add(scene.top_grid_level)
                assert scene.top_grid_level <= scene.depth
                assert scene.top_grid_bytes == (16 << (3 * scene.top_grid_level) if scene.top_grid_level else 0)
                if lv is None:       # the default for shallow trees: as deep as the tree (every leaf in the grid)
                    assert scene.top_grid_level == scene.depth
                for c, (ref, cnt) in zip((cam, ongrid), refs):
                    for flags in (sb.KERNEL_STACK, sb.KERNEL_STACK | sb.FLAG_COMPACT) + ((sb.KERNEL_STACK | sb.TUNE_ONE_KERNEL,) if sb._lib.EXPERIMENTS else ()):
                        img, st = scene.Draw(c, W, H, flags | sb.FLAG_COUNT, want_stats=True)
                        assert_frames_identical(img, ref, f"top grid {lv}")
                        assert (st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays) == tuple(int(v) for v in cnt), lv
                    if sb._lib.EXPERIMENTS and 0 < scene.top_grid_level <= 3 and scene.top_grid_level < scene.depth:
                        # the measurement variant that stages the top grid in LDS (64- and 256-thread workgroups)
                        for block in (0, 3):
                            img = scene.Draw(c, W, H, sb.KERNEL_STACK | sb.TUNE_ONE_KERNEL | sb._lib.TUNE_LDS_TOP | (block << 12))
                            assert_frames_identical(img, ref, f"top grid {lv} staged in LDS, block knob {block}")
                    pimg = scene.DrawPath(c, W, H, pt=sb.PathTrace(spp=2))
                    pref, _ = oracle_mod.render_pt(od.Structs, od.Values, c.State, W, H, spp=2)
                    assert_frames_identical(pimg, pref, f"top grid {lv}, path-traced")
        assert len(levels_seen) >= 4
        # split grids: a coarse dense level whose internal cells point at blocks of finer cells
        for sp in (1, 2, 3, 5):
            if sp >= od_depth(sb, od):
                continue
            with sb.Scene(od, top_grid_split=sp) as scene:
                assert scene.top_grid_level == int(sp) and scene.top_grid_bytes >= 16 << (3 * int(sp))
                for c, (ref, cnt) in zip((cam, ongrid), refs):
                    for flags in (sb.KERNEL_STACK, sb.KERNEL_STACK | sb.FLAG_COMPACT) + ((sb.KERNEL_STACK | sb.TUNE_ONE_KERNEL,) if sb._lib.EXPERIMENTS else ()):
                        img, st = scene.Draw(c, W, H, flags | sb.FLAG_COUNT, want_stats=True)
                        assert_frames_identical(img, ref, f"split grid {sp}")
                        assert (st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays) == tuple(int(v) for v in cnt), sp
                        assert_frames_identical(scene.Draw(c, W, H, flags), ref, f"split grid {sp}, not counting")
                        if sb._lib.EXPERIMENTS:
                            assert_frames_identical(scene.Draw(c, W, H, flags | sb._lib.TUNE_BYTE_CELLS), ref, f"split grid {sp}, byte cells")
                    pimg = scene.DrawPath(c, W, H, pt=sb.PathTrace(spp=2))
                    pref, _ = oracle_mod.render_pt(od.Structs, od.Values, c.State, W, H, spp=2)
                    assert_frames_identical(pimg, pref, f"split grid {sp}, path-traced")


def od_depth(sb, od):
    with sb.Scene(od) as sc:
        return sc.depth


def test_sparse_wire_format_is_lossless_within_its_capacity(sb, gpu_scenes):
    lab_only(sb)
    # dense wire shares compacted on the rendering side (codes + per-tile mask and slot index + packed
    # non-zero floats), expanded by the rank-0 kernel: the same frame bit for bit; too small a
    # capacity is reported, never silent
    import torch
    BandLayout, deinterleave_sparse, render_bands_batch, sparse_count, sparse_share_bytes, wire_compact, wire_shape = (sb.tiles.BandLayout, sb.tiles.deinterleave_sparse, sb.tiles.render_bands_batch, sb.tiles.sparse_count, sb.tiles.sparse_share_bytes, sb.tiles.wire_compact, sb.tiles.wire_shape)
    stream = torch.cuda.current_stream().cuda_stream
    scene = gpu_scenes["torus_d6"]
    for (W, H, world, band_rows, w0) in [(160, 96, 1, 96, 1.0), (150, 90, 3, 16, 1.0), (97, 61, 4, 8, 0.6), (64, 200, 8, 8, 0.5)]:
        lay = BandLayout(H, world, band_rows, w0)
        R = lay.rows_per_rank
        cams = [make_camera(n, W, H) for n in ("default", "rotated", "closeup")]
        for i, v in enumerate((0.9, 0.1, 0.2)):
            cams[1].State.light[i] = v
        full = [torch.from_numpy(scene.Draw(c, W, H)).cuda() for c in cams]
        dense = torch.zeros((world, 3) + wire_shape(R, W), dtype=torch.uint8, device="cuda")
        for r in range(world):
            render_bands_batch(scene, cams, W, lay, r, dense[r].data_ptr(), flags=sb.FLAG_WIRE, stream=stream)
        # generous capacity: every pixel could be lit
        cap = R * W
        nb = sparse_share_bytes(R, W, cap)
        sparse = torch.zeros((world, 3, nb), dtype=torch.uint8, device="cuda")
        for r in range(world):
            wire_compact(0, dense[r].data_ptr(), sparse[r].data_ptr(), W, R, 3, cap, stream=stream)
        frames = torch.full((3, H, W, 4), -1.0, dtype=torch.float32, device="cuda")
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        deinterleave_sparse(0, sparse.data_ptr(), frames.data_ptr(), W, lay, cap, stream=stream, frames=3, overflow_ptr=flag.data_ptr())
        torch.cuda.synchronize()
        assert int(flag.item()) == 0
        for f in range(3):
            assert torch.equal(frames[f].view(torch.int32), full[f].view(torch.int32)), (W, H, world, f)
        counts, over = sparse_count(sparse, R, W, cap)
        assert not any(over)
        # the counts are the pixels whose grey level has any bit set
        lit = sum(int((fr[..., 0].view(torch.int32) != 0).logical_and(fr[..., 0].view(torch.int32) == fr[..., 1].view(torch.int32)).sum())
                  for fr in full)      # bitwise: a NaN grey is a lit pixel too
        assert sum(counts) == lit, (sum(counts), lit)
        # exactly enough capacity works, one slot less is flagged
        tight = max(counts)
        if tight > 1:
            for capn, expect in ((tight, 0), (tight - 1, 1)):
                nb2 = sparse_share_bytes(R, W, capn)
                sp2 = torch.zeros((world, 3, nb2), dtype=torch.uint8, device="cuda")
                for r in range(world):
                    wire_compact(0, dense[r].data_ptr(), sp2[r].data_ptr(), W, R, 3, capn, stream=stream)
                flag.zero_()
                deinterleave_sparse(0, sp2.data_ptr(), frames.data_ptr(), W, lay, capn, stream=stream, frames=3, overflow_ptr=flag.data_ptr())
                torch.cuda.synchronize()
                assert int(flag.item()) == expect, (capn, tight)
                if not expect:
                    for f in range(3):
                        assert torch.equal(frames[f].view(torch.int32), full[f].view(torch.int32))
                    assert sparse_share_bytes(R, W, capn) < 5 * R * W or tight > R * W * 0.9


def test_sparse_wire_format_on_random_shares(sb):
    lab_only(sb)
    # the format itself, fed with arbitrary wire shares (any float bit pattern: NaNs, -0.0, denormals;
    # any legal code byte; ragged widths): expanding the sparse form equals expanding the dense form
    import torch
    BandLayout, deinterleave, deinterleave_sparse, sparse_count, sparse_share_bytes, wire_compact, wire_shape = (sb.tiles.BandLayout, sb.tiles.deinterleave, sb.tiles.deinterleave_sparse, sb.tiles.sparse_count, sb.tiles.sparse_share_bytes, sb.tiles.wire_compact, sb.tiles.wire_shape)
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(5)
    for (W, H, world, band_rows, frames, p_lit) in [(61, 40, 1, 8, 2, 0.3), (200, 64, 2, 16, 3, 0.05), (33, 96, 4, 8, 1, 0.9), (128, 24, 3, 8, 2, 0.0)]:
        lay = BandLayout(H, world, band_rows)
        R = lay.rows_per_rank
        n = world * frames * R * W
        bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (n,), generator=g, dtype=torch.int64).to(torch.int32)
        bits[torch.rand(n, generator=g) >= p_lit] = 0
        bits[:7] = torch.tensor([0x7FC00000, -0x400000, -2 ** 31, 1, 0x7F800000, 0, 0x00800000][:7], dtype=torch.int64).to(torch.int32)[:min(7, n)]
        codes = torch.randint(0, 141, (n,), generator=g, dtype=torch.int64)
        sky = torch.rand(n, generator=g) < 0.4
        codes[sky] = 255 - torch.randint(0, 101, (int(sky.sum()),), generator=g, dtype=torch.int64)
        dense = torch.zeros((world, frames) + wire_shape(R, W), dtype=torch.uint8)
        dense[:, :, :4] = bits.view(world, frames, R * W).view(torch.uint8).reshape(world, frames, 4, R, W)
        dense[:, :, 4] = codes.to(torch.uint8).view(world, frames, R, W)
        dense = dense.cuda()
        want = torch.zeros((frames, H, W, 4), dtype=torch.float32, device="cuda")
        deinterleave(0, dense.data_ptr(), want.data_ptr(), W, lay, stream=stream, pixel_bytes=5, frames=frames)
        cap = R * W
        sparse = torch.zeros((world, frames, sparse_share_bytes(R, W, cap)), dtype=torch.uint8, device="cuda")
        for r in range(world):
            wire_compact(0, dense[r].data_ptr(), sparse[r].data_ptr(), W, R, frames, cap, stream=stream)
        got = torch.full((frames, H, W, 4), -1.0, dtype=torch.float32, device="cuda")
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        deinterleave_sparse(0, sparse.data_ptr(), got.data_ptr(), W, lay, cap, stream=stream, frames=frames, overflow_ptr=flag.data_ptr())
        torch.cuda.synchronize()
        assert int(flag.item()) == 0
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (W, H, world)
        counts, _ = sparse_count(sparse, R, W, cap)
        assert sum(counts) == int((bits != 0).sum())


def test_two_handles_render_concurrently(sb, oracle_mod, scenes):
    # upload / render are callable from several threads on different handles
    cam = make_camera("default", 128, 128)
    refs = {k: oracle_mod.render(v.Structs, v.Values, cam.State, 128, 128, nthreads=4)[0] for k, v in scenes.items()}
    errors = []

    def work(name):
        try:
            with sb.Scene(scenes[name]) as sc:
                for _ in range(5):
                    assert_frames_identical(sc.Draw(cam, 128, 128), refs[name], name)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=work, args=(n,)) for n in ("sphere_d4", "torus_d6", "sphere_d4")]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_tile_order_hook_permutes_work_not_results(sb, oracle_mod, scenes):
    lab_only(sb)
    # sdfhip_debug_tile_order (the experiment hook of scripts/ab_tile_order.py and wave_iterations.py): workgroups take their
    # tiles from a permutation -- the frame must not change -- and the kernel reports every tile's wave-iterations: the longest
    # primary march of the tile in the low byte, the longest shadow march in the high byte
    import ctypes
    import torch
    lib, check = sb._lib.lib, sb._lib.check
    od = scenes["torus_d6"]
    W, H = 200, 120                                 # 25 x 15 tiles; the grid has 8 * ceil(15 / 8) * 25 = 400 workgroups
    cam = make_camera("closeup", W, H)
    ref, _ = oracle_mod.render(od.Structs, od.Values, cam.State, W, H, nthreads=8)
    tx, ty = (W + 7) // 8, (H + 7) // 8
    nblk = 8 * ((ty + 7) // 8) * tx
    rng = np.random.default_rng(5)
    with sb.Scene(od) as sc:
        costs = []
        for perm in (None, rng.permutation(nblk).astype(np.uint32)):
            cost = torch.zeros(tx * ty, dtype=torch.int16, device="cuda")
            buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
            p = None
            if perm is not None:
                # a permutation of the workgroups' tiles: entries >= the tile count idle, every tile appears once
                # (an entry is tile row << 16 | tile column)
                tiles = np.where(perm < tx * ty, (perm // tx) << 16 | (perm % tx), 0xFFFFFFFF).astype(np.uint32)
                p = torch.from_numpy(tiles.view(np.int32)).cuda()
            check(lib.sdfhip_debug_tile_order(sc._h, ctypes.c_void_p(p.data_ptr()) if p is not None else None, ctypes.c_void_p(cost.data_ptr())))
            sc.DrawDevice(cam, W, H, buf.data_ptr())
            torch.cuda.synchronize()
            assert_frames_identical(buf.cpu().numpy(), ref, "tile order hook, " + ("permuted" if perm is not None else "default order"))
            costs.append(cost.cpu().numpy().view(np.uint16).copy())
        check(lib.sdfhip_debug_tile_order(sc._h, None, None))
    assert (costs[0] == costs[1]).all()             # a tile's cost does not depend on which workgroup rendered it
    primary, shadow = costs[0] & 0xFF, costs[0] >> 8
    assert 1 <= primary.min() and primary.max() <= 100 and shadow.max() <= 40 and (shadow > 0).any()
    # a tile without a hit pixel casts no shadow ray, and its primary loop ran as long as its longest pixel: alpha = steps
    steps = ref[..., 3].astype(np.int64)
    sky = ref[..., 2] == np.float32(0.2)
    for t_ in range(tx * ty):
        y0, x0 = (t_ // tx) * 8, (t_ % tx) * 8
        if sky[y0:y0 + 8, x0:x0 + 8].all():
            assert shadow[t_] == 0 and primary[t_] == steps[y0:y0 + 8, x0:x0 + 8].max(), t_


@pytest.mark.parametrize("split", [None, "4"])
def test_tile_order_flag_changes_no_pixel(sb, oracle_mod, scenes, split):
    # SDFHIP_FLAG_TILE_ORDER: a frame's tiles are launched in descending order of their cost in the LAST frame of the same
    # geometry on the same stream.  Every frame must still be the oracle's: the first (default order), the same camera again
    # (its own costs), other cameras (a stale order), after a change of frame size and back, bands of a frame, two streams
    # taking turns, the counting build, the shadow-queue form -- output buffers start as NaN, so a tile that no workgroup
    # took would show
    import torch
    od = scenes["torus_d6"]
    with sb.Scene(od, top_grid_split=None if split is None else int(split)) as sc:
        F = sb.KERNEL_STACK | sb.FLAG_TILE_ORDER
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        refs = {}

        def ref(name, W, H):
            if (name, W, H) not in refs:
                refs[(name, W, H)] = oracle_mod.render(od.Structs, od.Values, make_camera(name, W, H).State, W, H, nthreads=8)
            return refs[(name, W, H)]

        def draw(name, W, H, flags=F, stream=None):
            buf = torch.full((H, W, 4), float("nan"), dtype=torch.float32, device="cuda")
            # (the fill runs on torch's stream, the render on its own -- torch's streams do not wait for the default stream: without
            # this, a late part of the fill lands on rendered pixels.  Seen once the library gave its process eight hardware queues.)
            torch.cuda.synchronize()
            st = sb.Stats()
            sc.DrawDevice(make_camera(name, W, H), W, H, buf.data_ptr(), flags=flags,
                          stream=stream.cuda_stream if stream is not None else None, stats=st if flags & sb.FLAG_COUNT else None)
            torch.cuda.synchronize()
            assert_frames_identical(buf.cpu().numpy(), ref(name, W, H)[0], f"tile order, {name} {W}x{H} flags {flags:#x}")
            if flags & sb.FLAG_COUNT:
                assert (st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays) == tuple(int(v) for v in ref(name, W, H)[1])

        for name, W, H in (("closeup", 200, 120), ("closeup", 200, 120), ("rotated", 200, 120), ("default", 200, 120),
                           ("closeup", 97, 61), ("closeup", 200, 120), ("rotated", 200, 120), ("rotated", 200, 120)):
            draw(name, W, H)
        for k in range(6):                                   # two streams, each with its own last frame
            draw(("closeup", "rotated", "default")[k % 3], 200, 120, stream=streams[k & 1])
        draw("closeup", 200, 120, flags=F | sb.FLAG_COUNT)
        draw("rotated", 200, 120, flags=F | sb.FLAG_COUNT)
        if sb._lib.EXPERIMENTS:
            draw("closeup", 200, 120, flags=F | sb.TUNE_SHADOW_QUEUE)
            draw("rotated", 200, 120, flags=F | sb.TUNE_SHADOW_QUEUE)
        # bands of a frame (what a rank of the sharded pipeline renders): rows 16..31, 48..63, ... twice, then the other ranks' bands
        W, H = 200, 128
        whole = ref("closeup", W, H)[0]
        for first in (1, 1, 0, 1):
            rows = [y for y in range(H) if (y // 16) % 2 == first]
            buf = torch.full((len(rows), W, 4), float("nan"), dtype=torch.float32, device="cuda")
            sc.DrawDevice(make_camera("closeup", W, H), W, H, buf.data_ptr(), nrows_out=len(rows), band_rows=16, band_first=first, band_stride=2, flags=F)
            torch.cuda.synchronize()
            assert_frames_identical(buf.cpu().numpy(), whole[rows], f"tile order, bands from {first}")


def test_frames_in_flight_on_one_handle_do_not_share_scratch(sb, oracle_mod, scenes, gpu_scenes):
    # Several streams render on ONE scene handle at once, each its own camera (frames in flight): the hit
    # queues of the queued-shadow pipeline (SDFHIP_TUNE_SHADOW_QUEUE) and the tile queues of the compact kernel are per-stream scratch, so
    # no launch may see another's.  (The compact kernel once kept its queue heads in the scene: a second
    # stream's launch reset them under the first one's feet.)
    import torch
    od, scene = scenes["torus_d6"], gpu_scenes["torus_d6"]
    W, H = 320, 200
    cams = []
    for k in range(6):
        c = sb.Logic(W, H); c.Position = (0.2 + 0.1 * k, 0.3 + 0.05 * k, -0.3 + 0.05 * k); c.Heading = (-0.1 * k, 0.15 * k)
        cams.append(c)
    refs = [oracle_mod.render(od.Structs, od.Values, c.State, W, H, nthreads=8)[0] for c in cams]
    streams = [torch.cuda.Stream() for _ in cams]
    for flags in (sb.KERNEL_STACK, sb.KERNEL_STACK | sb.FLAG_COMPACT, sb.KERNEL_STACK | sb.FLAG_TILE_ORDER) + ab(sb, sb.KERNEL_STACK | sb._lib.TUNE_SHADOW_QUEUE, sb.KERNEL_STACK | sb.TUNE_ONE_KERNEL):
        bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in cams]
        torch.cuda.synchronize()                   # (the fills run on torch's stream, the renders on their own)
        for rep in range(8):                       # keep every stream busy so that the launches really overlap
            for c, b, st in zip(cams, bufs, streams):
                scene.DrawDevice(c, W, H, b.data_ptr(), flags=flags, stream=st.cuda_stream)
        torch.cuda.synchronize()
        for k, (b, ref) in enumerate(zip(bufs, refs)):
            assert_frames_identical(b.cpu().numpy(), ref, f"flags {flags:#x}, stream {k}")


def test_a_stream_per_frame_and_counters_per_stream(sb, oracle_mod, scenes):
    # (1) A host that makes a new stream for every frame: more than 16 streams over the handle's life.  The 17th takes over the least
    # recently used scratch whose stream has drained (it used to be an error for ever after).  (2) SDFHIP_FLAG_COUNT renders count into
    # their own stream's scratch: a counting render without stats on one stream leaves another stream's figures alone.
    import torch
    od = scenes["torus_d6"]
    W, H = 160, 120
    cam = make_camera("rotated", W, H)
    ref, cnt = oracle_mod.render(od.Structs, od.Values, cam.State, W, H, nthreads=8)
    with sb.Scene(od) as sc:
        buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()                   # (the fill runs on torch's stream, the renders on their own)
        for k in range(40):
            st = torch.cuda.Stream()
            flags = (sb.KERNEL_STACK, sb.KERNEL_STACK | sb.FLAG_TILE_ORDER, sb.KERNEL_STACK | (sb._lib.TUNE_SHADOW_QUEUE if sb._lib.EXPERIMENTS else sb.FLAG_COUNT), sb.KERNEL_STACK | sb.FLAG_COMPACT)[k % 4]
            sc.DrawDevice(cam, W, H, buf.data_ptr(), flags=flags, stream=st.cuda_stream)
            st.synchronize()
            assert_frames_identical(buf.cpu().numpy(), ref, f"frame {k} on its own new stream")
            del st
        a, b = torch.cuda.Stream(), torch.cuda.Stream()
        other = make_camera("closeup", W, H)
        bufb = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        stats = sb.Stats()
        for _ in range(6):                         # stream b counts without ever reading its counters, while stream a's are read
            sc.DrawDevice(other, W, H, bufb.data_ptr(), flags=sb.KERNEL_STACK | sb.FLAG_COUNT, stream=b.cuda_stream)
        sc.DrawDevice(cam, W, H, buf.data_ptr(), flags=sb.KERNEL_STACK | sb.FLAG_COUNT, stream=a.cuda_stream, stats=stats)
        torch.cuda.synchronize()
        assert (stats.n_nodes, stats.n_samples, stats.n_steps, stats.n_shadow_rays) == tuple(int(c) for c in cnt)


def test_streams_created_and_destroyed_by_the_host(sb, oracle_mod, scenes):
    # A host that really destroys its per-frame streams (hipStreamCreate / hipStreamDestroy, not torch's pooled streams): the
    # handle's scratch slots must never hand a destroyed stream back to HIP -- idleness is asked of the library's own events --
    # and the 17th, 18th, ... stream must find a slot.
    import ctypes
    import torch
    hip = ctypes.CDLL("libamdhip64.so")
    od = scenes["torus_d6"]
    W, H = 160, 120
    cam = make_camera("rotated", W, H)
    ref, _ = oracle_mod.render(od.Structs, od.Values, cam.State, W, H, nthreads=8)
    with sb.Scene(od) as sc:
        buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        for k in range(48):
            st = ctypes.c_void_p()
            assert hip.hipStreamCreate(ctypes.byref(st)) == 0
            flags = (sb.KERNEL_STACK | sb.FLAG_TILE_ORDER, sb.KERNEL_STACK | sb.FLAG_COMPACT, sb.KERNEL_STACK | sb.FLAG_COUNT)[k % 3]
            sc.DrawDevice(cam, W, H, buf.data_ptr(), flags=flags, stream=st.value)
            assert hip.hipStreamSynchronize(st) == 0
            assert_frames_identical(buf.cpu().numpy(), ref, f"frame {k} on a stream of its own")
            assert hip.hipStreamDestroy(st) == 0


def test_every_path_hits_and_the_scatter_grid_is_prepared(sb, oracle_mod, scenes):
    # the path-traced pipeline's hit queues hold the worst case: a camera inside the solid, every camera ray a hit at its first
    # step; sdfhip_scene_prepare_path builds the bounce levels' grid ahead of the first path-traced render (same pixels either way)
    od = scenes["torus_d6"]
    W, H = 128, 96
    cam = sb.Logic(W, H); cam.Position = (0.5 + 0.25, 0.5, 0.5); cam.Heading = (0.3, 1.0)       # inside the tube of the torus
    pt = sb.PathTrace(spp=8, max_bounces=2)
    ref, _ = oracle_mod.render_pt(od.Structs, od.Values, cam.State, W, H, spp=8, max_bounces=2, nthreads=8)
    assert (ref[..., 3] > 0).all()
    with sb.Scene(od) as sc:
        before = sc.top_grid_bytes
        sb._lib.check(sb._lib.lib.sdfhip_scene_prepare_path(sc._h))
        sb._lib.check(sb._lib.lib.sdfhip_scene_prepare_path(sc._h))                  # again: nothing happens
        import ctypes
        nb = ctypes.c_uint64()
        sb._lib.check(sb._lib.lib.sdfhip_scene_top_grid(sc._h, None, ctypes.byref(nb)))
        assert nb.value > before                                                     # the second grid is there before any render
        assert_frames_identical(sc.DrawPath(cam, W, H, pt=pt), ref, "every path hits, prepared scatter grid")
    with sb.Scene(od) as sc:                                                         # ... and built by the first render instead
        assert_frames_identical(sc.DrawPath(cam, W, H, pt=pt), ref, "every path hits")


def test_first_frame_on_a_fresh_handle(sb, oracle_mod, scenes):
    # The very first render on a new handle meets freshly allocated scratch (queue fill counts, tile-queue heads): it must
    # be zeroed in the stream that uses it.  (A hipMemset on the null stream is not ordered against the handle's
    # non-blocking stream: now and then the first frame met counters that were not zero yet and lost hit pixels.)
    od = scenes["torus_d6"]
    cam = make_camera("closeup", 256, 192)
    ref, _ = oracle_mod.render(od.Structs, od.Values, cam.State, 256, 192, nthreads=8)
    for rep in range(2):        # (one handle per flag would do: the cause -- a memset on the null stream -- is gone by construction)
        for flags in (sb.KERNEL_STACK, sb.KERNEL_STACK | sb.FLAG_COMPACT, sb.KERNEL_STACK | sb.FLAG_COUNT) + ab(sb, sb.KERNEL_STACK | sb._lib.TUNE_SHADOW_QUEUE):
            with sb.Scene(od) as sc:
                assert_frames_identical(sc.Draw(cam, 256, 192, flags), ref, f"first frame, flags {flags:#x}, handle {rep}")


# ---- BASELINE.json full sizes: size-independent properties + sampled oracle rows ----
@pytest.fixture(scope="module")
def dragon(sb):
    od = sb.dragon_standin(9)
    sc = sb.Scene(od)
    yield od, sc
    sc.close()


@pytest.mark.parametrize("size,view", [((1920, 1080), "bench"), ((3840, 2160), "bench"), ((1920, 1080), "closeup")])
def test_full_size_properties(sb, oracle_mod, dragon, size, view):
    od, sc = dragon
    W, H = size
    assert sc.stack_kernel_ok and sc.depth == 9
    cam = sb.Logic(W, H)
    if view == "bench":
        cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)   # cfg-2/3 camera
    else:
        cam.Position = (0.5, 0.5, 0.02)                                 # the object fills the frame
    imgs, stats = {}, {}
    for variant in variants_of(sb):
        imgs[variant], stats[variant] = sc.Draw(cam, W, H, flags_of(sb, variant) | sb.FLAG_COUNT, want_stats=True)
    base = imgs["generic"]
    # (1) every kernel variant produces the same bits, and the same algorithmic counters
    for variant in variants_of(sb)[1:]:
        assert_frames_identical(imgs[variant], base, f"{variant} vs generic at {W}x{H}")
        for f in ("n_nodes", "n_samples", "n_steps", "n_shadow_rays"):
            assert getattr(stats[variant], f) == getattr(stats["generic"], f)
    # (2) alpha is the step count: integers in [0, 140], summing to the step counter;
    #     one sample per step
    a = base[..., 3]
    assert (a == np.floor(a)).all() and a.min() >= 0 and a.max() <= 140
    assert int(a.astype(np.float64).sum()) == stats["generic"].n_steps == stats["generic"].n_samples
    # (3) colours are grey, the sky constant, or NaN in all three channels (a march that
    #     ends in a cell of 8 equal bytes has a zero gradient: normalize(0), as in the shader)
    sky = (base[..., 0] == np.float32(0.005)) & (base[..., 1] == np.float32(0.01)) & (base[..., 2] == np.float32(0.2))
    grey = (base[..., 0] == base[..., 1]) & (base[..., 1] == base[..., 2])
    nan3 = np.isnan(base[..., :3]).all(axis=-1)
    assert (sky | grey | nan3).all() and (sky.any() or view == "closeup") and (grey & (base[..., 0] > 0.0051)).any()
    assert nan3.mean() < 0.01
    # (4) deterministic: a second render gives the same digest
    again = sc.Draw(cam, W, H, flags_of(sb, "stack"))
    assert hashlib.sha256(again.tobytes()).digest() == hashlib.sha256(imgs["stack"].tobytes()).digest()
    # (5) the oracle on sampled 8-row bands of the same frame (seconds of CPU time)
    for y0 in list(range(0, H - 8, H // 6)) + [H - 8]:
        ref, _ = oracle_mod.render(od.Structs, od.Values, cam.State, W, H, row0=y0, nrows=8, nthreads=16)
        assert_frames_identical(base[y0:y0 + 8], ref, f"rows {y0}..{y0 + 7} of {W}x{H}")


def test_bench_frame_equals_the_oracle_on_every_pixel(sb, oracle_mod, dragon):
    # BASELINE cfg-2 exactly as bench.py renders it -- 1920x1080, the depth-9 stand-in, the bench
    # camera -- against the oracle on all 2 073 600 pixels, with the four algorithmic counters
    od, sc = dragon
    W, H = 1920, 1080
    cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
    ref, cnt = oracle_mod.render(od.Structs, od.Values, cam.State, W, H, nthreads=os.cpu_count() or 8)
    img, st = sc.Draw(cam, W, H, sb.FLAG_COUNT, want_stats=True)
    assert_frames_identical(img, ref, "bench frame")
    assert (st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays) == tuple(int(c) for c in cnt)
    assert st.n_nodes == 433305936 and st.n_samples == 52988750          # the figures DESIGN.md section 6 quotes


def test_cfg3_frame_equals_the_oracle_on_every_pixel(sb, oracle_mod, dragon):
    # BASELINE cfg-3 / cfg-4's frame -- 3840x2160, the depth-9 stand-in, the bench camera -- against the oracle on all 8 294 400
    # pixels (a second of CPU time on the box's 16 CPUs), with the default kernel and with wavefront ray compaction on, and the
    # four algorithmic counters; the 4K frame sharded over repeated device lists is held against it in tests/test_multi.py
    od, sc = dragon
    W, H = 3840, 2160
    cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
    ref, cnt = oracle_mod.render(od.Structs, od.Values, cam.State, W, H, nthreads=min(64, os.cpu_count() or 8))
    img, st = sc.Draw(cam, W, H, sb.FLAG_COUNT, want_stats=True)
    assert_frames_identical(img, ref, "cfg-3 frame, default kernel")
    assert (st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays) == tuple(int(c) for c in cnt)
    assert_frames_identical(sc.Draw(cam, W, H, sb.FLAG_COMPACT), ref, "cfg-3 frame, SDFHIP_FLAG_COMPACT")
    assert_frames_identical(sc.Draw(cam, W, H, sb.FLAG_TILE_ORDER), ref, "cfg-3 frame, tile-order flag (ignored above 65 536 tiles)")


def test_host_frames_in_page_locked_memory(sb, dragon):
    # sdfhip_host_alloc / sdfhip_host_register: below 4 M pixels sdfhip_render's kernel stores its pixels straight into the
    # host's array (no device frame, no copy), above it the band copies go into it.  Both must leave the frame a pageable
    # destination gets -- every byte, NaN pixels included -- for a moving camera (the launch order of the tiles is the last
    # frame's), for RGBA32F and the fused display pass, for a window inside a larger registered array, and after release.
    import ctypes
    od, sc = dragon
    L = sb._lib
    for W, H in ((1920, 1080), (3840, 2160), (333, 211)):
        cams = []
        for k in range(3):
            c = sb.Logic(W, H); c.Position = (0.5 + 0.02 * k, 0.5, -0.35 + 0.05 * k); c.Heading = (-0.2 + 0.1 * k, 0.35)
            cams.append(c)
        plain = [sc.Draw(c, W, H).copy() for c in cams]
        plain8 = [sc.DrawDisplay(c, W, H).copy() for c in cams]
        with sb.HostFrame(H, W, np.float32) as hf, sb.HostFrame(H, W, np.uint8) as hf8:
            for k, c in enumerate(cams):
                hf.array[...] = np.float32(-1.0); hf8.array[...] = 7
                assert sc.Draw(c, W, H, out=hf.array) is hf.array
                assert_frames_identical(hf.array, plain[k], f"library-allocated host frame, {W}x{H}, camera {k}")
                sc.DrawDisplay(c, W, H, out=hf8.array)
                assert np.array_equal(hf8.array, plain8[k])
        own = np.full((H + 16, W, 4), -1.0, dtype=np.float32)
        with sb.HostFrame(array=own):
            window = own[8:8 + H]                                  # a frame somewhere inside the registered range
            for k, c in enumerate(cams):
                sc.Draw(c, W, H, out=window)
                assert_frames_identical(window, plain[k], f"registered host frame, {W}x{H}, camera {k}")
            assert (own[:8] == -1.0).all() and (own[8 + H:] == -1.0).all()
        sc.Draw(cams[0], W, H, out=own[:H])                        # released: an ordinary destination again
        assert_frames_identical(own[:H], plain[0], "after release")
    assert L.lib.sdfhip_host_release(ctypes.c_void_p(own.ctypes.data)) == L.ERR_ARG and b"neither" in L.lib.sdfhip_last_error()
    assert L.lib.sdfhip_host_release(None) == L.OK
    assert L.lib.sdfhip_host_register(None, 16) == L.ERR_ARG and L.lib.sdfhip_host_alloc(0, ctypes.byref(ctypes.c_void_p())) == L.ERR_ARG


# ---- fuzz: random trees, on-grid cameras, axis-aligned rays ---------------------------------
def _random_tree(rng, max_depth, p_split, max_nodes=60000):
    """A consistent octree with random splits (DFS pre-order, like SdfGen) and random bytes."""
    structs = [[-1, -1]]

    def grow(node, depth):
        if depth >= max_depth or len(structs) + 8 > max_nodes or rng.random() > p_split:
            return
        c = len(structs)
        structs[node][1] = c
        for _ in range(8):
            structs.append([node, -1])
        for k in range(8):
            grow(c + k, depth + 1)

    grow(0, 0)
    s = np.array(structs, dtype=np.int32)
    mode = rng.integers(3)
    if mode == 0:
        v = rng.integers(0, 256, size=(len(s), 8), dtype=np.uint8)
    elif mode == 1:      # mostly flat cells (exercises the flat fast path next to non-flat lanes)
        v = np.repeat(rng.integers(0, 256, size=(len(s), 1), dtype=np.uint8), 8, axis=1)
        noisy = rng.random(len(s)) < 0.2
        v[noisy] = rng.integers(0, 256, size=(int(noisy.sum()), 8), dtype=np.uint8)
    else:                # a crude distance-like field: larger values near the cube faces
        v = rng.integers(60, 200, size=(len(s), 8), dtype=np.uint8)
    return s, v


def test_nan_coordinates_select_the_low_cells(sb, oracle_mod):
    # A NaN coordinate fails every comparison of inside() (up to the root) and saturates to 0 in the descent
    # (Compute.hlsl:93-106): the shader lands in the cell at the LOW end of that axis.  On a tree whose corners hold
    # different values the two ends give different distances -- v_cvt_flr_i32_f32 turns NaN into INT_MAX, which a
    # clamp would send to the HIGH end (found by the fuzz campaign, seed 26, through NaN bounce directions).
    rng = np.random.default_rng(77)
    s, v = _random_tree(rng, 5, p_split=0.75)
    v = rng.integers(0, 256, size=v.shape, dtype=np.uint8)
    od = sb.OctData(s, v)
    W, H = 48, 40
    cams = []
    for axes in ((0,), (1,), (2,), (0, 2), (0, 1, 2)):
        c = sb.Logic(W, H); c.Position = (0.3, 0.6, -0.2); c.Heading = (0.2, 0.4)
        for a in axes:
            c.State.position[a] = float("nan")
        cams.append(c)
    c = sb.Logic(W, H); c.Position = (0.4, 0.5, 0.3); c.State.heading[0][0] = float("nan"); cams.append(c)      # NaN ray directions
    with sb.Scene(od) as sc:
        for ci, cam in enumerate(cams):
            ref, cnt = oracle_mod.render(s, v, cam.State, W, H, nthreads=4)
            for variant in variants_of(sb):
                img, st = sc.Draw(cam, W, H, flags_of(sb, variant) | sb.FLAG_COUNT, want_stats=True)
                assert_frames_identical(img, ref, f"NaN camera {ci} {variant}")
                assert (st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays) == tuple(int(x) for x in cnt), (ci, variant)
                assert_frames_identical(sc.Draw(cam, W, H, flags_of(sb, variant)), ref, f"NaN camera {ci} {variant}, not counting")
            pref, _ = oracle_mod.render_pt(s, v, cam.State, W, H, spp=2, nthreads=4)
            for fl in (sb.KERNEL_STACK, sb.KERNEL_GENERIC) + ab(sb, sb.KERNEL_STACK | sb.TUNE_ONE_KERNEL):
                assert_frames_identical(sc.DrawPath(cam, W, H, pt=sb.PathTrace(spp=2), flags=fl), pref, f"NaN camera {ci} path-traced {fl:#x}")


@pytest.mark.parametrize("seed", range(int(os.environ.get("SDFHIP_FUZZ_SEEDS", "4"))))
def test_pre_decoded_cells_of_split_grids(sb, oracle_mod, seed):
    lab_only(sb)
    # SDFHIP_SAMPLE_RECORDS=1 (an experiment, measured slower: DESIGN.md section 4.3): the default kernel reads the grid's second form,
    # one 4-byte word per cell of the deepest level + 64-byte sample records with pre-decoded corners (CursorFF).  Random trees with
    # random bytes -- non-flat leaves at every level -- behind dense grids and split grids with blocks of 1 to 4 levels, cameras
    # inside, outside, on cell faces, at tiny and denormal coordinates and at NaN; against the oracle, and the 16-byte cells (the
    # A/B knob SDFHIP_TUNE_BYTE_CELLS) beside it
    rng = np.random.default_rng(4000 + seed)
    depth = [6, 7, 8, 9][seed % 4]
    s, v = _random_tree(rng, depth, p_split=[0.7, 0.6, 0.5, 0.45][seed % 4])
    if seed % 2:
        v = rng.integers(0, 256, size=v.shape, dtype=np.uint8)              # no flat cell anywhere
    od = sb.OctData(s, v)
    W, H = 64, 48
    cams = []
    for kind in range(7):
        cam = sb.Logic(W, H)
        if kind == 0:
            cam.Position = tuple(float(x) for x in rng.uniform(0.05, 0.95, 3))
        elif kind == 1:
            cam.Position = tuple(float(x) for x in rng.uniform(-0.6, 1.6, 3))
        elif kind == 2:
            k = int(rng.integers(1, 10))
            cam.Position = tuple(float(rng.integers(0, 2 ** k + 1)) / 2 ** k for _ in range(3))
        elif kind == 3:
            cam.Position = (1e-42, 0.5, 3e-39)                               # denormal coordinates inside the cube
        elif kind == 4:
            cam.Position = (-1e-42, 1.0, 0.99999994)                         # a denormal below zero; the far faces
        elif kind == 5:
            cam.Position = (0.5, 0.25, -0.25); cam.State.fov = 0.0           # every pixel the same axis ray
        else:
            cam.Position = (0.3, 0.6, 0.2); cam.State.position[1] = float("nan")
        if kind != 5:
            cam.Heading = (float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-3, 3)))
        cams.append(cam)
    os.environ["SDFHIP_SAMPLE_RECORDS"] = "1"
    try:
        for fb in (0, 1, 2, 3, 4):
            if depth - fb < 1:
                continue
            # (fb 0: the tree's default grid, dense up to depth 8)
            with sb.Scene(od, top_grid_split=(depth - fb) if fb else None) as sc:
                if fb and sc.depth <= depth - fb:
                    continue
                if fb:
                    assert sc.top_grid_level == depth - fb
                if 1 <= sc.depth <= 10 and sc.top_grid_level + (fb if fb else 0) == sc.depth:
                    assert sc.top_grid_bytes > (16 << (3 * sc.top_grid_level)) + (4 << (3 * sc.depth))     # the second form is there
                for ci, cam in enumerate(cams):
                    ref, _ = oracle_mod.render(s, v, cam.State, W, H, nthreads=8)
                    assert_frames_identical(sc.Draw(cam, W, H), ref, f"seed {seed} blocks of {fb} levels, camera {ci}: pre-decoded cells")
                    assert_frames_identical(sc.Draw(cam, W, H, sb._lib.TUNE_BYTE_CELLS), ref, f"seed {seed} blocks of {fb} levels, camera {ci}: byte cells")
                    assert_frames_identical(sc.Draw(cam, W, H, sb.FLAG_TILE_ORDER), ref, "tile order")
                    assert (sc.DrawDisplay(cam, W, H) == sc.DrawDisplay(cam, W, H, flags=sb._lib.TUNE_BYTE_CELLS)).all()
    finally:
        os.environ.pop("SDFHIP_SAMPLE_RECORDS", None)


# SDFHIP_FUZZ_SEEDS=n runs n seeds instead of 6 (a one-off campaign; the committed default stays small); SDFHIP_FUZZ_FIRST=f: the
# n seeds f .. f + n - 1 (a second campaign over seeds the first one did not see)
_FUZZ_FIRST = int(os.environ.get("SDFHIP_FUZZ_FIRST", "0"))


@pytest.mark.parametrize("seed", sorted(set(range(_FUZZ_FIRST, _FUZZ_FIRST + int(os.environ.get("SDFHIP_FUZZ_SEEDS", "6")))) | {26}))
def test_fuzz_random_trees_and_on_grid_cameras(sb, oracle_mod, seed):
    rng = np.random.default_rng(1000 + seed)
    depth = [3, 5, 7, 9, 11, 12][seed % 6]
    s, v = _random_tree(rng, depth, p_split=[0.9, 0.7, 0.55, 0.45, 0.42, 0.4][seed % 6])
    od = sb.OctData(s, v)
    W, H = 72, 56
    cams = []
    for _ in range(5):
        cam = sb.Logic(W, H)
        kind = rng.integers(4)
        if kind == 0:      # position exactly on the 2^-k grid (cell faces / corners), generic heading
            k = int(rng.integers(1, 13))
            cam.Position = tuple(float(rng.integers(0, 2 ** k + 1)) / 2 ** k for _ in range(3))
            cam.Heading = (float(rng.uniform(-1, 1)), float(rng.uniform(-3, 3)))
        elif kind == 1:    # axis-aligned centre ray from a grid point: positions stay on grid lines
            k = int(rng.integers(1, 8))
            cam.Position = (float(rng.integers(0, 2 ** k + 1)) / 2 ** k, float(rng.integers(0, 2 ** k + 1)) / 2 ** k, -0.25)
            cam.State.fov = 0.0                     # every pixel shoots the same axis ray (0, 0, 1)
        elif kind == 2:    # outside the cube, looking in
            cam.Position = tuple(float(x) for x in rng.uniform(-0.6, 1.6, 3))
            cam.Heading = (float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-3, 3)))
        else:              # inside, light moved, margins changed
            cam.Position = tuple(float(x) for x in rng.uniform(0.05, 0.95, 3))
            cam.Heading = (float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-3, 3)))
            for i in range(3):
                cam.State.light[i] = float(rng.uniform(-0.5, 1.5))
            cam.State.margin = float(rng.choice([1e-5, 4e-4, 3e-3]))
        cams.append(cam)
    with sb.Scene(od) as sc:
        assert sc.stack_kernel_ok
        for ci, cam in enumerate(cams):
            ref, cnt = oracle_mod.render(s, v, cam.State, W, H, nthreads=8)
            for variant in variants_of(sb):
                img, st = sc.Draw(cam, W, H, flags_of(sb, variant) | sb.FLAG_COUNT, want_stats=True)
                assert_frames_identical(img, ref, f"seed {seed} cam {ci} {variant}")
                assert (st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays) == tuple(int(c) for c in cnt), (seed, ci, variant)
                # the kernels that do not count are separate instances (and the timed ones)
                assert_frames_identical(sc.Draw(cam, W, H, flags_of(sb, variant)), ref, f"seed {seed} cam {ci} {variant}, not counting")
            if ci == 0:      # the path-traced mode on the same tree (both find() forms)
                pref, pcnt = oracle_mod.render_pt(s, v, cam.State, W, H, spp=2, nthreads=8)
                for kern in (sb.KERNEL_STACK, sb.KERNEL_GENERIC):
                    pimg, pst = sc.DrawPath(cam, W, H, pt=sb.PathTrace(spp=2), flags=kern | sb.FLAG_COUNT, want_stats=True)
                    assert_frames_identical(pimg, pref, f"seed {seed} path-traced, kernel {kern}")
                    assert (pst.n_nodes, pst.n_samples, pst.n_steps, pst.n_shadow_rays) == tuple(int(c) for c in pcnt), (seed, kern)
                    assert_frames_identical(sc.DrawPath(cam, W, H, pt=sb.PathTrace(spp=2), flags=kern), pref, f"seed {seed} path-traced, not counting")
    if depth >= 7:           # the same tree behind a split grid (what trees of depth 10-12 get when they are large)
        coarse = min(8, depth - int(rng.integers(1, 5)))
        with sb.Scene(od, top_grid_split=coarse) as sc:
            if sc.depth > coarse:     # a random tree may be shallower than asked
                for ci, cam in enumerate(cams[:3]):
                    ref, cnt = oracle_mod.render(s, v, cam.State, W, H, nthreads=8)
                    img, st = sc.Draw(cam, W, H, sb.FLAG_COUNT, want_stats=True)
                    assert_frames_identical(img, ref, f"seed {seed} cam {ci} split grid")
                    assert (st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays) == tuple(int(c) for c in cnt), (seed, ci)
                    assert_frames_identical(sc.Draw(cam, W, H), ref, f"seed {seed} cam {ci} split grid, not counting")


def test_upload_options_are_arguments_and_the_environment_is_the_laboratorys(sb, scenes, monkeypatch):
    # sdfhip_scene_upload_ex: the grid choices as arguments (checked: size, ranges); the measurement knobs of the environment
    # (SDFHIP_TOP_GRID_LEVEL ...) are defaults of those options in the laboratory library only -- the product does not read them
    import ctypes
    od = scenes["torus_d6"]
    with sb.Scene(od) as sc:
        assert sc.top_grid_level == sc.depth == 6                       # the default for a shallow tree: a dense grid as deep as the tree
    with sb.Scene(od, top_grid_level=3) as sc:
        assert sc.top_grid_level == 3 and sc.top_grid_bytes == 16 << 9
    with sb.Scene(od, top_grid_level=0) as sc:
        assert sc.top_grid_level == 0 and sc.top_grid_bytes == 0
    with sb.Scene(od, top_grid_split=4) as sc:
        assert sc.top_grid_level == 4 and sc.top_grid_bytes > 16 << 12
    monkeypatch.setenv("SDFHIP_TOP_GRID_LEVEL", "2")
    with sb.Scene(od) as sc:
        assert sc.top_grid_level == (2 if sb._lib.EXPERIMENTS else 6)
    with sb.Scene(od, top_grid_level=4) as sc:                          # an argument goes before the environment
        assert sc.top_grid_level == 4
    monkeypatch.delenv("SDFHIP_TOP_GRID_LEVEL")
    L = sb._lib
    h = ctypes.c_void_p()
    opt = L.UploadOptions(top_grid_level=11)
    assert L.lib.sdfhip_scene_upload_ex(0, od.Structs.ctypes.data, od.Values.ctypes.data, od.Length, ctypes.byref(opt), ctypes.byref(h)) == L.ERR_ARG
    opt = L.UploadOptions(); opt.size = 8
    assert L.lib.sdfhip_scene_upload_ex(0, od.Structs.ctypes.data, od.Values.ctypes.data, od.Length, ctypes.byref(opt), ctypes.byref(h)) == L.ERR_ARG
    # `size` lets the struct grow: a caller built against a NEWER header (a larger struct) is accepted while the fields this library
    # does not know say "choose" (-1), and refused when one of them asks for something

    class NewerOptions(ctypes.Structure):
        _fields_ = [("size", ctypes.c_uint32), ("top_grid_level", ctypes.c_int32), ("top_grid_split", ctypes.c_int32),
                    ("scatter_grid", ctypes.c_int32), ("scatter_order", ctypes.c_int32), ("a_later_field", ctypes.c_int32), ("another", ctypes.c_int32)]
    newer = NewerOptions(ctypes.sizeof(NewerOptions), 3, -1, -1, -1, -1, -1)
    as_ours = ctypes.cast(ctypes.pointer(newer), ctypes.POINTER(L.UploadOptions))      # (the binding names this library's struct)
    assert L.lib.sdfhip_scene_upload_ex(0, od.Structs.ctypes.data, od.Values.ctypes.data, od.Length, as_ours, ctypes.byref(h)) == L.OK
    level, nbytes = ctypes.c_int32(), ctypes.c_uint64()
    assert L.lib.sdfhip_scene_top_grid(h, ctypes.byref(level), ctypes.byref(nbytes)) == L.OK and level.value == 3
    assert L.lib.sdfhip_scene_free(h) == L.OK
    newer.another = 1
    assert L.lib.sdfhip_scene_upload_ex(0, od.Structs.ctypes.data, od.Values.ctypes.data, od.Length, as_ours, ctypes.byref(h)) == L.ERR_ARG
    d = L.UploadOptions(1, 2, 3, 0)
    L.lib.sdfhip_upload_options_default(ctypes.byref(d))
    assert (d.size, d.top_grid_level, d.top_grid_split, d.scatter_grid, d.scatter_order) == (ctypes.sizeof(L.UploadOptions), -1, -1, -1, -1)
