"""Pins the CPU oracle: (1) bit-for-bit against an independent numpy-float32
restatement of Compute.hlsl that goes through an emulated R8 texture with the
reference's swizzle; (2) against closed-form facts of a sphere; (3) against the
committed golden frames.  The reference has no tests or vectors of its own
for this path."""
import os

import numpy as np
import pytest

from conftest import CAMERAS, GOLDEN, assert_frames_identical, bits_equal, make_camera


def test_c_oracle_equals_python_restatement(sb, oracle_mod, scenes):
    from py_restatement import Shader
    pixels = [(0, 0), (32, 32), (63, 63), (10, 50), (40, 20), (31, 5), (17, 33)]
    for sname, od in scenes.items():
        for cname in CAMERAS:
            cam = make_camera(cname, 64, 64)
            sh = Shader(od.Structs, od.Values, cam.State)
            for (x, y) in pixels[: 7 if sname == "sphere_d4" else 3]:
                o, c = oracle_mod.pixel(od.Structs, od.Values, cam.State, x, y)
                p = np.array(sh.main(x, y), dtype=np.float32)
                assert bits_equal(o, p).all(), (sname, cname, x, y, o, p)
                assert (sh.nodes, sh.samples) == (int(c[0]), int(c[1])), (sname, cname, x, y)


def test_c_oracle_equals_vectorised_restatement_on_whole_frames(sb, oracle_mod, scenes):
    # The same independent restatement, vectorised over pixels (tests/py_restatement_vec.py): EVERY pixel of every
    # 64x64 scene/camera frame (4 096 each: silhouettes, shadowed and back-facing pixels, 100-140-step pixels, a NaN
    # grey) and of a 320x180 frame of the bench scene's shape (gyroid shell, depth 7) under the bench camera --
    # colours, step counts and the per-pixel algorithmic read counts, bit for bit.
    from py_restatement import Shader
    from py_restatement_vec import ShaderV
    seen = {"steps>=100": 0, "nan": 0, "shadowed": 0, "lit": 0, "sky": 0, "backfacing": 0}
    cases = [(sname, od, make_camera(cname, 64, 64), 64, 64) for sname, od in scenes.items() for cname in CAMERAS]
    gy = sb.dragon_standin(7, nthreads=4)
    cam = sb.Logic(320, 180); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
    cases.append(("gyroid_d7", gy, cam, 320, 180))
    for sname, od, cam, W, H in cases:
        ys, xs = np.mgrid[0:H, 0:W]
        sh = ShaderV(od.Structs, od.Values, cam.State)
        out, nodes, samples = sh.main(xs.ravel(), ys.ravel())
        ref, cnt, pix = oracle_mod.render(od.Structs, od.Values, cam.State, W, H, nthreads=4, per_pixel_nodes=True)
        assert_frames_identical(out.reshape(H, W, 4), ref, f"vectorised restatement, {sname}")
        assert (nodes.reshape(H, W) == pix).all() and int(samples.sum()) == int(cnt[1]) and sh.shadow_rays == int(cnt[3])
        sky = (ref[..., 2] == np.float32(0.2)) & (ref[..., 0] == np.float32(0.005))
        seen["steps>=100"] += int((ref[..., 3] >= 100).sum()); seen["nan"] += int(np.isnan(ref[..., 0]).sum())
        seen["sky"] += int(sky.sum()); seen["lit"] += int((~sky & (ref[..., 0] > 0)).sum())
        seen["shadowed"] += int((~sky & (ref[..., 0] == 0)).sum())
    assert all(v > 0 for v in seen.values() if v is not seen["backfacing"]), seen
    # and the scalar twin (exact rational fma) agrees with the vectorised one on its handful of pixels
    od = scenes["sphere_d4"]; cam = make_camera("closeup", 64, 64)
    px = [(0, 0), (32, 32), (63, 63), (10, 50)]
    v, _, _ = ShaderV(od.Structs, od.Values, cam.State).main([p[0] for p in px], [p[1] for p in px])
    s1 = Shader(od.Structs, od.Values, cam.State)
    for k, (x, y) in enumerate(px):
        assert bits_equal(v[k], np.array(s1.main(x, y), dtype=np.float32)).all()


def test_oracle_reproduces_golden_frames(oracle_mod, scenes):
    g = np.load(os.path.join(GOLDEN, "frames.npz"))
    for sname, od in scenes.items():
        for cname in CAMERAS:
            cam = make_camera(cname, 64, 64)
            img, cnt = oracle_mod.render(od.Structs, od.Values, cam.State, 64, 64)
            assert_frames_identical(img, g[f"{sname}/{cname}/rgba"], f"{sname}/{cname}")
            assert cnt.tolist() == g[f"{sname}/{cname}/counters"].tolist()
            # alpha is the step count: its sum is the step counter
            assert int(img[..., 3].sum()) == int(cnt[2])


def test_oracle_reproduces_golden_path_traced_frames(oracle_mod, scenes):
    g = np.load(os.path.join(GOLDEN, "frames.npz"))
    for sname, od in scenes.items():
        cam = make_camera("default", 48, 32)
        img, cnt = oracle_mod.render_pt(od.Structs, od.Values, cam.State, 48, 32, spp=4, max_bounces=3, nthreads=4)
        assert_frames_identical(img, g[f"{sname}/path4/rgba"], f"{sname}/path4")
        assert cnt.tolist() == g[f"{sname}/path4/counters"].tolist()
        assert int(img[..., 3].astype(np.float64).sum()) == int(cnt[2])


def test_path_traced_mode_sanity(oracle_mod, scenes):
    od = scenes["sphere_d4"]
    cam = make_camera("default", 40, 30)
    direct, _ = oracle_mod.render(od.Structs, od.Values, cam.State, 40, 30)
    # 0 bounces, albedo 1: every sample is the direct-lighting result of its jittered ray, so the
    # mean stays close to main()'s image wherever that is smooth
    pt0, _ = oracle_mod.render_pt(od.Structs, od.Values, cam.State, 40, 30, spp=8, max_bounces=0, albedo=1.0)
    lit = (direct[..., 0] > 0.05) & (direct[..., 2] != np.float32(0.2))
    assert lit.sum() > 200
    rel = np.abs(pt0[..., 0][lit] - direct[..., 0][lit]) / direct[..., 0][lit]
    assert np.median(rel) < 0.1
    # bounces only add light (sky + indirect); the seed changes the noise, not the mean much
    pt3, _ = oracle_mod.render_pt(od.Structs, od.Values, cam.State, 40, 30, spp=8, max_bounces=3, albedo=1.0)
    assert np.nanmean(pt3[..., 2]) >= np.nanmean(pt0[..., 2])
    a, _ = oracle_mod.render_pt(od.Structs, od.Values, cam.State, 40, 30, spp=8, seed=1)
    b, _ = oracle_mod.render_pt(od.Structs, od.Values, cam.State, 40, 30, spp=8, seed=2)
    assert not np.array_equal(a, b) and abs(np.nanmean(a[..., :3]) - np.nanmean(b[..., :3])) < 0.02
    # rows and threads do not change pixels (the RNG is keyed by the pixel, not by the visit order)
    part, _ = oracle_mod.render_pt(od.Structs, od.Values, cam.State, 40, 30, spp=8, seed=1, row0=7, nrows=5, nthreads=3)
    assert_frames_identical(part, a[7:12], "path rows")


def test_threads_and_row_windows_do_not_change_pixels(oracle_mod, scenes):
    od = scenes["torus_d6"]
    cam = make_camera("rotated", 96, 40)
    full, cnt = oracle_mod.render(od.Structs, od.Values, cam.State, 96, 40)
    mt, cnt_mt = oracle_mod.render(od.Structs, od.Values, cam.State, 96, 40, nthreads=4)
    assert_frames_identical(mt, full, "threads")
    assert cnt.tolist() == cnt_mt.tolist()
    part, _ = oracle_mod.render(od.Structs, od.Values, cam.State, 96, 40, row0=13, nrows=9)
    assert_frames_identical(part, full[13:22], "row window")
    # the timed baseline's entry point (pinned pool, chunks of 64 pixels from one counter, clock inside): the same pixels and counts,
    # with more threads than this machine has CPUs, with and without storing the frame
    for nt in (1, 3, 40):
        img, c, sec, topo = oracle_mod.bench_rows(od.Structs, od.Values, cam.State, 96, 40, nthreads=nt)
        assert_frames_identical(img, full, f"bench_rows, {nt} threads")
        assert c.tolist() == cnt.tolist() and sec > 0 and topo["threads"] == nt and topo["numa_nodes"] >= 1
    img, c, _, _ = oracle_mod.bench_rows(od.Structs, od.Values, cam.State, 96, 40, row0=1, row_step=3, nthreads=2)
    assert_frames_identical(img, full[1::3], "bench_rows, every third row")
    none, c2, _, _ = oracle_mod.bench_rows(od.Structs, od.Values, cam.State, 96, 40, row0=1, row_step=3, nthreads=2, store=False)
    assert none is None and c2.tolist() == c.tolist()


def _leaf_corners(od, count, seed):
    """(x, y, z) of corner k of randomly chosen leaves, found by walking the tree."""
    s = od.Structs
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        i, lower, scale = 0, np.zeros(3), 1.0
        while s[i, 1] >= 0:
            k = int(rng.integers(8))
            scale /= 2
            lower = lower + np.array([k % 2, k // 2 % 2, k // 4 % 2]) * scale
            i = s[i, 1] + k
        k = int(rng.integers(8))
        out.append(lower + np.array([k % 2, k // 2 % 2, k // 4 % 2]) * scale)
    return out


def _torus(p):
    d = p - 0.5
    q = np.hypot(d[0], d[2]) - 0.25
    return np.hypot(q, d[1]) - 0.09


def test_corner_values_pin_layout_and_quantiser(oracle_mod, scenes):
    # At a cell corner the trilinear weights are 0/1, so find() + interpol_world must
    # return that corner's dequantised byte: floor-quantised (FromFloat, dllmain.cpp:
    # 192-196) it sits within one step 2s/255 below the true distance, or at the clamp
    # -0.5 s / 1.5 s.  The torus (axis y) is not symmetric under axis swaps, so this pins
    # the corner order k = x + 2y + 4z and the lerp axis order too.
    for od, sdf in [(scenes["sphere_d4"], lambda p: np.linalg.norm(p - 0.5) - 0.3), (scenes["torus_d6"], _torus)]:
        checked = 0
        for p in _leaf_corners(od, 400, 11):
            d, idx, scale = oracle_mod.distance_at(od.Structs, od.Values, *p)
            if not np.allclose(p / scale, np.round(p / scale)):
                continue          # find() chose a coarser neighbour: p is on its face, not a corner
            true = min(max(sdf(p), -0.5 * scale), 1.5 * scale)
            assert true - 2 * scale / 255 - 1e-6 <= d <= true + 1e-6, (p, d, true, scale)
            checked += 1
        assert checked > 300


def test_interpolation_between_corners(oracle_mod, scenes):
    # inside a leaf the value is the trilinear blend of its 8 corners: compare with a
    # float64 trilinear of the dequantised bytes at random interior points
    od = scenes["torus_d6"]
    s, v = od.Structs, od.Values
    rng = np.random.default_rng(5)
    for _ in range(300):
        i, lower, scale = 0, np.zeros(3), 1.0
        while s[i, 1] >= 0:
            k = int(rng.integers(8)); scale /= 2
            lower = lower + np.array([k % 2, k // 2 % 2, k // 4 % 2]) * scale
            i = s[i, 1] + k
        t = rng.uniform(0.05, 0.95, 3)
        p = lower + t * scale
        d, idx, sc = oracle_mod.distance_at(s, v, *p)
        assert idx == i and sc == scale
        c = v[i].astype(np.float64) / 255
        x0 = c[0::2] * (1 - t[0]) + c[1::2] * t[0]          # pairs (0,1) (2,3) (4,5) (6,7)
        y0 = x0[0::2] * (1 - t[1]) + x0[1::2] * t[1]
        val = (y0[0] * (1 - t[2]) + y0[1] * t[2] - 0.25) * scale * 2
        assert abs(d - val) < 1e-6


def test_range_clamps(oracle_mod, scenes):
    od = scenes["sphere_d4"]
    d, _, scale = oracle_mod.distance_at(od.Structs, od.Values, 0.1, 0.1, 0.1)
    assert d == pytest.approx(1.5 * scale)
    d, _, scale = oracle_mod.distance_at(od.Structs, od.Values, 0.5, 0.5, 0.5)
    assert d == pytest.approx(-0.5 * scale)


def test_centre_ray_hits_the_sphere_where_it_should(sb, oracle_mod, scenes):
    od = scenes["sphere_d4"]
    W = H = 65                                         # odd size: pixel 32 is almost the axis
    cam = sb.Logic(W, H)                               # camera (0.5,0.5,0.1) looking down +z
    img, _ = oracle_mod.render(od.Structs, od.Values, cam.State, W, H)
    centre = img[32, 32]
    assert centre[0] > 0.0051 and centre[0] == centre[1] == centre[2]      # lit, grey
    # expected shade at the analytic hit point (0.5,0.5,0.2), normal (0,0,-1), light at 0:
    # the reference's normal comes from raw 8-bit cell values, so allow a wide band
    L = np.array([-0.5, -0.5, -0.2]); dist = np.linalg.norm(L) / 2
    ideal = (0.2 / np.linalg.norm(L)) / dist ** 2 * (2 ** 0.2 - 1)
    assert 0.3 * ideal < centre[0] < 1.5 * ideal
    # sky pixels: exactly the shader's constant, steps > 0
    corner = img[0, 0]
    assert corner[:3].tolist() == [np.float32(0.005), np.float32(0.01), np.float32(0.2)]
    assert 1 <= corner[3] <= 100
    # the image of the sphere is symmetric under x <-> y only where the light is too
    # (light at the origin, camera on the diagonal plane): transpose symmetry
    assert np.allclose(img[..., 0], img[..., 0].T, atol=2e-3)


def test_hit_depth_along_the_axis(sb, oracle_mod, scenes):
    # march the axis ray by hand with the oracle's distance: it must stop at the true
    # surface z = 0.2 to within 2*margin plus the zero-crossing error of a depth-4 cell
    # whose inner corners are clamped at -0.5 s (measured 0.005 = 0.08 s; allow 0.1 s)
    od = scenes["sphere_d4"]
    z, steps = np.float32(0.1), 0
    while steps < 100:
        d, _, scale = oracle_mod.distance_at(od.Structs, od.Values, 0.5, 0.5, float(z))
        if not (d > 2 * 0.0004 or d < 0):
            break
        z = np.float32(z + np.float32(d)); steps += 1
    assert abs(float(z) - 0.2) < 2 * 0.0004 + 0.1 / 16
    assert steps < 40


def test_unorm_table_is_byte_over_255(oracle_mod):
    t = oracle_mod.unorm_table()
    assert t[0] == 0 and t[255] == 1
    assert (t == (np.arange(256, dtype=np.float32) / np.float32(255))).all()


def test_display_pass_rules(oracle_mod):
    # DisplayFrag.hlsl: pow(val, 1/2.2) then R8G8B8A8_UNorm; debug: (1,1,1,0) * w / 140
    px = np.array([[[0.0, 1.0, 0.2, 26.0], [0.005, 0.01, 0.2, 0.0], [np.nan, -1.0, 2.0, 140.0],
                    [0.5, 0.5, 0.5, 70.0]]], dtype=np.float32)
    out = oracle_mod.display(px)
    assert out[0, 0].tolist() == [0, 255, round(0.2 ** (1 / 2.2) * 255), 255]
    assert out[0, 1].tolist() == [round(0.005 ** (1 / 2.2) * 255), round(0.01 ** (1 / 2.2) * 255), 123, 0]
    assert out[0, 2].tolist() == [0, 0, 255, 255]            # NaN -> 0, negative -> NaN -> 0, > 1 clamps
    assert out[0, 3].tolist() == [186, 186, 186, 255]
    heat = oracle_mod.display(px, debug=True)
    assert heat[0, 0].tolist() == [round(26 / 140 * 255)] * 3 + [0]
    assert heat[0, 2].tolist() == [255, 255, 255, 0] and heat[0, 1].tolist() == [0, 0, 0, 0]
