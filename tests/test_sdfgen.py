"""N1: point cloud -> ASDF.  CPU: the readers and the SdfGen oracle (the restatement of
SdfGen/dllmain.cpp) against the facts SURVEY.md recorded from a real SdfGen run.
GPU: the level-synchronous HIP builder against the oracle, byte for byte."""
import os
import struct

import numpy as np
import pytest


def fib_sphere(n, r=0.5, centre=(0.0, 0.0, 0.0)):
    i = np.arange(n) + 0.5
    phi = np.arccos(1 - 2 * i / n)
    th = np.pi * (1 + 5 ** 0.5) * i
    p = np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], 1)
    return np.concatenate([p * r + np.asarray(centre), p], 1).astype(np.float32)


def torus_cloud(n, R=0.6, r=0.2, seed=0):
    rng = np.random.default_rng(seed)
    u, v = rng.uniform(0, 2 * np.pi, n), rng.uniform(0, 2 * np.pi, n)
    nrm = np.stack([np.cos(u) * np.cos(v), np.sin(v), np.sin(u) * np.cos(v)], 1)
    pos = np.stack([np.cos(u) * R, np.zeros(n), np.sin(u) * R], 1) + nrm * r
    return np.concatenate([pos, nrm], 1).astype(np.float32)


def write_ply(path, verts, fmt="binary_little_endian"):
    with open(path, "wb") as f:
        f.write((f"ply\nformat {fmt} 1.0\ncomment made by tests\nelement vertex {len(verts)}\n"
                 "property float x\nproperty float y\nproperty float z\n"
                 "property float nx\nproperty float ny\nproperty float nz\nend_header\n").encode())
        f.write(np.ascontiguousarray(verts, dtype="<f4").tobytes())


def levels_of(structs):
    lvl = np.zeros(len(structs), dtype=np.int32)
    for i in range(len(structs)):
        c = structs[i, 1]
        if c >= 0:
            lvl[c:c + 8] = lvl[i] + 1
    return lvl


# ---- readers ------------------------------------------------------------------------------
def test_ply_reader(sb, tmp_path):
    v = fib_sphere(1234)
    p = tmp_path / "s.ply"
    write_ply(p, v)
    got = sb.OctData.LoadPly(str(p))
    assert got.shape == (1234, 6) and (got == v).all()
    for fmt in ("ascii", "binary_big_endian"):                     # refused, as in ply_reader.cpp:47-54
        write_ply(p, v, fmt)
        with pytest.raises(sb.SdfHipError):
            sb.OctData.LoadPly(str(p))
    p.write_bytes(b"plx\n")
    with pytest.raises(sb.SdfHipError):
        sb.OctData.LoadPly(str(p))
    write_ply(p, v)
    p.write_bytes(p.read_bytes()[:-100])                           # truncated payload
    with pytest.raises(sb.SdfHipError):
        sb.OctData.LoadPly(str(p))
    with pytest.raises(sb.SdfHipError):
        sb.OctData.LoadPly(str(tmp_path / "missing.ply"))


def test_obj_reader(sb, tmp_path):
    p = tmp_path / "m.obj"
    p.write_text("# a comment\no thing\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0.5 0.5 1\nvt 0 0\nvn 0 0 1\nvn 1 0 0\ns off\n"
                 "f 1/1/1 2/1/1 3/1/1\nf 2//2 3//2\n")
    got = sb.OctData.LoadObj(str(p))
    assert got.shape == (4, 6)
    assert got[:, :3].tolist() == [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0.5, 0.5, 1]]
    # a face assigns its normals to its vertices (later faces win); unmentioned vertices keep (0,0,0)
    assert got[:, 3:].tolist() == [[0, 0, 1], [1, 0, 0], [1, 0, 0], [0, 0, 0]]
    p.write_text("v 0 0 0\nvn 0 0 1\nf 1/1/2\n")                   # normal index out of range
    with pytest.raises(sb.SdfHipError):
        sb.OctData.LoadObj(str(p))
    p.write_text("v 0 0 0\nq nonsense\n")
    with pytest.raises(sb.SdfHipError):
        sb.OctData.LoadObj(str(p))


# ---- the SdfGen oracle against the reference's recorded run -----------------------------------
def test_oracle_reproduces_the_survey_run(oracle_mod):
    # SURVEY.md / BASELINE.md: sphere point cloud r = 0.5 with normals, 20 000 points, depth 4,
    # real SdfGen: 4 529 nodes, 72 472-byte file, level histogram 1/8/64/512/3944
    o = oracle_mod.sdfgen(fib_sphere(20000), 4)
    assert len(o["structs"]) == 4529 and 8 + 16 * len(o["structs"]) == 72472
    assert np.bincount(levels_of(o["structs"])).tolist() == [1, 8, 64, 512, 3944]
    assert abs(o["scale"] - 1.1) < 1e-3 and all(abs(c - 0.003) < 1e-4 for c in o["offset"])
    # corner values are distances to the sphere in unit-cube units: check the root's
    s = o["scale"]
    for k in range(8):
        corner_model = (np.array([k % 2, 1 - (k // 2 % 2), k // 4 % 2]) - 0.5) * s + np.array(o["offset"])
        assert abs(o["float_values"][0, k] - (np.linalg.norm(corner_model) - 0.5) / s) < 0.02


def test_oracle_structure_and_inheritance(oracle_mod):
    o = oracle_mod.sdfgen(torus_cloud(6000), 5)
    s, fv = o["structs"], o["float_values"]
    assert s[0].tolist()[0] == -1
    internal = np.nonzero(s[:, 1] >= 0)[0]
    for i in internal[:300]:
        c = s[i, 1]
        assert (s[c:c + 8, 0] == i).all()
        for k in range(8):                      # child k inherits corner k (dllmain.cpp:181)
            assert fv[c + k, k] == fv[i, k]
    # the reference throws "Did not find" / "NaN distance" when no point can be the nearest one
    # (the pruning radius always keeps the nearest point, so that takes NaN input); the oracle
    # reports it as an error
    with pytest.raises(RuntimeError):
        oracle_mod.sdfgen(np.full((10, 6), np.nan, dtype=np.float32), 3)


# ---- the HIP builder -----------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("form", ["default", "sibling blocks everywhere"])
@pytest.mark.parametrize("cloud,depth", [("sphere20k", 4), ("sphere20k", 6), ("torus", 5), ("torus", 7), ("two", 5)])
def test_gpu_builder_equals_oracle(sb, oracle_mod, cloud, depth, form, monkeypatch):
    # two kernel forms per level: several 256-thread workgroups per node (levels of fewer than 16 384 nodes, or with long lists) and a
    # wavefront per block of eight siblings, their shared list staged through LDS (the others; the last level's centre and corner
    # searches fused); SDFHIP_GEN_WIDE=8 takes even the smallest levels below the root through the second
    if form != "default":                   # (a measurement knob: the laboratory library reads it, the product does not)
        import sdfbox_amd.lab
        sb = sdfbox_amd.lab.load()
        monkeypatch.setenv("SDFHIP_GEN_WIDE", "8")
    v = {"sphere20k": fib_sphere(20000), "torus": torus_cloud(30000),
         "two": np.concatenate([fib_sphere(4000, 0.3, (-0.2, 0.1, 0.0)), fib_sphere(3000, 0.25, (0.35, -0.1, 0.2))])}[cloud]
    o = oracle_mod.sdfgen(v, depth)
    od, st = sb.OctData.SdfGen(v, depth, want_stats=True)
    assert od.Length == len(o["structs"]) == st.nodes
    assert (od.Structs == o["structs"]).all()
    assert (od.Values == o["values"]).all()
    assert st.global_scale == np.float32(o["scale"]) and tuple(st.global_offset) == tuple(np.float32(c) for c in o["offset"])
    assert od.validate() == (int(levels_of(od.Structs).max()), True)


@pytest.mark.gpu
def test_builder_gives_its_pool_back(sb):
    # the builder keeps its work arrays' device memory for the next build; sdfhip_sdfgen_trim returns it, and the next build is the same
    import torch
    v = fib_sphere(20000)
    a = sb.OctData.SdfGen(v, 6)
    held = torch.cuda.mem_get_info()[0]
    sb.sdfgen_trim()
    assert torch.cuda.mem_get_info()[0] >= held + (64 << 20)
    sb.sdfgen_trim()                                     # (nothing left: still fine)
    b = sb.OctData.SdfGen(v, 6)
    assert (a.Structs == b.Structs).all() and (a.Values == b.Values).all()


@pytest.mark.gpu
def test_gpu_builder_end_to_end(sb, oracle_mod, tmp_path):
    # .ply -> SdfGen on the GPU -> .asdf -> render: the whole of Logic.MakeData + Program.Draw
    v = fib_sphere(20000)
    ply = tmp_path / "sphere.ply"
    write_ply(ply, v)
    od = sb.OctData.SdfGen(sb.OctData.LoadPly(str(ply)), 5)
    od.Save(str(tmp_path / "sphere.asdf"))
    back = sb.OctData.LoadAsdf(str(tmp_path / "sphere.asdf"))
    cam = sb.Logic(96, 96)
    ref, _ = oracle_mod.render(back.Structs, back.Values, cam.State, 96, 96, nthreads=8)
    with sb.Scene(back) as sc:
        img = sc.Draw(cam, 96, 96)
    same = (img.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(img) & np.isnan(ref))
    assert same.all()
    assert (img[..., 0] > 0.0051).sum() > 500                 # the sphere is there and lit
    with pytest.raises(sb.SdfHipError):                      # no usable point: an error code, not a crash
        sb.OctData.SdfGen(np.full((10, 6), np.nan, dtype=np.float32), 3)


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [9, 10])
def test_mesh_scale_import_and_render(sb, oracle_mod, tmp_path, depth):
    # The reference's mesh flow at dragon scale (Program.cs:613-650, Model.MaxDepth = 10): a 1 M-point .ply ->
    # LoadPly -> SdfGen on the GPU -> Save -> LoadAsdf -> 1080p render, checked against the oracle on sampled rows
    # (the oracle's own SdfGen would need minutes for this cloud: the builder's byte parity is held at depths <= 8 above)
    W, H = 1920, 1080
    ply = tmp_path / "knot.ply"
    sb.write_ply(str(ply), sb.knot_point_cloud(1_000_000))
    pts = sb.OctData.LoadPly(str(ply))
    assert pts.shape == (1_000_000, 6)
    od, st = sb.OctData.SdfGen(pts, depth, want_stats=True)
    assert od.validate() == (depth, True) and st.nodes == od.Length > (2_000_000 if depth == 9 else 8_000_000)
    od.Save(str(tmp_path / "knot.asdf"))
    back = sb.OctData.LoadAsdf(str(tmp_path / "knot.asdf"))
    assert (back.Structs == od.Structs).all() and (back.Values == od.Values).all()
    cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
    with sb.Scene(back) as sc:
        assert sc.stack_kernel_ok and sc.depth == depth and sc.top_grid_level > 0
        img, stt = sc.Draw(cam, W, H, sb.FLAG_COUNT, want_stats=True)
        img2 = sc.Draw(cam, W, H, sb.KERNEL_GENERIC)          # the shader's own traversal, and the compaction kernel, beside the default
        img3 = sc.Draw(cam, W, H, sb.FLAG_COMPACT)
    assert ((img.view(np.uint32) == img2.view(np.uint32)) | (np.isnan(img) & np.isnan(img2))).all()
    assert ((img.view(np.uint32) == img3.view(np.uint32)) | (np.isnan(img) & np.isnan(img3))).all()
    assert int(img[..., 3].astype(np.float64).sum()) == stt.n_steps
    hit = img[..., 2] != np.float32(0.2)
    assert 0.05 < hit.mean() < 0.6 and (img[..., 0][hit] > 0).sum() > 10000        # the knot is there and lit
    rows = list(range(3, H, 41))
    ref, cnt = oracle_mod.render(back.Structs, back.Values, cam.State, W, H, row0=3, nrows=len(rows), row_step=41, nthreads=16)
    got = img[rows]
    assert ((got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))).all()


@pytest.mark.gpu
def test_points_to_scene_without_leaving_the_device(sb):
    # sdfhip_sdfgen_scene (the viewer's generate -> upload flow in one call): the handle it returns renders what
    # sdfhip_sdfgen + sdfhip_scene_upload render, and the host copy it can also return is the same tree, byte for byte
    from conftest import assert_frames_identical, make_camera
    v = torus_cloud(60_000)
    for depth in (5, 8):
        od = sb.OctData.SdfGen(v, depth)
        sc, od2, st = sb.Scene.FromPoints(v, depth, want_octdata=True, want_stats=True)
        try:
            assert st.nodes == od.Length == sc.Length and sc.depth == depth
            assert (od2.Structs == od.Structs).all() and (od2.Values == od.Values).all()
            with sb.Scene(od) as ref:
                assert (sc.top_grid_level, sc.top_grid_bytes) == (ref.top_grid_level, ref.top_grid_bytes)
                for cname, (W, H) in (("default", (160, 120)), ("rotated", (200, 96)), ("closeup", (64, 64))):
                    cam = make_camera(cname, W, H)
                    assert_frames_identical(sc.Draw(cam, W, H), ref.Draw(cam, W, H), f"depth {depth} {cname}")
                    assert_frames_identical(sc.Draw(cam, W, H, sb.KERNEL_GENERIC), ref.Draw(cam, W, H), f"depth {depth} {cname}, generic kernel")
        finally:
            sc.close()
        only = sb.Scene.FromPoints(v, depth)                     # without the host copy
        try:
            cam = make_camera("rotated", 96, 96)
            with sb.Scene(od) as ref:
                assert_frames_identical(only.Draw(cam, 96, 96), ref.Draw(cam, 96, 96), "no host copy")
        finally:
            only.close()
    with pytest.raises(sb.SdfHipError):
        sb.Scene.FromPoints(np.full((10, 6), np.nan, np.float32), 3)


@pytest.mark.gpu
@pytest.mark.parametrize("flavour", ["product", "lab"])
def test_builder_scene_agrees_with_the_uploaded_tree_on_edge_clouds(flavour):
    # sdfhip_sdfgen_scene hands its own tree to the scene without validating it again (the product; the laboratory
    # library validates it and refuses a verdict that differs from the builder's word).  On the clouds where a builder could
    # stop early or build a single node, the scene it makes must be the scene an upload of the same tree makes: depth, kernel
    # choice, grid, pixels -- or both ways must fail.
    from conftest import assert_frames_identical, make_camera
    if flavour == "lab":
        import sdfbox_amd.lab
        sb = sdfbox_amd.lab.load()
    else:
        import sdfbox_amd as sb
    one = np.array([[0.1, 0.2, 0.3, 0.0, 0.0, 1.0]], np.float32)
    cases = {"one point": (one, 3), "two coincident points": (np.repeat(one, 2, 0), 4),
             "eight points, depth 10": (fib_sphere(8), 10), "depth 0: the root alone": (fib_sphere(500), 0),
             "depth 1": (fib_sphere(500), 1), "a flat cloud (one plane)": (np.concatenate([fib_sphere(300)[:, :2], np.zeros((300, 1), np.float32),
                                                                           np.tile(np.float32([0, 0, 1]), (300, 1))], 1), 6)}
    for name, (v, depth) in cases.items():
        try:
            od = sb.OctData.SdfGen(v, depth)
        except sb.SdfHipError:
            with pytest.raises(sb.SdfHipError):
                sb.Scene.FromPoints(v, depth)
            continue
        sc = sb.Scene.FromPoints(v, depth)
        try:
            with sb.Scene(od) as ref:
                assert (sc.Length, sc.depth, sc.stack_kernel_ok) == (ref.Length, ref.depth, ref.stack_kernel_ok), name
                assert (sc.top_grid_level, sc.top_grid_bytes) == (ref.top_grid_level, ref.top_grid_bytes), name
                assert od.validate() == (sc.depth, True), name
                cam = make_camera("rotated", 96, 64)
                assert_frames_identical(sc.Draw(cam, 96, 64), ref.Draw(cam, 96, 64), name)
        finally:
            sc.close()
