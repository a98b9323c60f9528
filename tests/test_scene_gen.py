"""The analytic builder against SdfGen's construction rules
(SdfGen/dllmain.cpp:163-207) and the facts SURVEY.md recorded from the real
SdfGen."""
import hashlib

import numpy as np
import pytest


def levels_of(structs):
    lvl = np.full(len(structs), -1, dtype=np.int32)
    lvl[0] = 0
    for i in range(len(structs)):                      # parents precede children in append order
        c = structs[i, 1]
        if c >= 0:
            lvl[c:c + 8] = lvl[i] + 1
    return lvl


def test_sdfgen_sphere_facts_from_the_survey(sb):
    # SdfGen on a sphere point cloud r=0.5: bbox -> GlobalScale 1.1, GlobalOffset 0.003,
    # y flipped (dllmain.cpp:67-88).  In unit-cube coordinates that is this sphere; the
    # real SdfGen gave 4 529 nodes / 72 472 B, level histogram 1/8/64/512/3944.
    c, r = 0.003 / 1.1, 0.5 / 1.1
    od = sb.OctData.Generate(sb._lib.SHAPE_SPHERE, [0.5 - c, 0.5 + c, 0.5 - c, r], 4, nthreads=1)
    assert od.Length == 4529 and 8 + od.nbytes == 72472
    assert np.bincount(levels_of(od.Structs)).tolist() == [1, 8, 64, 512, 3944]


def test_structure_children_contiguous_dfs_order(sb):
    od = sb.torus_d6()
    s = od.Structs
    assert s[0].tolist()[0] == -1
    internal = np.nonzero(s[:, 1] >= 0)[0]
    # every children block is 8 wide, starts at 1 + 8k, and points back at its parent
    assert ((s[internal, 1] - 1) % 8 == 0).all()
    for i in internal[:200]:
        assert (s[s[i, 1]:s[i, 1] + 8, 0] == i).all()
    # blocks are appended in DFS pre-order (construct recurses into child k right
    # after appending the block): the descendants of child k follow those of child
    # k-1 without a gap, the first starting right behind the block itself.
    def subtree_end(c):            # c = first index of a children block; returns next free index
        cur = c + 8
        for k in range(8):
            cc = s[c + k, 1]
            if cc >= 0:
                assert cc == cur, (c, k, cc, cur)
                cur = subtree_end(cc)
        return cur
    assert subtree_end(1) == od.Length
    assert od.validate() == (6, True)


def test_split_rule_and_quantiser_against_numpy(sb):
    # sphere |p - c| - r: recompute construct()'s decision and FromFloat for every node
    cx, cy, cz, r = 0.5, 0.5, 0.5, 0.3
    od = sb.sphere_d4()
    s, v = od.Structs, od.Values
    lvl = levels_of(s)
    pos = np.zeros((od.Length, 3), dtype=np.float32)
    for i in range(od.Length):
        c = s[i, 1]
        if c >= 0:
            half = np.float32(0.5) ** np.float32(lvl[i]) / np.float32(2)
            for k in range(8):
                pos[c + k] = pos[i] + np.array([k % 2, k // 2 % 2, k // 4 % 2], dtype=np.float32) * half
    f32 = np.float32
    for i in range(od.Length):
        scale = f32(0.5) ** f32(lvl[i])
        ctr = pos[i].astype(np.float64) + float(scale) / 2
        center_value = f32(abs(np.sqrt(((ctr - [cx, cy, cz]) ** 2).sum()) - r))
        assert (s[i, 1] >= 0) == bool(center_value < scale * 2 and lvl[i] < 4)
        for k in range(8):
            p = pos[i].astype(np.float64) + np.array([k % 2, k // 2 % 2, k // 4 % 2]) * float(scale)
            f = f32(np.sqrt(((p - [cx, cy, cz]) ** 2).sum()) - r)
            normd = f32(f32(f / f32(2)) / scale)
            byte = int(np.floor(min(max(f32(normd + f32(0.25)), f32(0)), f32(1)) * f32(255)))
            assert v[i, k] == byte, (i, k)


def test_output_does_not_depend_on_thread_count(sb):
    p = [0.5, 0.5, 0.5, 0.42, 12.0 * np.pi, 0.004]
    a = sb.OctData.Generate(sb._lib.SHAPE_GYROID, p, 6, nthreads=1)
    b = sb.OctData.Generate(sb._lib.SHAPE_GYROID, p, 6, nthreads=5)
    assert a.Length == b.Length
    assert (a.Structs == b.Structs).all() and (a.Values == b.Values).all()


def test_dragon_standin_is_deterministic(sb):
    # pinned digest: the same bytes must come out on every host (own sin/cos, no libm)
    od = sb.dragon_standin(6)
    h = hashlib.sha256(od.Structs.tobytes() + od.Values.tobytes()).hexdigest()
    assert od.Length == 148745, od.Length
    assert h == DRAGON_D6_SHA256, h


DRAGON_D6_SHA256 = "98ede0b613992d7679afa8780ab79c8517c1c24d9be7b4caf734b9dd26d208f1"
