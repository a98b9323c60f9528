"""One-off campaign: the GPU SdfGen builder against the CPU oracle on random point clouds (byte-identical
structs and values expected).  python scripts/sdfgen_fuzz.py [cases [first case]]"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import sdfbox_amd.lab
sb = sdfbox_amd.lab.load()          # SDFHIP_GEN_WIDE is a knob of the laboratory library
import oracle
import os
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for case in range(first, first + n_cases):
    # every third case: the sibling-block kernels on every level below the root (by default only levels of 16 384 nodes and more)
    if case % 3 == 0:
        os.environ["SDFHIP_GEN_WIDE"] = "8"
    else:
        os.environ.pop("SDFHIP_GEN_WIDE", None)
    rng = np.random.default_rng(5000 + case)
    n = int(rng.choice([50, 300, 2000, 20000]))
    kind = int(rng.integers(4))
    if kind == 0:      # ellipsoid shell
        d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        pos = d * rng.uniform(0.1, 0.45, 3) + 0.5; nrm = d / rng.uniform(0.1, 0.45, 3)
    elif kind == 1:    # noisy blob: normals need not be consistent
        pos = rng.normal(0.5, 0.12, (n, 3)); nrm = rng.normal(size=(n, 3))
    elif kind == 2:    # two clusters far apart, tiny and large coordinates
        pos = np.concatenate([rng.normal(-3.0, 0.05, (n // 2, 3)), rng.normal(7.0, 0.3, (n - n // 2, 3))]); nrm = rng.normal(size=(n, 3))
    else:              # points on a plane patch (degenerate extent in z)
        pos = np.concatenate([rng.uniform(0, 1, (n, 2)), np.full((n, 1), 0.25)], 1); nrm = np.tile([0, 0, 1.0], (n, 1))
    nrm = nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-9)
    v = np.concatenate([pos, nrm], 1).astype(np.float32)
    depth = int(rng.integers(1, 9 if n <= 2000 else 8))
    try:
        o = oracle.sdfgen(v, depth)
    except Exception as e:          # the reference throws on some inputs ("Did not find"): the builder must refuse too
        try:
            sb.OctData.SdfGen(v, depth)
            print(f"case {case}: oracle raised {e!r}, GPU builder did not"); bad += 1
        except sb.SdfHipError:
            pass
        continue
    od = sb.OctData.SdfGen(v, depth)
    same = od.Length == len(o["structs"]) and (od.Structs == o["structs"]).all() and (od.Values == o["values"]).all()
    if not same:
        print(f"case {case}: kind {kind} n {n} depth {depth}: DIFFERENT ({od.Length} vs {len(o['structs'])} nodes)"); bad += 1
print(f"{n_cases} clouds (cases {first} .. {first + n_cases - 1}), {bad} differences")
sys.exit(1 if bad else 0)
