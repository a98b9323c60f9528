"""How far is the arithmetic contract from the other legal readings of the shader?

The oracle (and the kernels) fuse the multiply-adds HLSL compiles to `mad` and sample the value texture
with full fp32 weights.  Nothing in the reference pins that reading: D3D11 leaves mad's fusing
to the implementation, and its texture units filter with as few as 8 fractional weight bits.  The variants of
oracle/sdf_oracle.c restate those other readings; this test evaluates SURVEY.md 8d's parity statement --
"RGB abs-err <= 1e-5 and alpha (step count) equal on >= 99.9 % of pixels" -- between the contract and each of
them, on cfg-1 (256x256 sphere_d4, default camera) and on the 1080p bench frame (cfg-2), and holds the measured
numbers (DESIGN.md section 2 quotes them).  Parity with the reference stays "unpinned": these are distances
between restatements, not to the HLSL on hardware."""
import numpy as np
import pytest


def parity(a, b):
    """SURVEY 8d's statement + looser views of the same difference."""
    rgb = np.abs(a[..., :3] - b[..., :3]).max(-1)
    rgb = np.where(np.isnan(a[..., :3]).any(-1) & np.isnan(b[..., :3]).any(-1), 0.0, rgb)
    same_alpha = a[..., 3] == b[..., 3]
    return {
        "statement": float(((rgb <= 1e-5) & same_alpha).mean()),          # RGB <= 1e-5 and equal step count
        "rgb_1e-5": float((rgb <= 1e-5).mean()),
        "rgb_1/255": float((rgb <= 1.0 / 255).mean()),                    # invisible after the display pass's 8-bit output
        "alpha_equal": float(same_alpha.mean()),
        "alpha_within_2": float((np.abs(a[..., 3] - b[..., 3]) <= 2).mean()),
        "bit_identical": float((a.view(np.uint32) == b.view(np.uint32)).all(-1).mean()),
    }


@pytest.fixture(scope="module")
def frames(sb, oracle_mod):
    out = {}
    od = sb.sphere_d4()
    cam = sb.Logic(256, 256)
    out["cfg1"] = {v: oracle_mod.render(od.Structs, od.Values, cam.State, 256, 256, nthreads=8, native=v)[0]
                   for v in (False, "unfused", "lerp_mathcs", "sampler8", "rsqrt1ulp", "sampler8_unfused")}
    od = sb.dragon_standin(9, nthreads=8)
    cam = sb.Logic(1920, 1080); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
    out["bench"] = {v: oracle_mod.render(od.Structs, od.Values, cam.State, 1920, 1080, nthreads=8, native=v)[0]
                    for v in (False, "unfused", "lerp_mathcs", "sampler8", "rsqrt1ulp", "sampler8_unfused")}
    return out


def test_fused_vs_unfused_contract(frames):
    # SURVEY 8c's literal "lerp = a + t(b-a), FMA contraction off" against the fused contract the kernels implement
    c1, b = parity(frames["cfg1"][False], frames["cfg1"]["unfused"]), parity(frames["bench"][False], frames["bench"]["unfused"])
    print("fused vs unfused:", c1, b)
    # measured: cfg-1 1.0000 (45 % of the pixels bit-identical); bench frame 0.99825 -- 0.17 % of the pixels (lit
    # surface) differ by more than 1e-5 in grey level, 0.002 % in step count; all but 5 pixels within one 8-bit level
    assert c1["statement"] == 1.0
    assert 0.997 <= b["statement"] < 0.999              # just short of 8d's ">= 99.9 %": say so, do not hide it
    assert b["alpha_equal"] >= 0.9999 and b["rgb_1/255"] >= 0.99999


def test_fused_vs_the_references_own_cpu_lerp(frames):
    # Math.cs:25-28 (a*(1-p) + b*p; zero callers in the reference)
    c1, b = parity(frames["cfg1"][False], frames["cfg1"]["lerp_mathcs"]), parity(frames["bench"][False], frames["bench"]["lerp_mathcs"])
    print("fused vs Math.cs lerp:", c1, b)
    # measured: cfg-1 0.99998 (4 pixels of 65 536 change their step count), bench frame 0.99825
    assert c1["statement"] >= 0.9999 and 0.997 <= b["statement"] < 0.999
    assert b["alpha_equal"] >= 0.9999 and b["rgb_1/255"] >= 0.99999


def test_fp32_weights_vs_8_bit_sampler_weights(frames):
    c1, b = parity(frames["cfg1"][False], frames["cfg1"]["sampler8"]), parity(frames["bench"][False], frames["bench"]["sampler8"])
    print("fp32 vs 8-bit sampler weights:", c1, b)
    # This is the large one.  Measured: cfg-1 0.428 -- 8 % of the pixels finish on a different step and half of the
    # lit sphere's grey levels move by more than 1e-5 (99.9 % stay within one 8-bit display level); bench frame
    # 0.9550 (the sky, 80 % of it, is untouched; 0.16 % of the pixels change their step count).  The 8d statement
    # does NOT hold between the fp32 restatement and a sampler of D3D11's minimum precision: the fp32 restatement
    # is the specification, not a prediction of any particular GPU's texture unit.
    assert 0.35 <= c1["statement"] <= 0.55 and 0.94 <= b["statement"] <= 0.97
    assert c1["rgb_1/255"] >= 0.998 and b["rgb_1/255"] >= 0.9995
    assert c1["alpha_within_2"] >= 0.99 and b["alpha_equal"] >= 0.998


def test_exact_vs_one_ulp_reciprocal_square_roots(frames):
    # normalize(v) = v * rsqrt(dot(v, v)) in HLSL; the contract takes the correctly rounded 1 / sqrtf.  Every rsqrt one ulp off
    # (up or down by the lowest bit of its argument): ray directions, light directions, normals
    c1, b = parity(frames["cfg1"][False], frames["cfg1"]["rsqrt1ulp"]), parity(frames["bench"][False], frames["bench"]["rsqrt1ulp"])
    print("exact vs 1-ulp rsqrt:", c1, b)
    # measured: cfg-1 1.0000 (44 % of the pixels bit-identical); bench frame 0.99698 -- 0.30 % of the pixels differ by more than
    # 1e-5 in grey level, 0.003 % in step count; all but 8 pixels within one 8-bit level
    assert c1["statement"] == 1.0
    assert 0.995 <= b["statement"] < 0.999
    assert b["alpha_equal"] >= 0.9999 and b["rgb_1/255"] >= 0.99999


def test_fp32_fused_vs_8_bit_weights_and_unfused_mads_together(frames):
    # the reading a real D3D11 GPU may run: the texture unit's 8-bit bilinear weights AND separately rounded mads, both at once
    c1, b = parity(frames["cfg1"][False], frames["cfg1"]["sampler8_unfused"]), parity(frames["bench"][False], frames["bench"]["sampler8_unfused"])
    s1, sb_ = parity(frames["cfg1"]["sampler8"], frames["cfg1"]["sampler8_unfused"]), parity(frames["bench"]["sampler8"], frames["bench"]["sampler8_unfused"])
    print("fp32 fused vs 8-bit weights + unfused:", c1, b)
    print("8-bit weights: fused vs unfused:", s1, sb_)
    # Measured: against the contract the combination sits where the 8-bit sampler alone sits -- cfg-1 0.4282 (alone: 0.4283),
    # bench frame 0.95497 (alone: 0.95500) -- the sampler dominates; and between the two 8-bit readings, fusing or not moves
    # 0.05 % of the pixels by the statement (cfg-1 0.99950, bench frame 0.99944; 46 % / 96 % stay bit-identical).
    assert 0.35 <= c1["statement"] <= 0.55 and 0.94 <= b["statement"] <= 0.97
    assert c1["rgb_1/255"] >= 0.998 and b["rgb_1/255"] >= 0.9995 and b["alpha_equal"] >= 0.998
    assert s1["statement"] >= 0.99 and 0.99 <= sb_["statement"] <= 1.0
