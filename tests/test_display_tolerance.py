"""The one closeness statement that survives every legal reading of the shader: the DISPLAYED frame.

Parity of the kernels is bit-exact against oracle/sdf_oracle.c, which fixes ONE reading of Compute.hlsl (fused mads, fp32
bilinear weights, correctly rounded 1/sqrt); nothing in the reference pins that reading.  The oracle's contract
variants restate the other legal readings -- unfused mads, the reference's own CPU lerp (Math.cs:25-28), D3D11's 8-bit bilinear
weights (Compute.hlsl:15-29 through the texture unit), reciprocal square roots one ulp off, and 8-bit weights with unfused mads
together.  SURVEY.md 8d's float statement (RGB <= 1e-5 and equal step count on >= 99.9 % of the pixels) does not hold against
all of them (tests/test_oracle_variants.py: 42.8 % against the 8-bit sampler on cfg-1).  What the user of SdfBox sees is the
display pass's RGBA8 frame (DisplayFrag.hlsl:16-24: pow(v, 1/2.2) into an R8G8B8A8_UNorm swap chain), and THERE the GPU's
frame is within one 8-bit level per channel of every reading on >= 99.9 % of the pixels -- 99.85 % against the 8-bit sampler on
cfg-1's small sphere, where 96 silhouette pixels of 65 536 flip between hit and miss.  This test holds those numbers on the GPU:
sdfhip_render_display against oracle_display of every variant, on cfg-1 (256x256 sphere_d4, default camera) and on a 320x180
frame of the bench scene's shape (gyroid shell, depth 7) under the bench camera; the pixels beyond one level are listed by count."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VARIANTS = ("unfused", "lerp_mathcs", "rsqrt1ulp", "sampler8", "sampler8_unfused")
# the fraction of pixels within one 8-bit level per channel that the test holds, (cfg-1, bench-shape frame); measured on the
# contract's own display: 1.0 / 1.0 for the three arithmetic readings, 0.99854 / 0.99950 for the two with the 8-bit sampler
FLOOR = {"unfused": (0.999, 0.999), "lerp_mathcs": (0.999, 0.999), "rsqrt1ulp": (0.999, 0.999),
         "sampler8": (0.998, 0.999), "sampler8_unfused": (0.998, 0.999)}


def test_displayed_frame_is_within_one_level_of_every_reading_of_the_shader(sb, oracle_mod):
    cases = []
    od = sb.sphere_d4()
    cases.append(("cfg-1 256x256 sphere_d4", od, sb.Logic(256, 256), 256, 256, 0))
    od = sb.dragon_standin(7, nthreads=8)
    cam = sb.Logic(320, 180); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
    cases.append(("320x180 bench-shape frame (gyroid shell, depth 7)", od, cam, 320, 180, 1))
    report = []
    for name, od, cam, W, H, col in cases:
        with sb.Scene(od) as scene:
            got = scene.DrawDisplay(cam, W, H).astype(np.int32)                      # sdfhip_render_display: RGBA8, R,G,B,A
        contract, _ = oracle_mod.render(od.Structs, od.Values, cam.State, W, H, nthreads=8)
        d0 = np.abs(got - oracle_mod.display(contract).astype(np.int32)).max(-1)
        assert d0.max() <= 1, f"{name}: the GPU's display frame is more than one level from the contract's ({int((d0 > 1).sum())} pixels)"
        for v in VARIANTS:
            other, _ = oracle_mod.render(od.Structs, od.Values, cam.State, W, H, nthreads=8, native=v)
            d = np.abs(got - oracle_mod.display(other).astype(np.int32)).max(-1)
            frac, beyond = float((d <= 1).mean()), int((d > 1).sum())
            report.append(f"{name} vs {v}: {frac:.5f} within one level, {beyond} of {d.size} pixels beyond (largest difference {int(d.max())} levels), "
                          f"{float((d == 0).mean()):.5f} identical")
            assert frac >= FLOOR[v][col], report[-1]
    print("\n".join(report))


def test_device_bandwidth_is_a_streaming_rate(sb):
    # sdfhip_device_bandwidth: the roofline's measured denominator -- copy, triad and read-only rates of the library's
    # own kernels.  On an MI355X each is a few TB/s (the guide: 6.3 achievable of 8 nameplate); bad arguments are refused with a message.
    import ctypes
    c, t, r = sb.device_bandwidth(0, 512 << 20, 5)
    for name, v in (("copy", c), ("triad", t), ("read", r)):
        assert 2000.0 < v < 8000.0, (name, v)
    L = sb._lib
    x = ctypes.c_double()
    assert L.lib.sdfhip_device_bandwidth(0, 512 << 20, 3, None, None, ctypes.byref(x)) == L.OK and 2000.0 < x.value < 8000.0   # read only: one array
    assert L.lib.sdfhip_device_bandwidth(0, 1 << 10, 3, ctypes.byref(x), None, None) == L.ERR_ARG       # arrays below 1 MiB
    assert L.lib.sdfhip_device_bandwidth(0, 512 << 20, 3, None, None, None) == L.ERR_ARG                # nothing asked for
    assert L.lib.sdfhip_device_bandwidth(99, 512 << 20, 3, ctypes.byref(x), None, None) == L.ERR_DEVICE and b"device" in L.lib.sdfhip_last_error()
