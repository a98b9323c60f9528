"""Dev script: the GPU SdfGen builder on a larger cloud (1 M points on a torus knot tube), depths 8-10."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import sdfbox_amd as sb

v = sb.knot_point_cloud(1_000_000)
sb.OctData.SdfGen(v[:1000], 3)
for depth in (8, 9, 10):
    t0 = time.perf_counter(); od, st = sb.OctData.SdfGen(v, depth, want_stats=True); dt = time.perf_counter() - t0
    od.validate()
    print(f"depth {depth}: {od.Length} nodes ({od.nbytes / 1e6:.0f} MB), {st.candidate_entries / 1e6:.0f} M candidate entries, "
          f"{st.total_ms:.0f} ms in the library, {dt * 1e3:.0f} ms wall", flush=True)
W, H = 1920, 1080
sc = sb.Scene(od)
cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.3)
img, s = sc.Draw(cam, W, H, sb.FLAG_COUNT, want_stats=True)
print(f"render of the depth-10 knot: kernel {s.kernel_ms:.3f} ms (counting build), grid level {sc.top_grid_level}, "
      f"{(img[..., 0] != np.float32(0.005)).mean() * 100:.0f} % of the pixels hit", flush=True)
