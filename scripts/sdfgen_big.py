"""Dev script: the GPU SdfGen builder on a larger cloud (1 M points on a torus knot tube), depths 8-10."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import sdfbox_amd as sb

def knot_cloud(n, seed=1):
    rng = np.random.default_rng(seed)
    t = rng.uniform(0, 2 * np.pi, n); a = rng.uniform(0, 2 * np.pi, n)
    p, q, R, r, tube = 2, 3, 0.28, 0.11, 0.035
    c = np.stack([(R + r * np.cos(q * t)) * np.cos(p * t), (R + r * np.cos(q * t)) * np.sin(p * t), r * np.sin(q * t)], 1)
    d = np.stack([-(R + r * np.cos(q * t)) * p * np.sin(p * t) - r * q * np.sin(q * t) * np.cos(p * t),
                  (R + r * np.cos(q * t)) * p * np.cos(p * t) - r * q * np.sin(q * t) * np.sin(p * t),
                  r * q * np.cos(q * t)], 1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    u = np.cross(d, [0, 0, 1.0]); u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = np.cross(d, u)
    nrm = np.cos(a)[:, None] * u + np.sin(a)[:, None] * v
    pos = c + tube * nrm + 0.5
    return np.concatenate([pos, nrm], 1).astype(np.float32)

v = knot_cloud(1_000_000)
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
