"""Dev script (GPU): what a level of the SdfGen builder costs when it has next to nothing to do -- a 2 000-point cloud at depth 10
(eleven levels of launches, scans and host round trips), against the 1 M-point build."""
import sys
sys.path.insert(0, ".")
import sdfbox_amd as sb
sb.OctData.SdfGen(sb.knot_point_cloud(1000), 3)
for n in (2000, 20000):
    pts = sb.knot_point_cloud(n)
    for d in (6, 10):
        best = 1e9
        for _ in range(5):
            od, st = sb.OctData.SdfGen(pts, d, want_stats=True); best = min(best, st.total_ms)
        print(f"{n} points, depth {d}: {best:.2f} ms for {st.levels} levels = {best / st.levels:.3f} ms per level, {od.Length} nodes", flush=True)
