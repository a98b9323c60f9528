"""Dev script (GPU): SdfGen on the 1 M-point knot, depth 9 and 10: time in the library, node count, a hash of the bytes."""
import sys, hashlib
sys.path.insert(0, ".")
import sdfbox_amd as sb
pts = sb.knot_point_cloud(1000000)
sb.OctData.SdfGen(sb.knot_point_cloud(1000), 3)
for d in (9, 10):
    best = 1e9
    for _ in range(3):
        od, st = sb.OctData.SdfGen(pts, d, want_stats=True); best = min(best, st.total_ms)
    print(f"depth {d}: {best:.1f} ms, {od.Length} nodes, sha {hashlib.sha256(od.Structs.tobytes() + od.Values.tobytes()).hexdigest()[:16]}", flush=True)
