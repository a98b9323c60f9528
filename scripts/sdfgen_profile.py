"""Dev script (GPU, under rocprofv3): ONE build of the 1 M-point knot at the depth given (default 9) after a warm-up build of a
small cloud -- the kernels' per-build totals are then what the profiler's sums hold (scripts/profile_sdfgen.sh)."""
import sys
sys.path.insert(0, ".")
import sdfbox_amd as sb
d = int(sys.argv[1]) if len(sys.argv) > 1 else 9
pts = sb.knot_point_cloud(1000000)
od, st = sb.OctData.SdfGen(pts, d, want_stats=True)
print(f"depth {d}: {od.Length} nodes, {st.candidate_entries} candidate entries, {st.total_ms:.1f} ms in the library")
