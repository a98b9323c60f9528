import os, sys, time
sys.path.insert(0, ".")
import sdfbox_amd.lab
sb = sdfbox_amd.lab.load()
pts = sb.knot_point_cloud(1000000)
sb.Scene.FromPoints(pts[:2000], 3).close()
for rep in range(3):
    t = time.perf_counter(); sc, st = sb.Scene.FromPoints(pts, 10, want_stats=True); dt = time.perf_counter() - t; sc.close()
    print(f"scene from points depth 10: {dt*1e3:.2f} ms wall, {st.total_ms:.2f} ms in the library", flush=True)
os.environ["SDFHIP_GEN_LEVELS"] = "1"
sc, st = sb.Scene.FromPoints(pts, 10, want_stats=True); sc.close()
os.environ.pop("SDFHIP_GEN_LEVELS")
os.environ["SDFHIP_GEN_TIMING"] = "1"
sc, st = sb.Scene.FromPoints(pts, 10, want_stats=True); sc.close()
