"""The viewer's generate -> upload flow (Program.cs:613-650 + :147-152) on the 1 M-point knot: sdfhip_sdfgen (tree to the
host) + sdfhip_scene_upload (and back), against sdfhip_sdfgen_scene (the tree never leaves HBM).  Wall time around the calls
and time inside the library, best of 4 after a warm-up call of each flow (a build that follows the release of gigabytes of
device memory pays for the driver's clean-up: the two flows are timed in separate loops).
GPU: gpurun -- python scripts/points_to_scene.py"""
import sys
import time

sys.path.insert(0, ".")
import sdfbox_amd as sb

pts = sb.knot_point_cloud(1000000)
sb.OctData.SdfGen(sb.knot_point_cloud(1000), 3)
for d in (9, 10):
    gen = up = one = lib1 = 1e9
    for rep in range(5):
        t = time.perf_counter()
        od, st = sb.OctData.SdfGen(pts, d, want_stats=True)
        t1 = time.perf_counter()
        sc = sb.Scene(od)
        t2 = time.perf_counter()
        lvl, gb = sc.top_grid_level, sc.top_grid_bytes
        sc.close()
        if rep:
            gen = min(gen, t1 - t); up = min(up, t2 - t1)
    for rep in range(5):
        t = time.perf_counter()
        sc, st = sb.Scene.FromPoints(pts, d, want_stats=True)
        dt = time.perf_counter() - t
        sc.close()
        if rep:
            one = min(one, dt); lib1 = min(lib1, st.total_ms)
    print(f"depth {d}: {od.Length} nodes; sdfgen {gen * 1e3:.1f} ms + upload {up * 1e3:.1f} ms = {(gen + up) * 1e3:.1f} ms;  "
          f"sdfgen_scene {one * 1e3:.1f} ms ({lib1:.1f} ms inside the library)  (grid level {lvl}, {gb / 1e6:.0f} MB)", flush=True)
