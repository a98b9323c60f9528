"""Dev script: what sdfhip_scene_upload costs for the bench scene (records + top grid), by grid level."""
import os, sys, time
sys.path.insert(0, ".")
import sdfbox_amd as sb
od = sb.dragon_standin(9)
for lv in (0, 6, 8, None):
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); sc = sb.Scene(od, top_grid_level=lv); t = time.perf_counter() - t0
        lvl, nb = sc.top_grid_level, sc.top_grid_bytes
        sc.close(); best = min(best, t)
    print(f"top_grid_level={lv}: grid level {lvl}, {nb / 1e6:.1f} MB, upload {best * 1e3:.1f} ms ({od.nbytes / 1e6:.0f} MB of arrays over PCIe)", flush=True)
