"""Dev script (CPU, oracle): which octree levels do find()'s descents load on the bench frame?
Answers whether an LDS copy of the top levels could take a meaningful share of the node loads."""
import sys
sys.path.insert(0, ".")
import numpy as np
import sdfbox_amd as sb
import oracle
W, H = 1920, 1080
od = sb.dragon_standin(9, nthreads=8)
cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
hist, cnt = oracle.descent_levels(od.Structs, od.Values, cam.State, W, H, nthreads=8, row_step=8)
tot = int(hist.sum())
print(f"{W}x{H} dragon_standin_d9, every 8th row: {int(cnt[0])} node reads, {int(cnt[1])} steps; descents load {tot} records")
acc = 0
for lvl in range(1, 13):
    acc += int(hist[lvl])
    print(f"level {lvl:2d}: {int(hist[lvl]):10d}  {100 * int(hist[lvl]) / tot:5.1f} %   cumulative {100 * acc / tot:5.1f} %")
