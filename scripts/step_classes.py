"""Lane-steps of the bench frame by the kind of cell they sample (sdfhip_debug_step_classes): what the
layout of the grid's cells is tuned by.  GPU: gpurun -- python scripts/step_classes.py [depth] [WxH]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sdfbox_amd as sb

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 9
W, H = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1920x1080").split("x"))
od = sb.dragon_standin(depth, nthreads=32)
scene = sb.Scene(od, device=0)
cam = sb.Logic(W, H)
cam.Position = (0.5, 0.5, -0.35)
cam.Heading = (-0.2, 0.35)
buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
st = sb.Stats()
s = torch.cuda.current_stream().cuda_stream
scene.DrawDevice(cam, W, H, buf.data_ptr(), flags=sb.FLAG_COUNT, stream=s, stats=st)
cls = scene.step_classes(s)
total = st.n_samples
print(f"{W}x{H} dragon_standin_d{depth}: {total} lane-steps, grid level {scene.top_grid_level}, {scene.top_grid_bytes / 1e6:.0f} MB")
for k, v in cls.items():
    print(f"  {k:20s} {v:12d}  {100.0 * v / total:6.2f} %")
