"""The fixed cost of a frame through sdfhip_multi_render: a 64x64 frame (its march is ~45 us: the rest is the pipeline) and the 4K
bench frame, over device lists that name the one GPU 1..8 times; beside them the floor -- one launch of the same frame on one
stream, waited for.   python scripts/multi_fixed_cost.py [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdfbox_amd as sb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def med(f, n):
    for _ in range(20):
        f()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return float(np.median(t)) * 1e3


for name, od, (W, H), reps in (("64x64 torus_d6", sb.torus_d6(), (64, 64), n), ("3840x2160 dragon_standin_d9", sb.dragon_standin(9), (3840, 2160), max(20, n // 10))):
    cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
    with sb.Scene(od) as sc:
        buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        st = torch.cuda.Stream()

        def one():
            sc.DrawDevice(cam, W, H, buf.data_ptr(), stream=st.cuda_stream); st.synchronize()
        print(f"{name}: one launch on one stream, waited for: {med(one, reps):.4f} ms")
    host = np.empty((H, W, 4), dtype=np.float32)
    for devs in ([0], [0, 0], [0, 0, 0, 0], [0] * 8):
        with sb.MultiScene(od, devs) as ms:
            t_host = med(lambda: ms.Draw(cam, W, H, out=host), reps)

            def dev():
                ms.Submit(0, cam, W, H); ms.Wait(0)
            t_dev = med(dev, reps)
            print(f"{name}: {len(devs)} rank(s): sdfhip_multi_render to a host array {t_host:.4f} ms; submit + wait (frame stays on the device) {t_dev:.4f} ms")
