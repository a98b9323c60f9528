"""Dev script (GPU): whole 1080p frames of the bench scene G to a launch (sdfhip_render_batch_device: grid.y = frame) with B launches in
flight, against one frame per launch on four streams -- does one dispatch of several frames overlap their tails better than the queues do?"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdfbox_amd as sb
W, H = 1920, 1080
od = sb.dragon_standin(9, nthreads=16)
sc = sb.Scene(od)
cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
for G, B in ((1, 4), (2, 2), (2, 3), (4, 1), (4, 2), (4, 3), (8, 1), (8, 2)):
    streams = [torch.cuda.Stream() for _ in range(B)]
    bufs = [torch.zeros((G, H, W, 4), dtype=torch.float32, device="cuda") for _ in range(B)]
    def go(k):
        s = streams[k % B]
        if G == 1: sc.DrawDevice(cam, W, H, bufs[k % B].data_ptr(), stream=s.cuda_stream)
        else: sc.DrawBatchDevice([cam.State] * G, W, H, bufs[k % B].data_ptr(), stream=s.cuda_stream)
    for n in (20, 400):
        launches = max(1, n // G)
        for k in range(max(2 * B, launches // 8)): go(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(launches): go(k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (launches * G)
        print(f"{G} frame(s) per launch, {B} launch(es) in flight, {launches * G} frames: {dt * 1e3:.4f} ms per frame", flush=True)
