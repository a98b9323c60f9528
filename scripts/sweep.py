"""Dev script: time every kernel variant on the bench workload in ONE process
(interleaved rounds, same device), print per-variant kernel ms and roofline."""
import argparse, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import sdfbox_amd as sb

ap = argparse.ArgumentParser()
ap.add_argument("--depth", type=int, default=9)
ap.add_argument("--sizes", default="1920x1080,3840x2160")
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--camera", default="bench")
a = ap.parse_args()
t0 = time.time(); od = sb.dragon_standin(a.depth); print(f"scene d{a.depth}: N={od.Length} ({od.nbytes/1e6:.1f} MB) built in {time.time()-t0:.1f}s", flush=True)
sc = sb.Scene(od); print("depth", sc.depth, "stack ok", sc.stack_kernel_ok, flush=True)
stream = torch.cuda.current_stream().cuda_stream
variants = [("generic", sb.KERNEL_GENERIC), ("stack", sb.KERNEL_STACK), ("stack o1", sb.KERNEL_STACK | 0x100), ("stack slabs", sb.KERNEL_STACK | 0x200),
            ("stack b128", sb.KERNEL_STACK | 0x2000), ("stack b256", sb.KERNEL_STACK | 0x3000),
            ("stack+compact", sb.KERNEL_STACK | sb.FLAG_COMPACT)]
for size in a.sizes.split(","):
    W, H = (int(v) for v in size.split("x"))
    cam = sb.Logic(W, H)
    if a.camera == 'bench':
        cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
    else:
        cam.Position = (0.5, 0.5, 0.02)      # close-up: the object fills the frame
    buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    st = sb.Stats()
    sc.DrawDevice(cam, W, H, buf.data_ptr(), flags=sb.KERNEL_STACK | sb.FLAG_COUNT, stream=stream, stats=st)
    alg = 8 * st.n_nodes + 8 * st.n_samples + 16 * W * H
    print(f"{size}: nodes {st.n_nodes} samples {st.n_samples} steps {st.n_steps} -> {alg/1e9:.3f} GB algorithmic, "
          f"{st.n_steps/(W*H):.1f} steps/px, {st.n_nodes/max(1,st.n_samples):.2f} nodes/step", flush=True)
    ref = buf.clone()
    times = {n: [] for n, _ in variants}
    for r in range(a.rounds):
        for n, fl in variants:
            sc.DrawDevice(cam, W, H, buf.data_ptr(), flags=fl, stream=stream, stats=st)
            times[n].append(st.kernel_ms)
            if r == 0:
                torch.cuda.synchronize()
                print(f"   {n}: identical to stack image: {torch.equal(buf.view(torch.int32), ref.view(torch.int32))}", flush=True)
    # throughput with 2 frames in flight (two streams, two buffers)
    s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
    bufs = [buf, torch.zeros_like(buf)]
    thr = {}
    for n, fl in variants:
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(40):
                sc.DrawDevice(cam, W, H, bufs[k % 2].data_ptr(), flags=fl, stream=s2[k % 2].cuda_stream)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
        thr[n] = best
    for n, _ in variants:
        print(f"  {n:16s} 2-in-flight {thr[n]:8.3f} ms/frame -> {W*H/thr[n]/1e3:8.1f} Mray/s", flush=True)
    for n, _ in variants:
        t = np.array(times[n][1:])
        ms = np.median(t)
        print(f"  {n:16s} median {ms:8.3f} ms  min {t.min():8.3f}  -> {W*H/ms/1e3:8.1f} Mray/s  {alg/ms/1e6:8.1f} GB/s algorithmic ({alg/ms/1e6/8000*100:.1f}% of 8 TB/s)", flush=True)
sc.close()
