"""Dev script (GPU): north_star's "LDS caching of the hot inner nodes per workgroup", measured.  The cursor-stack
kernel with a top grid of level 3 (8^3 cells x 16 B = 8 KB) served from global memory (L1/L2) against the same
grid staged in LDS by every workgroup (64- and 256-thread workgroups), and against no grid / the default grid.
Run with SDFHIP_TOP_GRID_LEVEL=3 (and =0, and unset) in the environment:  python scripts/ab_lds_top.py [depth] [WxH]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import sdfbox_amd as sb
from sdfbox_amd import _lib
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 9
W, H = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1920x1080").split("x"))
od = sb.dragon_standin(depth, nthreads=32); sc = sb.Scene(od)
cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
ref = None
def run(flags, n=200, inflight=2):
    for k in range(20): sc.DrawDevice(cam, W, H, bufs[k % inflight].data_ptr(), flags=flags, stream=streams[k % inflight].cuda_stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n): sc.DrawDevice(cam, W, H, bufs[k % inflight].data_ptr(), flags=flags, stream=streams[k % inflight].cuda_stream)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"depth {depth}, {W}x{H}, top grid level {sc.top_grid_level} ({sc.top_grid_bytes} bytes), SDFHIP_TOP_GRID_LEVEL={os.environ.get('SDFHIP_TOP_GRID_LEVEL')}")
variants = [("default kernels", sb.KERNEL_STACK), ("one kernel, 64-thread workgroups", sb.KERNEL_STACK | sb.TUNE_ONE_KERNEL),
            ("one kernel, 256-thread workgroups", sb.KERNEL_STACK | sb.TUNE_ONE_KERNEL | (3 << 12))]
if 0 < sc.top_grid_level <= 3:
    variants += [("LDS top grid, 64-thread workgroups", sb.KERNEL_STACK | sb.TUNE_ONE_KERNEL | _lib.TUNE_LDS_TOP),
                 ("LDS top grid, 256-thread workgroups", sb.KERNEL_STACK | sb.TUNE_ONE_KERNEL | _lib.TUNE_LDS_TOP | (3 << 12))]
for name, fl in variants:
    ms2, ms1 = run(fl), run(fl, inflight=1)
    img = bufs[0].clone()
    if ref is None: ref = img
    same = bool(torch.equal(img.view(torch.int32), ref.view(torch.int32)))
    print(f"  {name:40s} {ms2:.4f} ms/frame pipelined, {ms1:.4f} single   same image: {same}")
