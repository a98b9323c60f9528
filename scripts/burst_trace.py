"""Dev script (GPU, under rocprofv3 --kernel-trace): the kernels of a peer's 20-step burst at 8 ranks, one burst after the other,
so that the trace shows when each launch of a burst starts and ends.  usage: python scripts/burst_trace.py [--order] [rank]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ".")
import torch
import sdfbox_amd as sb
from sdfbox_amd.tiles import BandLayout, SparseShareCall, balanced_owner, band_costs, sparse2_bytes
order = "--order" in sys.argv
rank = int([a for a in sys.argv[1:] if a.isdigit()][0]) if [a for a in sys.argv[1:] if a.isdigit()] else 1
W, H, world, G, nbuf, STEPS = 1920, 1080, 8, 8, 4, 20
od = sb.dragon_standin(9); sc = sb.Scene(od)
cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
whole = torch.zeros((H, W, 4), device="cuda")
sc.DrawDevice(cam, W, H, whole.data_ptr(), stream=torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
costs = band_costs(whole[..., 3], 16)
lay = BandLayout(H, world, 16, owner=balanced_owner(costs, world, extra0=0.04 * sum(costs)))
full = lay.rows_per_rank * W * G
streams = [torch.cuda.Stream() for _ in range(nbuf)]
shares = [torch.zeros(sparse2_bytes(lay.rows_per_rank, W, G, full), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
call = SparseShareCall(sc, W, lay, rank, full, max_frames=G, flags=sb.FLAG_TILE_ORDER if order else 0)
groups = {g: [cam] * g for g in range(1, G + 1)}
def burst(n):
    for s in shares: s[:4].zero_()
    torch.cuda.synchronize(); t0 = time.perf_counter(); k = 0
    while k < n:
        g = min(G, n - k); slot = (k // G) % nbuf
        call(groups[g], shares[slot].data_ptr(), 0, stream=streams[slot].cuda_stream); k += g
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e6
if "--whole" in sys.argv:       # (scripts/share_pmc.sh: whole-frame launches beside the share's, for the counters)
    wf = torch.zeros((H, W, 4), device="cuda")
    for _ in range(8):
        sc.DrawDevice(cam, W, H, wf.data_ptr(), stream=streams[0].cuda_stream)
    torch.cuda.synchronize()
burst(64)
for _ in range(6):
    print(f"burst of {STEPS}: {burst(STEPS):.1f} us", flush=True)
    time.sleep(0.01)
print(f"steady 400: {burst(400) / 400:.2f} us per frame")
