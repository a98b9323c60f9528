"""Dev script: host-side cost of one sdfhip_render_device call (tiny frame, GPU nearly idle)."""
import sys, time
sys.path.insert(0, ".")
import torch
import sdfbox_amd as sb
od = sb.sphere_d4(); sc = sb.Scene(od)
cam = sb.Logic(64, 64)
streams = [torch.cuda.Stream() for _ in range(8)]
bufs = [torch.zeros((64, 64, 4), device="cuda") for _ in range(8)]
for S in (1, 8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(2000):
        sc.DrawDevice(cam, 64, 64, bufs[k % S].data_ptr(), stream=streams[k % S].cuda_stream)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"streams {S}: host issue {(t1-t0)/2000*1e6:.1f} us per call, to completion {(t2-t0)/2000*1e6:.1f} us per call")
