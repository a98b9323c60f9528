"""Dev script: how fast do 451 MB of pageable host memory reach the device?  hipMemcpy as it is, after hipHostRegister of the
source (time of the registration included), and in chunks through a pinned staging ring filled by several threads."""
import ctypes, time, sys, threading
import numpy as np, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
n = 451 * 1000 * 1000
src = np.random.default_rng(0).integers(0, 255, n, dtype=np.uint8)
dst = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
a = t(lambda: hip.hipMemcpy(dst.data_ptr(), src.ctypes.data, n, 1))
print(f"hipMemcpy from pageable memory: {a*1e3:.1f} ms, {n/a/1e9:.1f} GB/s")
def reg():
    assert hip.hipHostRegister(src.ctypes.data, n, 0) == 0
    hip.hipMemcpy(dst.data_ptr(), src.ctypes.data, n, 1)
    hip.hipHostUnregister(src.ctypes.data)
b = t(reg)
print(f"hipHostRegister + hipMemcpy + hipHostUnregister: {b*1e3:.1f} ms, {n/b/1e9:.1f} GB/s")
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
c = t(lambda: dst.copy_(pin, non_blocking=True))
print(f"from pinned memory: {c*1e3:.1f} ms, {n/c/1e9:.1f} GB/s")
pn = pin.numpy()
d = t(lambda: np.copyto(pn, src))
print(f"one-thread memcpy pageable -> pinned: {d*1e3:.1f} ms, {n/d/1e9:.1f} GB/s")
def par(k):
    chunk = (n + k - 1) // k
    def work(i):
        lo, hi = i * chunk, min(n, (i + 1) * chunk)
        np.copyto(pn[lo:hi], src[lo:hi]); 
    def go():
        th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
        [x.start() for x in th]; [x.join() for x in th]
        dst.copy_(pin, non_blocking=True)
    return go
for k in (4, 8, 16):
    e = t(par(k))
    print(f"{k}-thread memcpy into pinned, then one DMA: {e*1e3:.1f} ms, {n/e/1e9:.1f} GB/s")
