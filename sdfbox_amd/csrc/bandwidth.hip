// libsdfhip.so, device half: the box's own HBM streaming rate -- the denominator SURVEY.md 8d asks the roofline to be
// quoted against ("measured device bandwidth on the box: device-to-device copy / triad microbench of >= 1 GB ... next to the
// 8 TB/s nameplate").  Not part of the frame's path: bench.py and a host that wants to know what its device delivers call it.
//
// Three streaming kernels over arrays far larger than the 256 MiB Infinity Cache, 16 bytes per lane and access:
//   copy   dst[i] = src[i]                   reads N, writes N bytes     one float4 per thread, no loop
//   triad  a[i]   = b[i] + s * c[i]          reads 2 N, writes N bytes   (STREAM's triad on float4), one float4 per thread
//   read   sum over src                      reads N bytes               grid-stride, 8 non-temporal loads in flight per lane
// Which shape reaches the chip's rate was measured (scripts/micro/bw_variants.hip, profiles/r05_bw_variants.txt): grid-stride
// copies with 1 - 8 accesses in flight per lane stay at 4.6 - 5.7 TB/s whatever the grid, hipMemcpyAsync at 5.0, the copy with
// one float4 per thread and no loop reaches 6.25 (the guide's 6.29), the read-only stream 6.3.  The rate is the bytes the
// kernel moves / its HIP-event time.
#include "scene.h"
#include "abi_guard.h"

using namespace sdfhip;

namespace {

constexpr int BW_THREADS = 256;
constexpr int BW_READ_UNROLL = 8;
typedef float f4 __attribute__((ext_vector_type(4)));       // (the nontemporal builtins take native vectors, not HIP's float4 class)

__global__ __launch_bounds__(BW_THREADS) void k_bw_copy(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * BW_THREADS + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

__global__ __launch_bounds__(BW_THREADS) void k_bw_triad(f4 *__restrict__ a, const f4 *__restrict__ b, const f4 *__restrict__ c, float s, size_t n)
{
    const size_t i = (size_t)blockIdx.x * BW_THREADS + threadIdx.x;
    if (i < n) a[i] = b[i] + s * c[i];
}

__global__ __launch_bounds__(BW_THREADS) void k_bw_read(const f4 *__restrict__ src, float *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * BW_THREADS;
    size_t i = (size_t)blockIdx.x * BW_THREADS + threadIdx.x;
    f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (; i + (BW_READ_UNROLL - 1) * stride < n; i += BW_READ_UNROLL * stride) {
        f4 v[BW_READ_UNROLL];
#pragma unroll
        for (int u = 0; u < BW_READ_UNROLL; u++) v[u] = __builtin_nontemporal_load(&src[i + u * stride]);
#pragma unroll
        for (int u = 0; u < BW_READ_UNROLL; u++) acc += v[u];
    }
    for (; i < n; i += stride) acc += src[i];
    const float t = acc.x + acc.y + acc.z + acc.w;
    if (t == 123456.789f) out[0] = t;                        // (never true for the fill below: keeps the loads)
}

__global__ __launch_bounds__(BW_THREADS) void k_bw_fill(f4 *p, size_t n, float v)
{
    const size_t stride = (size_t)gridDim.x * BW_THREADS;
    for (size_t i = (size_t)blockIdx.x * BW_THREADS + threadIdx.x; i < n; i += stride) p[i] = f4{v, v + 1.0f, v + 2.0f, v + 3.0f};
}

}  // namespace

extern "C" int sdfhip_device_bandwidth(int device, uint64_t bytes, uint32_t reps, double *copy_gbs, double *triad_gbs, double *read_gbs)
try {
    clear_error();
    if (!copy_gbs && !triad_gbs && !read_gbs) return fail(SDFHIP_ERR_ARG, "device_bandwidth: nothing asked for");
    if (bytes < (1ull << 20) || bytes > (64ull << 30) || reps == 0 || reps > 1000)
        return fail(SDFHIP_ERR_ARG, "device_bandwidth: arrays of 1 MiB .. 64 GiB, 1 .. 1000 repetitions");
    if (copy_gbs) *copy_gbs = 0.0;
    if (triad_gbs) *triad_gbs = 0.0;
    if (read_gbs) *read_gbs = 0.0;
    DeviceGuard g(device);
    if (!g.ok) {
        (void)hipGetLastError();     // (the runtime keeps a failed call's error for the next hipGetLastError(): a later launch check would report it as its own)
        return fail(SDFHIP_ERR_DEVICE, "device_bandwidth: no device %d", device);
    }
    const size_t n = (size_t)(bytes / sizeof(f4));
    f4 *buf[3] = { nullptr, nullptr, nullptr };
    float *sink = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = SDFHIP_OK;
    auto cleanup = [&]() {
        for (f4 *p : buf) if (p) (void)hipFree(p);
        if (sink) (void)hipFree(sink);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (st) (void)hipStreamDestroy(st);
    };
    auto tryhip = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == SDFHIP_OK) rc = fail(e == hipErrorOutOfMemory ? SDFHIP_ERR_NOMEM : SDFHIP_ERR_DEVICE, "device_bandwidth: %s: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    const int narr = triad_gbs ? 3 : copy_gbs ? 2 : 1;
    for (int i = 0; i < narr && rc == SDFHIP_OK; i++) tryhip(hipMalloc((void **)&buf[i], n * sizeof(f4)), "hipMalloc");
    if (rc == SDFHIP_OK) tryhip(hipMalloc((void **)&sink, 64), "hipMalloc");
    if (rc == SDFHIP_OK) tryhip(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
    if (rc == SDFHIP_OK) tryhip(hipEventCreate(&e0), "hipEventCreate");
    if (rc == SDFHIP_OK) tryhip(hipEventCreate(&e1), "hipEventCreate");
    if (rc != SDFHIP_OK) { cleanup(); return rc; }
    for (int i = 0; i < narr; i++) hipLaunchKernelGGL(k_bw_fill, dim3(4096), dim3(BW_THREADS), 0, st, buf[i], n, (float)i);
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    const dim3 flat((uint32_t)((n + BW_THREADS - 1) / BW_THREADS)), resident((uint32_t)cus * 32u);     // (the read: 32 workgroups per CU)
    double *outs[3] = { copy_gbs, triad_gbs, read_gbs };
    const double moved[3] = { 2.0, 3.0, 1.0 };
    for (int which = 0; which < 3 && rc == SDFHIP_OK; which++) {
        if (!outs[which]) continue;
        auto launch = [&]() {
            if (which == 0)      hipLaunchKernelGGL(k_bw_copy, flat, dim3(BW_THREADS), 0, st, buf[1], buf[0], n);
            else if (which == 1) hipLaunchKernelGGL(k_bw_triad, flat, dim3(BW_THREADS), 0, st, buf[0], buf[1], buf[2], 0.5f, n);
            else                 hipLaunchKernelGGL(k_bw_read, resident, dim3(BW_THREADS), 0, st, buf[0], sink, n);
        };
        double best = 0.0;
        for (int trial = 0; trial < 2; trial++) {            // (twice: the first pass also brings the clocks up)
            launch();
            if (!tryhip(hipEventRecord(e0, st), "hipEventRecord")) break;
            for (uint32_t r = 0; r < reps; r++) launch();
            if (!tryhip(hipEventRecord(e1, st), "hipEventRecord")) break;
            if (!tryhip(hipEventSynchronize(e1), "hipEventSynchronize")) break;
            float ms = 0.0f;
            if (!tryhip(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime")) break;
            if (ms > 0.0f) { const double gbs = moved[which] * (double)n * sizeof(f4) * reps / (ms * 1e-3) / 1e9; if (gbs > best) best = gbs; }
        }
        *outs[which] = best;
    }
    if (rc == SDFHIP_OK) tryhip(hipStreamSynchronize(st), "hipStreamSynchronize");
    cleanup();
    return rc;
}
SDFHIP_ABI_CATCH(sdfhip_device_bandwidth)
