// libsdfhip.so, device half: the box's own HBM streaming rate -- the denominator SURVEY.md 8d asks the roofline to be
// quoted against ("measured device bandwidth on the box: device-to-device copy / triad microbench of >= 1 GB ... next to the
// 8 TB/s nameplate").  Not part of the frame's path: bench.py and a host that wants to know what its device delivers call it.
//
// Two streaming kernels over arrays far larger than the 256 MiB Infinity Cache, 16 bytes per lane and access, grid-stride with
// four accesses in flight per lane:
//   copy   dst[i] = src[i]                   reads N, writes N bytes
//   triad  a[i]   = b[i] + s * c[i]          reads 2 N, writes N bytes   (STREAM's triad on float4)
// The rate is the bytes the kernel moves / its HIP-event time, best of four grid sizes (4 / 8 / 16 / 32 workgroups per CU) with
// plain and with non-temporal loads and stores: which is fastest differs by a few per cent between boxes.
#include "scene.h"

using namespace sdfhip;

namespace {

constexpr int BW_THREADS = 256;
constexpr int BW_UNROLL = 4;
typedef float f4 __attribute__((ext_vector_type(4)));       // (the nontemporal builtins take native vectors, not HIP's float4 class)

template <bool NT>
__global__ __launch_bounds__(BW_THREADS) void k_bw_copy(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * BW_THREADS;
    size_t i = (size_t)blockIdx.x * BW_THREADS + threadIdx.x;
    for (; i + (BW_UNROLL - 1) * stride < n; i += BW_UNROLL * stride) {
        f4 v[BW_UNROLL];
#pragma unroll
        for (int u = 0; u < BW_UNROLL; u++) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < BW_UNROLL; u++) { if (NT) __builtin_nontemporal_store(v[u], &dst[i + u * stride]); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

template <bool NT>
__global__ __launch_bounds__(BW_THREADS) void k_bw_triad(f4 *__restrict__ a, const f4 *__restrict__ b, const f4 *__restrict__ c,
                                                          float s, size_t n)
{
    const size_t stride = (size_t)gridDim.x * BW_THREADS;
    size_t i = (size_t)blockIdx.x * BW_THREADS + threadIdx.x;
    for (; i + (BW_UNROLL - 1) * stride < n; i += BW_UNROLL * stride) {
        f4 x[BW_UNROLL], y[BW_UNROLL];
#pragma unroll
        for (int u = 0; u < BW_UNROLL; u++) { x[u] = NT ? __builtin_nontemporal_load(&b[i + u * stride]) : b[i + u * stride]; y[u] = NT ? __builtin_nontemporal_load(&c[i + u * stride]) : c[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < BW_UNROLL; u++) {
            if (NT) __builtin_nontemporal_store(x[u] + s * y[u], &a[i + u * stride]); else a[i + u * stride] = x[u] + s * y[u];
        }
    }
    for (; i < n; i += stride) {
        a[i] = b[i] + s * c[i];
    }
}

__global__ __launch_bounds__(BW_THREADS) void k_bw_fill(f4 *p, size_t n, float v)
{
    const size_t stride = (size_t)gridDim.x * BW_THREADS;
    for (size_t i = (size_t)blockIdx.x * BW_THREADS + threadIdx.x; i < n; i += stride) p[i] = f4{v, v + 1.0f, v + 2.0f, v + 3.0f};
}

}  // namespace

extern "C" int sdfhip_device_bandwidth(int device, uint64_t bytes, uint32_t reps, double *copy_gbs, double *triad_gbs)
{
    clear_error();
    if (!copy_gbs && !triad_gbs) return fail(SDFHIP_ERR_ARG, "device_bandwidth: nothing asked for");
    if (bytes < (1ull << 20) || reps == 0 || reps > 1000) return fail(SDFHIP_ERR_ARG, "device_bandwidth: arrays of at least 1 MiB, 1..1000 repetitions");
    if (copy_gbs) *copy_gbs = 0.0;
    if (triad_gbs) *triad_gbs = 0.0;
    DeviceGuard g(device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "device_bandwidth: no device %d", device);
    const size_t n = (size_t)(bytes / sizeof(f4));
    f4 *buf[3] = { nullptr, nullptr, nullptr };
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = SDFHIP_OK;
    auto cleanup = [&]() {
        for (f4 *p : buf) if (p) (void)hipFree(p);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (st) (void)hipStreamDestroy(st);
    };
    auto tryhip = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == SDFHIP_OK) rc = fail(e == hipErrorOutOfMemory ? SDFHIP_ERR_NOMEM : SDFHIP_ERR_DEVICE, "device_bandwidth: %s: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    const int narr = triad_gbs ? 3 : 2;
    for (int i = 0; i < narr && rc == SDFHIP_OK; i++) tryhip(hipMalloc((void **)&buf[i], n * sizeof(f4)), "hipMalloc");
    if (rc == SDFHIP_OK) tryhip(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
    if (rc == SDFHIP_OK) tryhip(hipEventCreate(&e0), "hipEventCreate");
    if (rc == SDFHIP_OK) tryhip(hipEventCreate(&e1), "hipEventCreate");
    if (rc != SDFHIP_OK) { cleanup(); return rc; }
    for (int i = 0; i < narr; i++) hipLaunchKernelGGL(k_bw_fill, dim3(4096), dim3(BW_THREADS), 0, st, buf[i], n, (float)i);
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    // workgroups per CU: 8 x 256 threads fill a CU's 32 wave slots; fewer leave room in the memory pipeline, more only queue
    const uint32_t per_cu[] = { 4u, 8u, 16u, 32u };
    for (int which = 0; which < 2 && rc == SDFHIP_OK; which++) {
        double *out = which == 0 ? copy_gbs : triad_gbs;
        if (!out) continue;
        const double moved = (which == 0 ? 2.0 : 3.0) * (double)n * sizeof(f4);
        double best = 0.0;
        for (uint32_t trial = 0; trial < 8; trial++) {   // 4 grid sizes x {plain, non-temporal} loads and stores
            const dim3 grid((uint32_t)cus * per_cu[trial >> 1]);
            const bool nt = (trial & 1u) != 0;
            auto launch = [&]() {
                if (which == 0) {
                    if (nt) hipLaunchKernelGGL(k_bw_copy<true>, grid, dim3(BW_THREADS), 0, st, buf[1], buf[0], n);
                    else    hipLaunchKernelGGL(k_bw_copy<false>, grid, dim3(BW_THREADS), 0, st, buf[1], buf[0], n);
                } else {
                    if (nt) hipLaunchKernelGGL(k_bw_triad<true>, grid, dim3(BW_THREADS), 0, st, buf[0], buf[1], buf[2], 0.5f, n);
                    else    hipLaunchKernelGGL(k_bw_triad<false>, grid, dim3(BW_THREADS), 0, st, buf[0], buf[1], buf[2], 0.5f, n);
                }
            };
            launch();                                   // warm-up (page tables, clocks)
            if (!tryhip(hipEventRecord(e0, st), "hipEventRecord")) break;
            for (uint32_t r = 0; r < reps; r++) launch();
            if (!tryhip(hipEventRecord(e1, st), "hipEventRecord")) break;
            if (!tryhip(hipEventSynchronize(e1), "hipEventSynchronize")) break;
            float ms = 0.0f;
            if (!tryhip(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime")) break;
            if (ms > 0.0f) { const double gbs = moved * reps / (ms * 1e-3) / 1e9; if (gbs > best) best = gbs; }
        }
        *out = best;
    }
    if (rc == SDFHIP_OK) tryhip(hipStreamSynchronize(st), "hipStreamSynchronize");
    cleanup();
    return rc;
}
