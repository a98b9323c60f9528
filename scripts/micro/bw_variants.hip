// Dev microbenchmark: which streaming kernel shape reaches the chip's HBM rate?  (csrc/bandwidth.hip's copy / triad reached 5.1-5.3
// TB/s on the first try; scripts/micro/fetch_granule.hip's read-only stream 6.4; the guide quotes 6.29 for a float4 copy.)
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/bw_variants scripts/micro/bw_variants.hip     run: scripts/micro/bw_variants [GiB]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

// grid-stride, U accesses in flight per lane, NT = non-temporal
template <int U, bool NT, int BT>
__global__ __launch_bounds__(BT) void k_copy_gs(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * BT;
    size_t i = (size_t)blockIdx.x * BT + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v[u], &dst[i + u * stride]); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
// block-contiguous: a workgroup owns whole chunks of U * BT float4 (U consecutive rows of BT), chunks dealt grid-stride
template <int U, bool NT, int BT>
__global__ __launch_bounds__(BT) void k_copy_chunk(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n)
{
    const size_t chunk = (size_t)U * BT, nchunks = n / chunk;
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t base = c * chunk + threadIdx.x;
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(&src[base + (size_t)u * BT]) : src[base + (size_t)u * BT];
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v[u], &dst[base + (size_t)u * BT]); else dst[base + (size_t)u * BT] = v[u]; }
    }
}
// one float4 per thread, no loop
__global__ __launch_bounds__(256) void k_copy_flat(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const f4 *__restrict__ src, float *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
    }
    for (; i < n; i += stride) acc += src[i];
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123456.789f) out[0] = s;                    // (never true: keeps the loads)
}
template <bool NT>
__global__ __launch_bounds__(256) void k_write(f4 *__restrict__ dst, size_t n, float v)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const f4 x = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { if (NT) __builtin_nontemporal_store(x, &dst[i]); else dst[i] = x; }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_triad(f4 *__restrict__ a, const f4 *__restrict__ b, const f4 *__restrict__ c, float s, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; u++) { x[u] = NT ? __builtin_nontemporal_load(&b[i + u * stride]) : b[i + u * stride]; y[u] = NT ? __builtin_nontemporal_load(&c[i + u * stride]) : c[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(x[u] + s * y[u], &a[i + u * stride]); else a[i + u * stride] = x[u] + s * y[u]; }
    }
    for (; i < n; i += stride) a[i] = b[i] + s * c[i];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 2;
    const size_t bytes = gib << 30, n = bytes / sizeof(f4);
    f4 *a, *b, *c; float *out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 1, bytes)); CK(hipMemset(c, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 10;
    auto timeit = [&](const char *name, double moved, auto launch) {
        launch(); hipEventRecord(e0, 0);
        for (int r = 0; r < reps; r++) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f GB/s\n", name, moved * reps / (ms * 1e-3) / 1e9); fflush(stdout);
    };
    char nm[128];
    for (int wg : {1024, 2048, 4096, 8192, 16384}) {
        snprintf(nm, sizeof nm, "copy grid-stride U4 plain, %d wgs", wg); timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy_gs<4, false, 256>), dim3(wg), dim3(256), 0, 0, b, a, n); });
        snprintf(nm, sizeof nm, "copy grid-stride U4 nt,    %d wgs", wg); timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy_gs<4, true, 256>), dim3(wg), dim3(256), 0, 0, b, a, n); });
        snprintf(nm, sizeof nm, "copy grid-stride U8 plain, %d wgs", wg); timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy_gs<8, false, 256>), dim3(wg), dim3(256), 0, 0, b, a, n); });
        snprintf(nm, sizeof nm, "copy grid-stride U1 plain, %d wgs", wg); timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy_gs<1, false, 256>), dim3(wg), dim3(256), 0, 0, b, a, n); });
        snprintf(nm, sizeof nm, "copy grid-stride U2 1024t, %d wgs", wg / 4); timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy_gs<2, false, 1024>), dim3(wg / 4), dim3(1024), 0, 0, b, a, n); });
        snprintf(nm, sizeof nm, "copy chunk U4 plain,       %d wgs", wg); timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy_chunk<4, false, 256>), dim3(wg), dim3(256), 0, 0, b, a, n); });
        snprintf(nm, sizeof nm, "copy chunk U8 nt,          %d wgs", wg); timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy_chunk<8, true, 256>), dim3(wg), dim3(256), 0, 0, b, a, n); });
        snprintf(nm, sizeof nm, "read U4 plain,             %d wgs", wg); timeit(nm, 1.0 * bytes, [&] { hipLaunchKernelGGL((k_read<4, false>), dim3(wg), dim3(256), 0, 0, b, out, n); });
        snprintf(nm, sizeof nm, "read U8 nt,                %d wgs", wg); timeit(nm, 1.0 * bytes, [&] { hipLaunchKernelGGL((k_read<8, true>), dim3(wg), dim3(256), 0, 0, b, out, n); });
        snprintf(nm, sizeof nm, "write plain,               %d wgs", wg); timeit(nm, 1.0 * bytes, [&] { hipLaunchKernelGGL((k_write<false>), dim3(wg), dim3(256), 0, 0, a, n, 1.0f); });
        snprintf(nm, sizeof nm, "write nt,                  %d wgs", wg); timeit(nm, 1.0 * bytes, [&] { hipLaunchKernelGGL((k_write<true>), dim3(wg), dim3(256), 0, 0, a, n, 1.0f); });
        snprintf(nm, sizeof nm, "triad U4 plain,            %d wgs", wg); timeit(nm, 3.0 * bytes, [&] { hipLaunchKernelGGL((k_triad<4, false>), dim3(wg), dim3(256), 0, 0, a, b, c, 0.5f, n); });
        snprintf(nm, sizeof nm, "triad U2 nt,               %d wgs", wg); timeit(nm, 3.0 * bytes, [&] { hipLaunchKernelGGL((k_triad<2, true>), dim3(wg), dim3(256), 0, 0, a, b, c, 0.5f, n); });
    }
    timeit("copy one float4 per thread (no loop)", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy_flat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, b, a, n); });
    timeit("hipMemcpyAsync device to device", 2.0 * bytes, [&] { hipMemcpyAsync(a, b, bytes, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
