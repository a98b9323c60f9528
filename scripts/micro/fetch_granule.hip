// Dev microbenchmark: what does FETCH_SIZE count for SCATTERED 16-byte loads (the access shape of the march kernels' cell
// lookups), and in what granule does the L2 fetch a missed line from memory?  MI355X_MICROARCH.md calibrates FETCH_SIZE for
// wide coalesced streams only (it reports half their bytes) and says to calibrate any other shape on a known byte count.
//
// A table of 2^25 lines of 128 bytes (4 GiB: 16 x the Infinity Cache).  Every thread owns ONE line, chosen by an odd
// multiplier (a bijection of the line index: no line is touched twice, neighbours in a wave are far apart), and loads
//   one16:   16 bytes at +0
//   two64:   16 bytes at +0 and at +64      (the two halves of the line)
//   two32:   16 bytes at +0 and at +32      (two 32-byte sectors of one half)
//   four32:  16 bytes at +0, +32, +64, +96  (every sector)
//   one4:     4 bytes at +0
//   stream:  the whole table, 16 bytes per lane, coalesced (the guide's calibrated shape)
// Run plainly it prints times and lines per second; under `rocprofv3 --pmc FETCH_SIZE` (and, separately, TCC_EA0_RDREQ_sum
// TCC_EA0_RDREQ_32B_sum / TCC_MISS_sum TCC_REQ_sum) the per-kernel counters divided by 2^25 lines say what a line costs.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr uint32_t LINE_BITS = 25, LINES = 1u << LINE_BITS;

template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const uint4 *__restrict__ table, uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t line = (i * 0x9E3779B1u) & (LINES - 1u);
    const uint4 *p = table + (size_t)line * 8;             // 8 x 16 bytes per line
    uint32_t acc;
    if (MODE == 4) {
        acc = *reinterpret_cast<const uint32_t *>(p);
    } else {
        uint4 a = p[0];
        acc = a.x ^ a.y ^ a.z ^ a.w;
        if (MODE == 1 || MODE == 3) { uint4 b = p[4]; acc ^= b.x ^ b.y ^ b.z ^ b.w; }
        if (MODE == 2 || MODE == 3) { uint4 b = p[2]; acc ^= b.x ^ b.y ^ b.z ^ b.w; }
        if (MODE == 3) { uint4 b = p[6]; acc ^= b.x ^ b.y ^ b.z ^ b.w; }
    }
    if (acc == 0x12345678u) out[0] = i;                    // never true for the table's contents; keeps the loads
}

__global__ __launch_bounds__(256) void k_stream(const uint4 *__restrict__ table, uint32_t *__restrict__ out)
{
    // 8 x 16 bytes per thread, consecutive threads consecutive 16-byte words
    const size_t base = (size_t)blockIdx.x * 256u * 8u + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { uint4 a = table[base + (size_t)k * 256u]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) out[0] = (uint32_t)base;
}

__global__ void k_fill(uint4 *table, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        table[i] = make_uint4(1u, 2u, 4u, 8u);
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    const size_t words = (size_t)LINES * 8;
    uint4 *table; uint32_t *out;
    CHECK(hipMalloc(&table, words * 16));
    CHECK(hipMalloc(&out, 64));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, table, words);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[] = { "one16", "two64", "two32", "four32", "one4", "stream" };
    for (int mode = 0; mode < 6; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            const dim3 g(LINES / 256), b(256);
            switch (mode) {
            case 0: hipLaunchKernelGGL(k_gather<0>, g, b, 0, 0, table, out); break;
            case 1: hipLaunchKernelGGL(k_gather<1>, g, b, 0, 0, table, out); break;
            case 2: hipLaunchKernelGGL(k_gather<2>, g, b, 0, 0, table, out); break;
            case 3: hipLaunchKernelGGL(k_gather<3>, g, b, 0, 0, table, out); break;
            case 4: hipLaunchKernelGGL(k_gather<4>, g, b, 0, 0, table, out); break;
            default: hipLaunchKernelGGL(k_stream, g, b, 0, 0, table, out); break;
            }
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double lps = LINES / (best * 1e-3);
        printf("%-7s %8.3f ms  %7.2f G lines/s  = %6.2f TB/s if a line costs 128 B, %6.2f if 64 B, %6.2f if 32 B\n", names[mode], best, lps / 1e9,
               lps * 128 / 1e12, lps * 64 / 1e12, lps * 32 / 1e12);
    }
    CHECK(hipFree(table)); CHECK(hipFree(out));
    return 0;
}
