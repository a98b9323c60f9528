// Dev microbenchmark, follow-up of exec_mask_rate.hip: a VALU instruction under a sparse EXEC mask
// (<= 8 lanes for integer ops, <= 16 for v_fma_f32) takes 5x longer in wall time.  Is that the
// shader clock dropping (power management sees a nearly idle chip) or the instruction itself
// costing more cycles?  Every wave reads s_memtime (shader-clock ticks) and s_memrealtime (100 MHz)
// around its loop; waves with a dense and a sparse mask run side by side in one launch.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/exec_mask_clock.hip -o scripts/micro/exec_mask_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void k_int(unsigned long long *out, int iters, unsigned a, unsigned b,
                                             unsigned long long mask_a, unsigned long long mask_b, int every)
{
    const unsigned long long lanes = (blockIdx.x % every) == 0 ? mask_b : mask_a;
    if (!((lanes >> threadIdx.x) & 1ull)) return;
    unsigned x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const unsigned long long t0 = clock64(), r0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x0 = (x0 ^ a) + b; x1 = (x1 ^ a) + b; x2 = (x2 ^ a) + b; x3 = (x3 ^ a) + b;
            x4 = (x4 ^ a) + b; x5 = (x5 ^ a) + b; x6 = (x6 ^ a) + b; x7 = (x7 ^ a) + b;
        }
    }
    const unsigned long long t1 = clock64(), r1 = wall_clock64();
    const unsigned s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (__ffsll((long long)lanes) - 1 == (int)threadIdx.x) {
        out[blockIdx.x * 3 + 0] = t1 - t0;
        out[blockIdx.x * 3 + 1] = r1 - r0;
        out[blockIdx.x * 3 + 2] = s;
    }
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 2048;
    printf("%s: %d CUs, clock %d MHz; every wave: %d x 128 VALU instructions\n", p.gcnArchName, cus, p.clockRate / 1000, iters);
    struct C { unsigned long long a, b; int every; int wps; const char *name; };
    const C cases[] = {
        { ~0ull, ~0ull, 1, 8, "all waves dense (64 lanes)" },
        { 1ull, 1ull, 1, 8, "all waves sparse (lane 0)" },
        { ~0ull, 1ull, 2, 8, "every 2nd wave sparse" },
        { ~0ull, 1ull, 16, 8, "every 16th wave sparse" },
        { ~0ull, 1ull, 256, 8, "every 256th wave sparse" },
        { 1ull, ~0ull, 16, 8, "every 16th wave dense, rest sparse" },
        { ~0ull, ~0ull, 1, 1, "1 wave per SIMD, dense" },
        { 1ull, 1ull, 1, 1, "1 wave per SIMD, sparse" },
        { ~0ull, 1ull, 2, 1, "1 wave per SIMD, every 2nd sparse" },
        { 0xFFull, 0xFFull, 1, 8, "all waves 8 lanes" },
        { 0xFFFFull, 0xFFFFull, 1, 8, "all waves 16 lanes" },
        { 0xFFFull, 0xFFFull, 1, 8, "all waves 12 lanes" },
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (const C &c : cases) {
        const int blocks = cus * 4 * c.wps;
        unsigned long long *o; hipMalloc(&o, (size_t)blocks * 3 * 8);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_int, dim3(blocks), dim3(64), 0, 0, o, iters, 0x9e3779b9u, 12345u, c.a, c.b, c.every);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<unsigned long long> h((size_t)blocks * 3);
        hipMemcpy(h.data(), o, h.size() * 8, hipMemcpyDeviceToHost);
        double ta = 0, ra = 0, tb = 0, rb = 0; int na = 0, nb = 0;
        for (int b = 0; b < blocks; b++) {
            if (b % c.every == 0) { tb += h[b * 3]; rb += h[b * 3 + 1]; nb++; } else { ta += h[b * 3]; ra += h[b * 3 + 1]; na++; }
        }
        const double instr = (double)iters * 128;
        printf("%-36s kernel %7.3f ms |", c.name, ms);
        if (na) printf(" class A (%5d waves): %6.2f s_memtime ticks, %7.2f ns per instr |", na, ta / na / instr, ra / na / instr * 10.0);
        printf(" class B (%5d waves): %6.2f ticks, %7.2f ns per instr\n", nb, tb / nb / instr, rb / nb / instr * 10.0);
        hipFree(o);
    }
    return 0;
}
