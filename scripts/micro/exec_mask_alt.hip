// Dev microbenchmark, third of the series (exec_mask_rate / exec_mask_clock): VALU instructions under a
// sparse EXEC mask occupy the SIMD ~3x longer.  Does that cost attach to every sparse instruction, or does
// it take a stretch of sparse execution to set in?  One wave per SIMD alternates segments of SEG blocks of 16
// VALU instructions with all lanes on and with only lane 0 on, and times both kinds with s_memtime.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/exec_mask_alt.hip -o scripts/micro/exec_mask_alt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define BODY16  x0 = (x0 ^ a) + b; x1 = (x1 ^ a) + b; x2 = (x2 ^ a) + b; x3 = (x3 ^ a) + b; \
                x4 = (x4 ^ a) + b; x5 = (x5 ^ a) + b; x6 = (x6 ^ a) + b; x7 = (x7 ^ a) + b;
__global__ __launch_bounds__(64) void k_alt(unsigned long long *out, int rounds, int seg, unsigned a, unsigned b, int sparse_lanes)
{
    unsigned x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned long long td = 0, ts = 0;
    const bool on = (int)threadIdx.x < sparse_lanes;
    for (int r = 0; r < rounds; r++) {
        unsigned long long t0 = clock64();
        for (int i = 0; i < seg; i++) { BODY16 }
        unsigned long long t1 = clock64();
        if (on) {
            for (int i = 0; i < seg; i++) { BODY16 }
        }
        unsigned long long t2 = clock64();
        td += t1 - t0; ts += t2 - t1;
    }
    if (threadIdx.x == 0) {
        out[blockIdx.x * 3 + 0] = td; out[blockIdx.x * 3 + 1] = ts; out[blockIdx.x * 3 + 2] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s: %d CUs; segments of SEG x 16 VALU instructions, alternating all lanes / few lanes\n", p.gcnArchName, cus);
    for (int wps = 1; wps <= 8; wps *= 8)
        for (int lanes = 1; lanes <= 16; lanes *= 16)
            for (int seg = 1; seg <= 16384; seg *= 4) {
                const int blocks = cus * 4 * wps, rounds = (1 << 20) / (seg * 16) > 4 ? (1 << 20) / (seg * 16) : 4;
                unsigned long long *o; (void)hipMalloc(&o, (size_t)blocks * 3 * 8);
                for (int rep = 0; rep < 2; rep++) {
                    hipLaunchKernelGGL(k_alt, dim3(blocks), dim3(64), 0, 0, o, rounds, seg, 0x9e3779b9u, 12345u, lanes);
                    (void)hipDeviceSynchronize();
                }
                std::vector<unsigned long long> h((size_t)blocks * 3);
                (void)hipMemcpy(h.data(), o, h.size() * 8, hipMemcpyDeviceToHost);
                double td = 0, ts = 0;
                for (int b = 0; b < blocks; b++) { td += h[b * 3]; ts += h[b * 3 + 1]; }
                const double n = (double)blocks * rounds * seg * 16;
                printf("%d wave(s)/SIMD, sparse = %2d lane(s), segment %7d instr: dense %6.2f, sparse %6.2f ticks per instruction\n",
                       wps, lanes, seg * 16, td / n, ts / n);
                (void)hipFree(o);
            }
    return 0;
}
