// Dev microbenchmark: how many VALU wave-instructions per second does an MI355X issue?
// (to read SQ_INSTS_VALU per frame against)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k_fma(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ __launch_bounds__(64) void k_int(unsigned *out, int iters, unsigned a, unsigned b, unsigned long long lanes = ~0ull)
{
    if (!((lanes >> threadIdx.x) & 1ull)) return;          // partial EXEC: which lanes run the loop
    unsigned x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x0 = (x0 ^ a) + b; x1 = (x1 ^ a) + b; x2 = (x2 ^ a) + b; x3 = (x3 ^ a) + b;
            x4 = (x4 ^ a) + b; x5 = (x5 ^ a) + b; x6 = (x6 ^ a) + b; x7 = (x7 ^ a) + b;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 4096;
    printf("%s: %d CUs, clock %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = cus * 4 * wps;
        float *o; hipMalloc(&o, (size_t)blocks * 64 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int kind = 0; kind < 2; kind++) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(64), 0, 0, o, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(k_int, dim3(blocks), dim3(64), 0, 0, (unsigned *)o, iters, 0x9e3779b9u, 12345u);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double instr = (double)blocks * iters * (kind == 0 ? 32 : 128);   // wave instructions: 64 fma = 32 v_pk_fma_f32; 64 x (v_xor + v_add)
            printf("%s, %d waves per SIMD: %.3f ms, %.1f G wave-instr/s = %.2f per SIMD per ns -> %.2f cycles per wave instruction at %d MHz\n",
                   kind == 0 ? "v_pk_fma_f32" : "v_xor+v_add", wps, best, instr / best / 1e6, instr / best / 1e6 / (cus * 4),
                   (cus * 4) * (p.clockRate / 1e6) / (instr / best / 1e6), p.clockRate / 1000);
        }
        hipFree(o);
    }
    // does a VALU instruction cost less when whole 16-lane quarters of EXEC are off?
    {
        const int blocks = cus * 4 * 8;
        unsigned *o; hipMalloc(&o, (size_t)blocks * 64 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const unsigned long long masks[] = { ~0ull, 0xFFFFFFFFull, 0xFFFFull, 0x1ull, 0x0001000100010001ull, 0x00000000FFFF0000ull, 0xFFFF0000FFFF0000ull };
        const char *names[] = { "all 64 lanes", "lanes 0-31", "lanes 0-15", "lane 0", "one lane per quarter", "lanes 16-31", "lanes 16-31 and 48-63" };
        for (int m = 0; m < 7; m++) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_int, dim3(blocks), dim3(64), 0, 0, o, iters, 0x9e3779b9u, 12345u, masks[m]);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double instr = (double)blocks * iters * 64 * 2;
            printf("EXEC = %-24s: %.3f ms -> %.2f cycles per wave instruction\n", names[m], best, (cus * 4) * (p.clockRate / 1e6) / (instr / best / 1e6));
        }
    }
    return 0;
}
