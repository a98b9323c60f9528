// Dev microbenchmark: what does a VALU wave-instruction cost under a sparse EXEC mask?
// (r01_micro_valu_rate.txt showed 22 cycles per instruction with one lane on against 4 with
// 16 contiguous lanes on: the tail of a frame -- waves with a few 100-140-step pixels alive --
// would run 5x slower per instruction than the body.  This sweeps the mask shapes and the
// occupancy to find the rule.)
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/exec_mask_rate.hip -o /tmp/exec_mask_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k_int(unsigned *out, int iters, unsigned a, unsigned b, unsigned long long lanes)
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
// the same under a mask the loop itself keeps (no early return: EXEC narrowed by a branch around the body)
__global__ __launch_bounds__(64) void k_int_branch(unsigned *out, int iters, unsigned a, unsigned b, unsigned long long lanes)
{
    unsigned x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const bool on = (lanes >> threadIdx.x) & 1ull;
    for (int i = 0; i < iters; i++) {
        if (on) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x0 = (x0 ^ a) + b; x1 = (x1 ^ a) + b; x2 = (x2 ^ a) + b; x3 = (x3 ^ a) + b;
                x4 = (x4 ^ a) + b; x5 = (x5 ^ a) + b; x6 = (x6 ^ a) + b; x7 = (x7 ^ a) + b;
            }
        }
        asm volatile("" : "+v"(x0));
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ __launch_bounds__(64) void k_fma(float *out, int iters, float a, float b, unsigned long long lanes)
{
    if (!((lanes >> threadIdx.x) & 1ull)) return;
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            // odd/even operands differ so that the compiler cannot pack two chains into one v_pk_fma_f32
            x0 = __builtin_fmaf(x0, a, b); asm volatile("" : "+v"(x0));
            x1 = __builtin_fmaf(x1, a, b); asm volatile("" : "+v"(x1));
            x2 = __builtin_fmaf(x2, a, b); asm volatile("" : "+v"(x2));
            x3 = __builtin_fmaf(x3, a, b); asm volatile("" : "+v"(x3));
            x4 = __builtin_fmaf(x4, a, b); asm volatile("" : "+v"(x4));
            x5 = __builtin_fmaf(x5, a, b); asm volatile("" : "+v"(x5));
            x6 = __builtin_fmaf(x6, a, b); asm volatile("" : "+v"(x6));
            x7 = __builtin_fmaf(x7, a, b); asm volatile("" : "+v"(x7));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 2048;
    printf("%s: %d CUs, clock %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
    struct M { unsigned long long m; const char *name; };
    const M masks[] = {
        { ~0ull, "all 64" }, { 0xFFFFFFFFull, "lanes 0-31" }, { 0xFFFFull, "lanes 0-15" }, { 0xFFull, "lanes 0-7" },
        { 0xFull, "lanes 0-3" }, { 0x3ull, "lanes 0-1" }, { 0x1ull, "lane 0" }, { 0x8000000000000000ull, "lane 63" },
        { 0x0001000100010001ull, "1 per quarter" }, { 0x0101010101010101ull, "1 per 8 (8 lanes)" },
        { 0x1111111111111111ull, "1 per 4 (16 lanes)" }, { 0x5555555555555555ull, "every 2nd (32 lanes)" },
        { 0x00FF00FF00FF00FFull, "8 of each 16" }, { 0x000F000F000F000Full, "4 of each 16" },
        { 0x7FFFFFFFFFFFFFFFull, "all but lane 63" }, { 0xFFFFFFFFFFFF0000ull, "lanes 16-63" },
        { 0x0000000000010001ull, "lanes 0 and 16" }, { 0x00000000000000FFull << 20, "lanes 20-27" },
    };
    const int nm = sizeof masks / sizeof masks[0];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 3; kind++) {
        for (int wps = 8; wps >= 1; wps /= 8) {
            const int blocks = cus * 4 * wps;
            unsigned *o; hipMalloc(&o, (size_t)blocks * 64 * 4);
            for (int m = 0; m < nm; m++) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; rep++) {
                    hipEventRecord(e0);
                    if (kind == 0) hipLaunchKernelGGL(k_int, dim3(blocks), dim3(64), 0, 0, o, iters, 0x9e3779b9u, 12345u, masks[m].m);
                    else if (kind == 1) hipLaunchKernelGGL(k_int_branch, dim3(blocks), dim3(64), 0, 0, o, iters, 0x9e3779b9u, 12345u, masks[m].m);
                    else hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(64), 0, 0, (float *)o, iters, 1.0001f, 0.5f, masks[m].m);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                const double instr = (double)blocks * iters * (kind == 2 ? 64 : 128);
                printf("%-12s %d waves/SIMD  EXEC = %-22s: %8.3f ms -> %6.2f cycles per wave instruction\n",
                       kind == 0 ? "int(return)" : kind == 1 ? "int(branch)" : "v_fma_f32", wps, masks[m].name, best,
                       (cus * 4) * (p.clockRate / 1e6) / (instr / best / 1e6));
            }
            hipFree(o);
        }
    }
    return 0;
}
