// Dev microbenchmark: the issue cost of the VALU instructions the ray-march loop is made of, one kind at a time, with
// 8 waves per SIMD -- and of a VALU instruction under an EMPTY exec mask.  (The frame's counted VALU instructions
// divided by the frame time come out at 3.4-3.7 cycles each, below the 4.08 of scripts/micro/valu_rate.hip: which
// instructions are cheaper than a 16-lane pass x 4?)
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/valu_mix.hip -o scripts/micro/valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define KERNEL(name, body)                                                                   \
    __global__ __launch_bounds__(64) void name(float *out, int iters, float a)               \
    {                                                                                        \
        float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3; int s = 0;            \
        for (int i = 0; i < iters; i++) { asm volatile(R16(body) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s) : "v"(a) : "vcc"); } \
        out[blockIdx.x * 64 + threadIdx.x] = v0 + v1 + v2 + v3;                              \
    }
// 4 independent instructions per body line, 16 repeats = 64 instructions per loop iteration
KERNEL(k_mov,    "v_mov_b32 %0, %5\n v_mov_b32 %1, %5\n v_mov_b32 %2, %5\n v_mov_b32 %3, %5\n")
KERNEL(k_xor,    "v_xor_b32 %0, %0, %5\n v_xor_b32 %1, %1, %5\n v_xor_b32 %2, %2, %5\n v_xor_b32 %3, %3, %5\n")
KERNEL(k_mul,    "v_mul_f32 %0, %0, %5\n v_mul_f32 %1, %1, %5\n v_mul_f32 %2, %2, %5\n v_mul_f32 %3, %3, %5\n")
KERNEL(k_fma,    "v_fma_f32 %0, %0, %5, %5\n v_fma_f32 %1, %1, %5, %5\n v_fma_f32 %2, %2, %5, %5\n v_fma_f32 %3, %3, %5, %5\n")
KERNEL(k_fract,  "v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3\n")
KERNEL(k_cvtflr, "v_cvt_flr_i32_f32 %0, %0\n v_cvt_flr_i32_f32 %1, %1\n v_cvt_flr_i32_f32 %2, %2\n v_cvt_flr_i32_f32 %3, %3\n")
KERNEL(k_ubyte,  "v_cvt_f32_ubyte1 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte1 %2, %2\n v_cvt_f32_ubyte1 %3, %3\n")
KERNEL(k_med3,   "v_med3_i32 %0, %0, 0, %5\n v_med3_i32 %1, %1, 0, %5\n v_med3_i32 %2, %2, 0, %5\n v_med3_i32 %3, %3, 0, %5\n")
KERNEL(k_min3,   "v_min3_f32 %0, %0, %1, %5\n v_min3_f32 %1, %1, %2, %5\n v_min3_f32 %2, %2, %3, %5\n v_min3_f32 %3, %3, %0, %5\n")
KERNEL(k_cmp,    "v_cmp_lt_f32 vcc, %0, %5\n v_cmp_lt_f32 vcc, %1, %5\n v_cmp_lt_f32 vcc, %2, %5\n v_cmp_lt_f32 vcc, %3, %5\n")
KERNEL(k_cndmask,"v_cndmask_b32 %0, %0, %5, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %5, vcc\n v_cndmask_b32 %3, %3, %5, vcc\n")
KERNEL(k_lshl_or,"v_lshl_or_b32 %0, %0, 3, %5\n v_lshl_or_b32 %1, %1, 3, %5\n v_lshl_or_b32 %2, %2, 3, %5\n v_lshl_or_b32 %3, %3, 3, %5\n")
KERNEL(k_fmac,   "v_fmac_f32 %0, %1, %5\n v_fmac_f32 %1, %2, %5\n v_fmac_f32 %2, %3, %5\n v_fmac_f32 %3, %0, %5\n")
KERNEL(k_sub,    "v_sub_f32 %0, %0, %5\n v_sub_f32 %1, %1, %5\n v_sub_f32 %2, %2, %5\n v_sub_f32 %3, %3, %5\n")
KERNEL(k_addf,   "v_add_f32 %0, %0, %5\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %5\n")
KERNEL(k_and,    "v_and_b32 %0, %0, %5\n v_and_b32 %1, %1, %5\n v_and_b32 %2, %2, %5\n v_and_b32 %3, %3, %5\n")
KERNEL(k_addu,   "v_add_u32 %0, %0, %5\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5\n")
KERNEL(k_lshl,   "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n")
KERNEL(k_lshladd,"v_lshl_add_u32 %0, %0, 3, %5\n v_lshl_add_u32 %1, %1, 3, %5\n v_lshl_add_u32 %2, %2, 3, %5\n v_lshl_add_u32 %3, %3, 3, %5\n")
KERNEL(k_andor,  "v_and_or_b32 %0, %0, 7, %5\n v_and_or_b32 %1, %1, 7, %5\n v_and_or_b32 %2, %2, 7, %5\n v_and_or_b32 %3, %3, 7, %5\n")
KERNEL(k_bfe,    "v_bfe_u32 %0, %0, 1, 9\n v_bfe_u32 %1, %1, 1, 9\n v_bfe_u32 %2, %2, 1, 9\n v_bfe_u32 %3, %3, 1, 9\n")
KERNEL(k_max3u,  "v_max3_u32 %0, %0, %1, %5\n v_max3_u32 %1, %1, %2, %5\n v_max3_u32 %2, %2, %3, %5\n v_max3_u32 %3, %3, %0, %5\n")
KERNEL(k_cvtfi,  "v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n")
KERNEL(k_floor,  "v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n")
KERNEL(k_cmpu,   "v_cmp_lt_u32 vcc, %0, %5\n v_cmp_lt_u32 vcc, %1, %5\n v_cmp_lt_u32 vcc, %2, %5\n v_cmp_lt_u32 vcc, %3, %5\n")
KERNEL(k_cmp64,  "v_cmp_lt_f32 s[20:21], %0, %5\n v_cmp_lt_f32 s[22:23], %1, %5\n v_cmp_lt_f32 s[24:25], %2, %5\n v_cmp_lt_f32 s[26:27], %3, %5\n")
KERNEL(k_mulclamp,"v_mul_f32_e64 %0, %0, %5 clamp\n v_mul_f32_e64 %1, %1, %5 clamp\n v_mul_f32_e64 %2, %2, %5 clamp\n v_mul_f32_e64 %3, %3, %5 clamp\n")
KERNEL(k_mov_fma,"v_mov_b32 %0, %5\n v_fma_f32 %1, %1, %5, %5\n v_mov_b32 %2, %5\n v_fma_f32 %3, %3, %5, %5\n")
KERNEL(k_mul_fma,"v_mul_f32 %0, %0, %5\n v_fma_f32 %1, %1, %5, %5\n v_mul_f32 %2, %2, %5\n v_fma_f32 %3, %3, %5, %5\n")
// packed fp32: two floats per lane and instruction (register pairs)
#define KERNEL2(name, body)                                                                  \
    __global__ __launch_bounds__(64) void name(float *out, int iters, float a)               \
    {                                                                                        \
        typedef float f2 __attribute__((ext_vector_type(2)));                                \
        f2 v0 = {(float)threadIdx.x, 1.f}, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f, b = {a, a}; int s = 0; \
        for (int i = 0; i < iters; i++) { asm volatile(R16(body) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s) : "v"(b) : "vcc"); } \
        f2 r = v0 + v1 + v2 + v3; out[blockIdx.x * 64 + threadIdx.x] = r.x + r.y;             \
    }
KERNEL2(k_pkfma, "v_pk_fma_f32 %0, %0, %5, %5\n v_pk_fma_f32 %1, %1, %5, %5\n v_pk_fma_f32 %2, %2, %5, %5\n v_pk_fma_f32 %3, %3, %5, %5\n")
KERNEL2(k_pkmul, "v_pk_mul_f32 %0, %0, %5\n v_pk_mul_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %5\n v_pk_mul_f32 %3, %3, %5\n")
KERNEL2(k_pkadd, "v_pk_add_f32 %0, %0, %5\n v_pk_add_f32 %1, %1, %5\n v_pk_add_f32 %2, %2, %5\n v_pk_add_f32 %3, %3, %5\n")
KERNEL(k_exec0,  "s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0\n v_fma_f32 %0, %0, %5, %5\n v_fma_f32 %1, %1, %5, %5\n v_fma_f32 %2, %2, %5, %5\n v_fma_f32 %3, %3, %5, %5\n v_xor_b32 %0, %0, %5\n v_xor_b32 %1, %1, %5\n v_xor_b32 %2, %2, %5\n v_xor_b32 %3, %3, %5\n s_mov_b64 exec, s[20:21]\n")
struct K { const char *name; void (*fn)(float *, int, float); int per_iter; };
int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 1024, wps = 8, blocks = cus * 4 * wps;
    float *o; (void)hipMalloc(&o, (size_t)blocks * 64 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const K ks[] = { {"v_mov_b32", k_mov, 64}, {"v_xor_b32", k_xor, 64}, {"v_mul_f32", k_mul, 64}, {"v_fma_f32", k_fma, 64}, {"v_fract_f32", k_fract, 64},
                     {"v_cvt_flr_i32_f32", k_cvtflr, 64}, {"v_cvt_f32_ubyte1", k_ubyte, 64}, {"v_med3_i32", k_med3, 64}, {"v_min3_f32", k_min3, 64},
                     {"v_cmp_lt_f32 -> vcc", k_cmp, 64}, {"v_cndmask_b32", k_cndmask, 64}, {"v_lshl_or_b32", k_lshl_or, 64},
                     {"v_fmac_f32", k_fmac, 64}, {"v_sub_f32", k_sub, 64}, {"v_add_f32", k_addf, 64}, {"v_and_b32", k_and, 64}, {"v_add_u32", k_addu, 64},
                     {"v_lshlrev_b32", k_lshl, 64}, {"v_lshl_add_u32", k_lshladd, 64}, {"v_and_or_b32", k_andor, 64}, {"v_bfe_u32", k_bfe, 64},
                     {"v_max3_u32", k_max3u, 64}, {"v_cvt_f32_i32", k_cvtfi, 64}, {"v_floor_f32", k_floor, 64}, {"v_cmp_lt_u32 -> vcc", k_cmpu, 64},
                     {"v_cmp_lt_f32 -> sgpr pair (e64)", k_cmp64, 64}, {"v_mul_f32 clamp (e64)", k_mulclamp, 64},
                     {"v_mov_b32 / v_fma_f32 alternating", k_mov_fma, 64}, {"v_mul_f32 / v_fma_f32 alternating", k_mul_fma, 64},
                     {"v_pk_fma_f32 (2 fma per lane)", k_pkfma, 64}, {"v_pk_mul_f32", k_pkmul, 64}, {"v_pk_add_f32", k_pkadd, 64},
                     {"8 VALU under exec = 0 (+3 SALU)", k_exec0, 128} };
    printf("%s, %d CUs, %d waves per SIMD, clock %d MHz\n", p.gcnArchName, cus, wps, p.clockRate / 1000);
    for (const K &k : ks) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(64), 0, 0, o, iters, 1.0001f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double instr = (double)blocks * iters * k.per_iter;
        printf("%-36s %.3f ms -> %.2f cycles per VALU wave instruction, %.0f G/s chip-wide\n", k.name, best,
               (cus * 4) * (p.clockRate / 1e6) / (instr / best / 1e6), instr / best / 1e6);
    }
    return 0;
}
