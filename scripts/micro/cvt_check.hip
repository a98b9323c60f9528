// Dev check: v_cvt_flr_i32_f32 against (int)floorf(x), and v_fract_f32(x) == 0 against x == floorf(x), over edge cases and a sweep.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstring>
__global__ void k(const float *in, int *a, int *b, int *g1, int *g2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float u = in[i];
    int x; asm("v_cvt_flr_i32_f32_e32 %0, %1" : "=v"(x) : "v"(u));
    a[i] = x;
    float f = floorf(u);
    b[i] = (int)__builtin_amdgcn_fmed3f(f, -2147483648.0f, 2147483520.0f);
    g1[i] = __builtin_amdgcn_fractf(u) == 0.0f;
    g2[i] = u == f;
}
int main()
{
    std::vector<float> v = { 0.f, -0.f, 0.5f, -0.5f, 1.f, -1.f, 4095.999f, 4096.f, -1228.8f, 1e-40f, -1e-40f, 8388607.5f, 8388608.f, 16777216.f, 3e9f, -3e9f, 2147483648.f, INFINITY, -INFINITY, NAN,
                             0.99999994f, -0.99999994f, 1.0000001f, 123.00001f, -123.00001f, 2047.9999f, -7.0f, 7.0f };
    unsigned s = 12345u;
    for (int i = 0; i < 2000000; i++) { s = s * 1664525u + 1013904223u; unsigned bits = s; float f; memcpy(&f, &bits, 4); v.push_back(f); }
    for (int i = 0; i < 1000000; i++) { s = s * 1664525u + 1013904223u; v.push_back(((int)(s >> 8) - (1 << 23)) / 1024.0f); }      // +-8192 in steps of 2^-10
    int n = (int)v.size();
    float *d; int *a, *b, *g1, *g2;
    hipMalloc(&d, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&g1, n * 4); hipMalloc(&g2, n * 4);
    hipMemcpy(d, v.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, d, a, b, g1, g2, n);
    std::vector<int> ha(n), hb(n), h1(n), h2(n);
    hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h1.data(), g1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), g2, n * 4, hipMemcpyDeviceToHost);
    int bad_cvt = 0, bad_grid = 0, bad_small = 0;
    for (int i = 0; i < n; i++) {
        bool fin = std::isfinite(v[i]) && fabsf(v[i]) < 2e9f;
        if (fin && ha[i] != hb[i]) { if (bad_cvt++ < 10) printf("cvt  x=%.9g (%a): flr=%d floor=%d\n", v[i], v[i], ha[i], hb[i]); }
        if (h1[i] != h2[i]) { if (bad_grid++ < 10) printf("grid x=%.9g (%a): fract==0 %d, x==floor %d\n", v[i], v[i], h1[i], h2[i]); if (fabsf(v[i]) < 8192.f) bad_small++; }
        if (i < 28) printf("x=%-14.9g flr=%d floor=%d fract0=%d eqfloor=%d\n", v[i], ha[i], hb[i], h1[i], h2[i]);
    }
    printf("n=%d: cvt mismatches (finite, |x|<2e9) %d; on-grid mismatches %d (of which |x|<8192: %d)\n", n, bad_cvt, bad_grid, bad_small);
    return 0;
}
