// Device-side building blocks of the ray-march kernels (gfx950 only).
//
// Arithmetic contract (DESIGN.md "Numerics"; identical to oracle/sdf_oracle.c):
// every fp32 operation is individually rounded in the order written (the
// translation unit is built with -ffp-contract=off; the only fused operations
// are the explicit __builtin_fmaf calls in unorm8(), which reproduce the
// correctly rounded byte/255.0f), IEEE divide and sqrt.
//
// Two exact rewrites relative to the HLSL text, both bit-identical:
//   * x / box.scale  ->  x * box.inv   (scale is a power of two, inv = 1/scale
//     is tracked alongside it; multiplying by an exact power of two rounds
//     exactly like dividing by it);
//   * (int) saturate(v * 2)  ->  v >= 0.5f   (v*2 is exact; saturate maps NaN
//     to 0, and NaN >= 0.5f is false).
//
// Reference: SdfBox/Shaders/Compute.hlsl (line numbers cited per function).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdfhip {

// One fused node record, 16 bytes: {parent, children, values[0..3], values[4..7]}.
// OctS (SdfGen/dllmain.cpp:18-29) + its 8 value bytes, so one global_load_dwordx4
// brings topology and corner distances together.
typedef uint4 NodeRec;

// Kernel parameters: the Info block unpacked (Logic.cs:407-420) + frame geometry.
struct RenderParams {
    const NodeRec *nodes;
    uint32_t n_nodes;
    float4 *out;               // compact rows: nrows_out x width
    uint32_t width, height;    // full frame
    uint32_t band_rows, band_first, band_stride, nrows_out;
    uint32_t tiles_x, tiles_y, n_tiles;   // 8x8 (compact) or 16x16 (plain) tiles of the local rows
    float h0x, h0y, h0z, h1x, h1y, h1z, h2x, h2y, h2z;  // heading rows
    float posx, posy, posz, margin;
    float screen_w, screen_h, limit;
    float lightx, lighty, lightz;
    float fov, k_strength;     // k_strength = exp2f(strength) - 1, evaluated on the host
    unsigned long long *counters;  // [0] nodes [1] samples [2] steps (COUNT builds)
    uint32_t *queue;           // tile queue head (compact kernels)
};

__device__ __forceinline__ float sat(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f); }
__device__ __forceinline__ float lerp(float a, float b, float t) { return a + t * (b - a); }
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return (ax * bx + ay * by) + az * bz;
}

// R8_UNorm decode, bit-identical to (float)b / 255.0f for b = 0..255 (checked
// exhaustively by tests/test_gpu_parity.py::test_unorm_table): one Newton
// fix-up of b * fl(1/255).
__device__ __forceinline__ float unorm8(float b)
{
    const float r = 0x1.010102p-8f;  // fl(1/255)
    float q = b * r;
    float e = __builtin_fmaf(-255.0f, q, b);
    return __builtin_fmaf(e, r, q);
}

struct Texels { float v[8]; };
__device__ __forceinline__ Texels decode(uint32_t v0, uint32_t v1)
{
    Texels t;
    t.v[0] = unorm8((float)(v0 & 0xFFu));
    t.v[1] = unorm8((float)((v0 >> 8) & 0xFFu));
    t.v[2] = unorm8((float)((v0 >> 16) & 0xFFu));
    t.v[3] = unorm8((float)(v0 >> 24));
    t.v[4] = unorm8((float)(v1 & 0xFFu));
    t.v[5] = unorm8((float)((v1 >> 8) & 0xFFu));
    t.v[6] = unorm8((float)((v1 >> 16) & 0xFFu));
    t.v[7] = unorm8((float)(v1 >> 24));
    return t;
}
__device__ __forceinline__ float bilerp(float t00, float t10, float t01, float t11, float wx, float wy)
{
    float top = lerp(t00, t10, wx);
    float bot = lerp(t01, t11, wx);
    return lerp(top, bot, wy);
}

// The per-pixel cursor: the shader's static `index` + `box` (Compute.hlsl:12,61)
// plus the record of the node it sits on, kept in registers.
struct Cursor {
    float lx, ly, lz, scale, inv;  // Cube (Compute.hlsl:31-58); inv == 1/scale exactly
    int32_t children;              // Oct.children of the current node
    uint32_t v0, v1;               // its 8 value bytes
    // generic traversal only
    uint32_t index;
    int32_t parent;
    // stack traversal only
    int32_t level;
};

// Cube::inside, Compute.hlsl:50-53
__device__ __forceinline__ bool inside(const Cursor &c, float px, float py, float pz)
{
    float hx = c.lx + c.scale, hy = c.ly + c.scale, hz = c.lz + c.scale;
    return (c.lx <= px && c.ly <= py && c.lz <= pz) && (px <= hx && py <= hy && pz <= hz);
}
// Cube::scale_up, Compute.hlsl:36-40
__device__ __forceinline__ void scale_up(Cursor &c)
{
    c.scale *= 2.0f;
    c.inv *= 0.5f;
    c.lx = floorf(c.lx * c.inv) * c.scale;
    c.ly = floorf(c.ly * c.inv) * c.scale;
    c.lz = floorf(c.lz * c.inv) * c.scale;
}
// child octant of pos in the current cell: (int3) saturate((pos - lower) / scale * 2),
// Compute.hlsl:100, then Cube::scale_down, Compute.hlsl:41-45.  Returns p = x + 2y + 4z.
__device__ __forceinline__ int descend_box(Cursor &c, float px, float py, float pz)
{
    bool bx = (px - c.lx) * c.inv >= 0.5f;
    bool by = (py - c.ly) * c.inv >= 0.5f;
    bool bz = (pz - c.lz) * c.inv >= 0.5f;
    c.scale *= 0.5f;
    c.inv *= 2.0f;
    c.lx += bx ? c.scale : 0.0f;
    c.ly += by ? c.scale : 0.0f;
    c.lz += bz ? c.scale : 0.0f;
    return (bx ? 1 : 0) + (by ? 2 : 0) + (bz ? 4 : 0);
}

__device__ __forceinline__ void set_record(Cursor &c, const NodeRec &r)
{
    c.parent = (int32_t)r.x;
    c.children = (int32_t)r.y;
    c.v0 = r.z;
    c.v1 = r.w;
}

// find, Compute.hlsl:88-108 -- generic form: follows parent and children
// links through memory exactly as the shader does (the entry read of
// data[index] is served from the registers that already hold that record).
// Returns the number of node records the *reference* reads in this call.
__device__ __forceinline__ uint32_t find_generic(Cursor &c, const NodeRec *__restrict__ nodes,
                                                 uint32_t n_nodes, float px, float py, float pz)
{
    uint32_t reads = 1;
    while (!inside(c, px, py, pz) && c.parent >= 0) {
        c.index = (uint32_t)c.parent;
        set_record(c, nodes[c.index]);
        scale_up(c);
        reads++;
    }
    int iterations = 0;
    while (c.index < n_nodes && iterations < 12 && c.children >= 0) {
        int p = descend_box(c, px, py, pz);
        c.index = (uint32_t)(c.children + p);
        set_record(c, nodes[c.index]);
        iterations++;
        reads++;
    }
    return reads;
}

// find -- cursor-stack form, for consistent trees of depth <= 12 (checked at
// upload).  Ascending needs no memory: a node's parent is the level above on
// the descent path, and the only thing the descent loop needs from an
// ancestor is its `children` field, which was pushed to an LDS stack
// (stack[level * blockDim + tid]: conflict-free for any mix of levels) when
// the path went through it.  For such trees `index < buffer_size` always
// holds and `iterations < 12` never binds (a call descends at most `depth`
// levels), so the visited cells, and hence every result, are the shader's.
__device__ __forceinline__ uint32_t find_stack(Cursor &c, const NodeRec *__restrict__ nodes,
                                               int32_t *__restrict__ stack, uint32_t stride,
                                               float px, float py, float pz)
{
    uint32_t reads = 1;
    if (!inside(c, px, py, pz) && c.level > 0) {
        do {
            c.level--;
            scale_up(c);
            reads++;
        } while (!inside(c, px, py, pz) && c.level > 0);
        c.children = stack[(uint32_t)c.level * stride];
    }
    while (c.children >= 0) {
        stack[(uint32_t)c.level * stride] = c.children;
        int p = descend_box(c, px, py, pz);
        NodeRec r = nodes[(uint32_t)(c.children + p)];
        c.children = (int32_t)r.y;
        c.v0 = r.z;
        c.v1 = r.w;
        c.level++;
        reads++;
    }
    return reads;
}

// Cube::interpol_world -> sample_at, Compute.hlsl:54-58,19-29
__device__ __forceinline__ float interpol_world(const Cursor &c, float px, float py, float pz)
{
    float dx = sat((px - c.lx) * c.inv);
    float dy = sat((py - c.ly) * c.inv);
    float dz = sat((pz - c.lz) * c.inv);
    Texels t = decode(c.v0, c.v1);
    float loadL = bilerp(t.v[0], t.v[1], t.v[2], t.v[3], dx, dy);
    float loadH = bilerp(t.v[4], t.v[5], t.v[6], t.v[7], dx, dy);
    return (lerp(loadL, loadH, dz) - 0.25f) * c.scale * 2.0f;
}

// gradient, Compute.hlsl:112-130 (taps at integer x / y have bilinear weight 0
// towards the neighbouring texel: lerp(a, b, 0) == a)
__device__ __forceinline__ void gradient(const Cursor &c, float px, float py, float pz,
                                         float &gx, float &gy, float &gz)
{
    float dx = sat((px - c.lx) * c.inv);
    float dy = sat((py - c.ly) * c.inv);
    float dz = sat((pz - c.lz) * c.inv);
    Texels t = decode(c.v0, c.v1);
    float xl = lerp(lerp(t.v[0], t.v[2], dy), lerp(t.v[4], t.v[6], dy), dz);
    float xh = lerp(lerp(t.v[1], t.v[3], dy), lerp(t.v[5], t.v[7], dy), dz);
    float yl = lerp(lerp(t.v[0], t.v[1], dx), lerp(t.v[4], t.v[5], dx), dz);
    float yh = lerp(lerp(t.v[2], t.v[3], dx), lerp(t.v[6], t.v[7], dx), dz);
    float zl = bilerp(t.v[0], t.v[1], t.v[2], t.v[3], dx, dy);
    float zh = bilerp(t.v[4], t.v[5], t.v[6], t.v[7], dx, dy);
    gx = xh - xl;
    gy = yh - yl;
    gz = zh - zl;
}

// ray, Compute.hlsl:163-168
__device__ __forceinline__ void ray(const RenderParams &P, uint32_t cx, uint32_t cy, float &dx,
                                    float &dy, float &dz)
{
    float sx = (float)cx / P.screen_h - P.screen_w / P.screen_h * 0.5f;
    float sy = (float)cy / P.screen_h - 0.5f;
    float vx = sx * P.fov, vy = sy * P.fov, vz = 0.5f;
    float d0 = dot3(vx, vy, vz, P.h0x, P.h0y, P.h0z);
    float d1 = dot3(vx, vy, vz, P.h1x, P.h1y, P.h1z);
    float d2 = dot3(vx, vy, vz, P.h2x, P.h2y, P.h2z);
    float len = sqrtf(dot3(d0, d1, d2, d0, d1, d2));
    dx = d0 / len;
    dy = d1 / len;
    dz = d2 / len;
}

}  // namespace sdfhip
