// The ray-march kernels of libsdfhip.so (gfx950): k_march (the default), k_plain, k_compact, the path-traced pipeline
// and the pieces of Compute.hlsl's main() they are built from.  Device building blocks (cursors, find,
// sampling): raymarch_device.h.  Host side (launches, C ABI): render.hip.
//
// Replaces: SdfBox/Shaders/Compute.hlsl:180-231 main() and, fused into the epilogue,
// SdfBox/Shaders/DisplayFrag.hlsl:16-24.
//
// Kernel structure:
//   * one lane per pixel, 8x8 pixels per 64-lane wavefront, one wave per workgroup; blockIdx is remapped so that
//     XCD k (own L2) renders tile rows k, k+8, ...: every XCD sees the same mix of sky and object rows (balance)
//     while whole rows of neighbouring tiles share an L2;
//   * k_march (trees behind a full-depth or split grid): the primary march, the shading step and the shadow march of
//     Compute.hlsl:194-230 as three wave-converged phases -- two tight loops with one exit each and the shading between
//     them, once per wave;
//   * k_plain (other trees; A/B knob): the three phases as ONE per-lane state machine around a single find + sample body;
//   * k_march<..., QUEUE> + k_shadow (SDFHIP_FLAG_COMPACT): the shadow rays of sparse waves compacted into a queue by ballot +
//     prefix count and marched 64 to a wave;
//   * k_compact (SDFHIP_FLAG_COMPACT on trees without a full-depth grid): persistent waves pull 8x8 tiles from per-XCD queues
//     and refill finished lanes by ballot + prefix count, with refill and shading batched;
//   * k_pt_primary / k_pt_bounce / k_pt_resolve (and the one-kernel k_path): the path-traced mode of BASELINE config 5
//     (defined by the oracle).
#pragma once
#include "raymarch_device.h"

namespace sdfhip {

constexpr int MAX_STACK = LM;          // the shader's own descent limit (Compute.hlsl:98)
#ifndef COMPACT_WAVES_PER_SIMD
#define COMPACT_WAVES_PER_SIMD 8
#endif
#ifndef PATH_WAVES_PER_SIMD
#define PATH_WAVES_PER_SIMD 7
#endif
#ifndef PLAIN_WAVES_PER_SIMD
#define PLAIN_WAVES_PER_SIMD 8          // <= 64 VGPRs: 8 waves per SIMD (2nd launch-bound = waves per SIMD)
#endif

// cursor kinds of the kernels: generic, cursor stack, cursor stack with a top grid as deep as the tree (dense, or split)
// (experiments build: CUR_DENSE4, the default kernel through the grid's second form, CursorFF in lab_device.h)
enum { CUR_GENERIC = 0, CUR_STACK = 1, CUR_STACK_FULL = 2, CUR_STACK_SPLIT = 3, CUR_DENSE4 = 4 };
template <int CUR, bool COUNT> struct CursorOf { typedef CursorG type; };
template <bool COUNT> struct CursorOf<CUR_STACK, COUNT> { typedef CursorS type; };
template <bool COUNT> struct CursorOf<CUR_STACK_FULL, COUNT> { typedef CursorFT<COUNT, false> type; };
template <bool COUNT> struct CursorOf<CUR_STACK_SPLIT, COUNT> { typedef CursorFT<COUNT, true> type; };
// the grid a launch's find() reads
__device__ __forceinline__ GridRef grid_of(const RenderParams &P, const TopCell *top = nullptr)
{
#ifdef SDFHIP_EXPERIMENTS
    return GridRef{top ? top : P.top, P.fine, P.top_level, P.fine_bits, P.fine_order, P.d4, P.recs,
                   P.touch_top, P.touch_fine, P.touch_top_words, P.touch_fine_words};
#else
    return GridRef{top ? top : P.top, P.fine, P.top_level, P.fine_bits, P.fine_order};
#endif
}
// the cell a cursor sits in, for the gradient: the cursors carry its bytes (CursorFF looks them up again)
template <class CursorT>
__device__ __forceinline__ Cell cell_of(const CursorT &c, const RenderParams &) { return c.cell(); }
#ifdef SDFHIP_EXPERIMENTS
template <> struct CursorOf<CUR_DENSE4, false> { typedef CursorFF type; };
__device__ __forceinline__ Cell cell_of(const CursorFF &c, const RenderParams &P) { return cell_of(c, grid_of(P), P.nodes); }
#endif
// the bounce kernels of the path tracer may read a split grid whose blocks are stored sub-cube by sub-cube (GridRef::fine_order)
template <int CUR, bool COUNT> struct ScatterCursorOf { typedef typename CursorOf<CUR, COUNT>::type type; };
template <bool COUNT> struct ScatterCursorOf<CUR_STACK_SPLIT, COUNT> { typedef CursorFT<COUNT, true, true> type; };

// lane states: marching (primary / shadow), march over and shading pending, no pixel
enum { PH_PRIMARY = 0, PH_SHADOW = 1, PH_SHADE = 2, PH_IDLE = 3, PH_DONE = 4 };   // DONE: idle, colour waiting in LDS

struct RayState {
    float px, py, pz;    // pos
    float dx, dy, dz;    // dir (primary direction, then direction to the light)
    float prox, angle, dist;
    int n, base, phase;  // steps of the current march, steps before it (i = base or n, j = n), PH_*
};

__device__ __forceinline__ uint32_t global_row(const RenderParams &P, uint32_t yl)
{
    // (every band of the frame, in order -- a whole frame, whatever its band height: local rows are frame rows, and no division)
    if (!P.n_band_list && P.band_first == 0u && P.band_stride == 1u) return yl;
    const uint32_t band = P.band_shift < 32u ? yl >> P.band_shift : yl / P.band_rows, within = yl - band * P.band_rows;
    if (P.n_band_list) {
        if (band >= P.n_band_list) return 0xFFFFFFFFu;
        const uint32_t b = P.n_band_list <= (uint32_t)INLINE_BAND_LIST ? P.band_list[band] : P.band_ptr[band];
        return b * P.band_rows + within;
    }
    return (P.band_first + band * P.band_stride) * P.band_rows + within;
}

template <class CursorT>
__device__ __forceinline__ void start_pixel(const FrameInfo &I, const NodeRec &root, uint32_t x,
                                            uint32_t y, RayState &r, CursorT &c)
{
    // Compute.hlsl:182-191
    r.px = I.posx; r.py = I.posy; r.pz = I.posz;
    ray(I, x, y, r.dx, r.dy, r.dz);
    r.prox = 1.0f;
    r.angle = 0.0f; r.dist = 0.0f;
    r.n = 0; r.base = 0; r.phase = PH_PRIMARY;
    c.reset(root);
}

// ---- main() between two find() calls, in three pieces -------------------------
// A finished pixel's colour (alpha = step count = r.base + r.n) is stored to *dst by the
// piece that finishes it.  (Storing there instead of returning the colour keeps four
// values out of the march loop's phi nodes.)

// ---- fused display pass: SdfBox/Shaders/DisplayFrag.hlsl:16-24 --------------------------
// float -> R8G8B8A8_UNorm as D3D11 converts render-target output: NaN -> 0, clamp to
// [0, 1], scale by 255, round to nearest.
__device__ __forceinline__ uint32_t to_unorm8(float c) { return (uint32_t)(sat(c) * 255.0f + 0.5f); }
// `return pow(val, 1 / 2.2)` on one channel
__device__ __forceinline__ uint32_t gamma8(float c) { return to_unorm8(powf(c, 1.0f / 2.2f)); }
// alpha = step count: pow(n, 1/2.2) >= 1 for n >= 1, and pow(0, .) = 0
__device__ __forceinline__ uint32_t alpha8(float steps) { return steps >= 1.0f ? 0xFF000000u : 0u; }
// `return float4(1, 1, 1, 0) * val.w / 140` (debug heat map)
__device__ __forceinline__ uint32_t heat8(float steps)
{
    uint32_t q = to_unorm8(steps / 140.0f);
    return q | (q << 8) | (q << 16);
}
// any colour (used where colours come back from LDS in the compact kernel)
__device__ __forceinline__ uint32_t display8(const float4 &v, uint32_t mode)
{
    if (mode == 2u) return heat8(v.w);
    return gamma8(v.x) | (gamma8(v.y) << 8) | (gamma8(v.z) << 16) | alpha8(v.w);
}

// Where a finished pixel's colour goes: straight to the frame (plain kernel) -- as
// RGBA32F, or through the display pass as RGBA8 -- or to the lane's LDS slot, to be
// flushed at the next refill (compact kernel: on gfx950 a store counts on vmcnt like a
// load, so a global store per finished pixel would stall the very next node load of
// the whole wave behind the store's completion).  Every finished pixel is the sky
// constant, black, or a grey level, which keeps the display pass to one pow.
// mode 3 is the wire format of the tile gather, lossless in 5 bytes per pixel: every pixel main()
// writes is (a, a, a, n) with n <= 140 steps, or the sky constant (0.005, 0.01, 0.2, n) with n <= 100.
// A frame's wire buffer is two planes, [rows * width] floats (the bits of a) and then [rows * width]
// bytes (n, or 255 - n for a sky pixel); wire_expand undoes it.
__device__ __forceinline__ float4 wire_expand(float a, uint32_t code)
{
    if (code > 140u) return make_float4(0.005f, 0.01f, 0.2f, (float)(255u - code));
    return make_float4(a, a, a, (float)code);
}
struct FrameSink {
    float4 *p;            // RGBA32F pixel, or (as uint32_t *) the RGBA8 pixel, or (as float *) the wire pixel's a
    uint32_t mode, sky8;
    const char *wire_base;   // mode 3 (wave-uniform): the frame's wire buffer, and its byte plane
    uint8_t *wire_codes;
    __device__ __forceinline__ void wire(float a, uint32_t code) const
    {
        *reinterpret_cast<float *>(p) = a;
        wire_codes[(reinterpret_cast<const char *>(p) - wire_base) >> 2] = (uint8_t)code;
    }
    __device__ __forceinline__ void sky(float steps) const
    {
        if (mode == 0u) *p = make_float4(0.005f, 0.01f, 0.2f, steps);
        else if (mode == 3u) wire(0.0f, 255u - (uint32_t)steps);
        else *reinterpret_cast<uint32_t *>(p) = mode == 2u ? heat8(steps) : (sky8 | alpha8(steps));
    }
    __device__ __forceinline__ void grey(float a, float steps) const
    {
        if (mode == 0u) *p = make_float4(a, a, a, steps);
        else if (mode == 3u) wire(a, (uint32_t)steps);
        else {
            uint32_t q = gamma8(a);
            *reinterpret_cast<uint32_t *>(p) = mode == 2u ? heat8(steps) : (q | (q << 8) | (q << 16) | alpha8(steps));
        }
    }
    __device__ __forceinline__ void black(float steps) const
    {
        if (mode == 0u) *p = make_float4(0.0f, 0.0f, 0.0f, steps);
        else if (mode == 3u) wire(0.0f, (uint32_t)steps);
        else *reinterpret_cast<uint32_t *>(p) = mode == 2u ? heat8(steps) : alpha8(steps);
    }
};
struct LdsSink {
    float4 *slot;      // points into a __shared__ array (address space known after inlining)
    __device__ __forceinline__ void sky(float steps) const { *slot = make_float4(0.005f, 0.01f, 0.2f, steps); }
    __device__ __forceinline__ void grey(float a, float steps) const { *slot = make_float4(a, a, a, steps); }
    __device__ __forceinline__ void black(float steps) const { *slot = make_float4(0.0f, 0.0f, 0.0f, steps); }
};

// Loop header + escape test of the primary march, Compute.hlsl:194-199.
// 0: take a march step; 1: the march is over, shade next; 2: pixel finished (sky).
template <class Sink>
__device__ __forceinline__ int check_primary(const FrameInfo &I, const RayState &r, const Sink &dst)
{
    if ((r.prox > I.margin2 || r.prox < 0.0f) && r.n < 100) {
        if (dot3(r.px, r.py, r.pz, r.px, r.py, r.pz) > I.limit) {
            dst.sky((float)r.n);
            return 2;
        }
        return 0;
    }
    return 1;
}

// Compute.hlsl:205-213: turn towards the light, Lambert term from the gradient.
// true: pixel finished (faces away).  Otherwise the lane enters the shadow march.
template <class CursorT, class Sink>
__device__ __forceinline__ bool shade(const FrameInfo &I, RayState &r, const CursorT &c, const Sink &dst)
{
    float lx = I.lightx - r.px, ly = I.lighty - r.py, lz = I.lightz - r.pz;
    float rl = 1.0f / sqrtf(dot3(lx, ly, lz, lx, ly, lz));
    r.dx = lx * rl; r.dy = ly * rl; r.dz = lz * rl;
    r.px = __builtin_fmaf(r.dx, I.margin, r.px);
    r.py = __builtin_fmaf(r.dy, I.margin, r.py);
    r.pz = __builtin_fmaf(r.dz, I.margin, r.pz);
    float gx, gy, gz;
    gradient(c.cell(), r.px, r.py, r.pz, gx, gy, gz);
    float rg = 1.0f / sqrtf(dot3(gx, gy, gz, gx, gy, gz));
    r.angle = dot3(r.dx, r.dy, r.dz, gx * rg, gy * rg, gz * rg);
    if (r.angle < 0.0f) {
        dst.black((float)r.n);
        return true;
    }
    lx = I.lightx - r.px; ly = I.lighty - r.py; lz = I.lightz - r.pz;
    r.dist = sqrtf(dot3(lx, ly, lz, lx, ly, lz)) / 2.0f;
    r.phase = PH_SHADOW;
    r.base = r.n;                 // i stays, j starts
    r.n = 0;
    return false;
}

// Loop header and the three exits of the shadow march, Compute.hlsl:214-223,229.
// true: pixel finished.
template <class CursorT, class Sink>
__device__ __forceinline__ bool check_shadow(const FrameInfo &I, const RayState &r, const CursorT &c, const Sink &dst)
{
    if (!(r.n < 40 && r.prox > -I.margin)) {
        dst.black((float)(r.base + r.n));  // :229
        return true;
    }
    if (r.prox > r.dist || (r.px < 0.0f || r.py < 0.0f || r.pz < 0.0f) ||
        (r.px > 1.0f || r.py > 1.0f || r.pz > 1.0f)) {           // :215-219
        float a = r.angle / (r.dist * r.dist) * I.k_strength;
        dst.grey(a, (float)(r.base + r.n));
        return true;
    }
    if (r.prox < I.margin) {                                       // :221-223
        float gx, gy, gz;
        gradient(c.cell(), r.px, r.py, r.pz, gx, gy, gz);
        if (dot3(gx, gy, gz, r.dx, r.dy, r.dz) < 0.0f) {
            dst.black((float)(r.base + r.n));
            return true;
        }
    }
    return false;
}

// The three pieces in program order: what one lane does between two march steps.
template <class CursorT, class Sink>
__device__ __forceinline__ bool pre_step(const FrameInfo &I, RayState &r, const CursorT &c, const Sink &dst)
{
    if (r.phase == PH_PRIMARY) {
        int s = check_primary(I, r, dst);
        if (s == 0) return false;
        if (s == 2) return true;
        if (shade(I, r, c, dst)) return true;
    }
    return check_shadow(I, r, c, dst);
}

// find + interpol_world + advance: Compute.hlsl:200-202 / :225-227
// FRESH: the cursor comes straight from reset() (the first step of a pixel), see find_fresh
template <class CursorT, bool FRESH = false>
__device__ __forceinline__ uint32_t march_step(const RenderParams &P, const FrameInfo &I, RayState &r, CursorT &c,
                                               int32_t *stack, uint32_t stride, const TopCell *top = nullptr)
{
    typename CursorT::Pos u;
    // top: the top grid somewhere else than P.top (the workgroup's LDS copy, k_plain<..., LDSTOP>)
    const GridRef g = grid_of(P, top);
    uint32_t reads = FRESH ? find_fresh(c, P.nodes, g, P.n_nodes, stack, stride, r.px, r.py, r.pz, u)
                           : find(c, P.nodes, g, P.n_nodes, stack, stride, r.px, r.py, r.pz, u);
    r.prox = sample_after_find(c, u, r.px, r.py, r.pz);
    float step = r.phase ? r.prox + I.margin : r.prox;
    r.px = __builtin_fmaf(r.dx, step, r.px);
    r.py = __builtin_fmaf(r.dy, step, r.py);
    r.pz = __builtin_fmaf(r.dz, step, r.pz);
    r.n++;
    return reads;
}

// Diagnostics of the counting builds of k_march (sdfhip_debug_step_classes): what kind of cell every lane-step sampled.
// [0] flat leaf of the coarse level or above  [1] flat leaf inside a fine block (or, dense grid: below level 8)
// [2] non-flat, coarse  [3] non-flat, as deep as the grid (k = 0)  [4] non-flat below the coarse level but not at the
// grid's full depth  [5] non-flat steps whose position is outside the cube or NaN
// (experiments build only; the product's counting kernels carry empty stand-ins)
#ifndef SDFHIP_EXPERIMENTS
struct StepClasses {};
template <class CursorT>
__device__ __forceinline__ void classify_step(StepClasses &, const CursorT &, const RenderParams &, float, float, float) {}
__device__ __forceinline__ void flush_classes(const RenderParams &, StepClasses &) {}
#else
struct StepClasses { unsigned long long n[6] = {0, 0, 0, 0, 0, 0}; };
template <class CursorT>
__device__ __forceinline__ void classify_step(StepClasses &, const CursorT &, const RenderParams &, float, float, float) {}
template <bool SPLIT, bool ORDERED>
__device__ __forceinline__ void classify_step(StepClasses &k, const CursorFT<true, SPLIT, ORDERED> &c, const RenderParams &P, float px, float py, float pz)
{
    const int level = LM - (int)(c.s & 15u), F = P.top_level + (SPLIT ? P.fine_bits : 0), C = SPLIT ? P.top_level : (P.top_level < 8 ? P.top_level : 8);
    const bool flat = (c.s & FLAT_BIT) != 0, coarse = level <= C;
    const bool inside = px >= 0.0f && py >= 0.0f && pz >= 0.0f && px < 1.0f && py < 1.0f && pz < 1.0f;
    k.n[0] += (flat && coarse) ? 1u : 0u; k.n[1] += (flat && !coarse) ? 1u : 0u;
    k.n[2] += (!flat && coarse) ? 1u : 0u; k.n[3] += (!flat && !coarse && level == F) ? 1u : 0u;
    k.n[4] += (!flat && !coarse && level != F) ? 1u : 0u; k.n[5] += (!flat && !inside) ? 1u : 0u;
}
__device__ __forceinline__ void flush_classes(const RenderParams &P, StepClasses &k)
{
    for (int i = 0; i < 6; i++) {
        unsigned long long v = k.n[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&P.counters[6 + i], v);
    }
}
#endif

// loads: 16-byte node records / grid cells the kernel itself loaded (the cursor counts them); hits: shadow rays queued for k_shadow
__device__ __forceinline__ void flush_counters(const RenderParams &P, unsigned long long nodes,
                                               unsigned long long samples, unsigned long long steps,
                                               unsigned long long shadow_rays, unsigned long long loads,
                                               unsigned long long hits = 0)
{
    for (int off = 32; off > 0; off >>= 1) {
        nodes += __shfl_down(nodes, off);
        samples += __shfl_down(samples, off);
        steps += __shfl_down(steps, off);
        shadow_rays += __shfl_down(shadow_rays, off);
        loads += __shfl_down(loads, off);
        hits += __shfl_down(hits, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&P.counters[0], nodes);
        atomicAdd(&P.counters[1], samples);
        atomicAdd(&P.counters[2], steps);
        atomicAdd(&P.counters[3], shadow_rays);
        atomicAdd(&P.counters[4], loads);
        if (hits) atomicAdd(&P.counters[5], hits);
    }
}

// ---- one lane per pixel; a workgroup of BT threads renders a 16 x (BT/16) tile (BT >= 128)
// or one 8x8 wave tile (BT = 64) ------------------------------------------------
// LDSTOP (measurement variant, cursor-stack kernels with a top grid of level <= 3): every workgroup first copies the
// top grid -- the hot inner nodes: 8^3 cells x 16 B = 8 KB -- into LDS and find() reads it from there
// (north_star's "LDS caching of the hot inner nodes per workgroup"; DESIGN.md section 4.2 has the timing)
template <int CUR, bool COUNT, int BT, bool LDSTOP = false>
__global__ __launch_bounds__(BT, PLAIN_WAVES_PER_SIMD) void k_plain(RenderParams P)
{
    FrameInfo I = P.frames[blockIdx.y];              // batched launch: one frame per grid.y
    __shared__ TopCell top_lds[LDSTOP ? 512 : 1];
    // the three scalars every march step reads stay in SGPRs: left alone, the compiler reloads them
    // from the kernel arguments (s_load + s_waitcnt) in every iteration
    asm volatile("" : "+s"(I.margin), "+s"(I.margin2), "+s"(I.limit));
    __shared__ int32_t stack_lds[CUR ? MAX_STACK * BT : 1];
    constexpr uint32_t TW = BT >= 128 ? 16 : 8, TH = BT / 8 / (TW / 8);   // tile = TW x TH pixels
    // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2), so give
    // each of the 8 residue classes one contiguous run of tiles (bijective for
    // any grid size).
    const uint32_t nb = gridDim.x, bid = blockIdx.x;
    uint32_t tile;
#ifdef SDFHIP_EXPERIMENTS
    if (P.tile_order == 2) {          // one contiguous slab of tiles per XCD (load-imbalanced: kept for A/B runs)
        const uint32_t q = nb >> 3, rem = nb & 7u, xcd = bid & 7u;
        tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    } else if (P.tile_order == 1) {   // dispatch order = row-major tile order
        tile = bid;
    } else
#else
    (void)nb;
#endif
    {                                 // default: XCD k renders tile rows k, k+8, ... (grid padded to 8*ceil(tiles_y/8) rows)
        // tiles_y is rarely a multiple of 8: the first tiles_y % 8 labels get one tile row more.  A rank's
        // share of a sharded frame has few tile rows (17 at 8 ranks: 3 for label 0, 2 for the others), so
        // in a batched launch frame f shifts the labels by f * (tiles_y % 8): the extra rows go to
        // different XCDs frame after frame
        const uint32_t xcd = (bid + blockIdx.y * (P.tiles_y & 7u)) & 7u, j = bid >> 3;   // j-th block of this label
        const uint32_t r = j / P.tiles_x, cx = j - r * P.tiles_x;
        const uint32_t row = r * 8 + xcd;
        tile = row < P.tiles_y ? row * P.tiles_x + cx : 0xFFFFFFFFu;
    }
    if (tile >= P.n_tiles) return;
    const uint32_t tx = tile % P.tiles_x, ty = tile / P.tiles_x;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    constexpr uint32_t WX = TW / 8;                        // waves side by side in a tile
    const uint32_t x = tx * TW + (wave % WX) * 8 + (lane & 7u);
    const uint32_t yl = ty * TH + (wave / WX) * 8 + (lane >> 3);
    unsigned long long cn = 0, cs = 0, ct = 0, cr = 0, cl = 0;   // nodes, samples, steps, shadow rays, loads
    if (LDSTOP) {
        const uint32_t ncell = 1u << (3 * P.top_level);        // <= 512 (checked by the host)
        for (uint32_t i = tid; i < ncell; i += BT) top_lds[i] = P.top[i];
        __syncthreads();
    }
    bool live = x < P.width && yl < P.nrows_out;
    uint32_t y = 0;
    if (live) { y = global_row(P, yl); live = y < P.height; }
    if (live) {
        RayState r;
        typename CursorOf<CUR, COUNT>::type c;
        c.loads = 0;
        const NodeRec root = P.nodes[0];
        start_pixel(I, root, x, y, r, c);
        const size_t npx = (size_t)P.nrows_out * P.width, lidx = (size_t)yl * P.width + x;
        const size_t pidx = (size_t)blockIdx.y * npx + lidx;
        char *const wire_base = reinterpret_cast<char *>(P.out) + (size_t)blockIdx.y * npx * 5;   // mode 3: this frame's planes
        const FrameSink dst{P.out_mode == 0u ? P.out + pidx :
                            P.out_mode == 3u ? reinterpret_cast<float4 *>(reinterpret_cast<float *>(wire_base) + lidx) :
                                               reinterpret_cast<float4 *>(reinterpret_cast<uint32_t *>(P.out) + pidx),
                            P.out_mode, P.sky8, wire_base, reinterpret_cast<uint8_t *>(wire_base) + 4 * npx};
        while (!pre_step(I, r, c, dst)) {
            uint32_t reads = march_step(P, I, r, c, stack_lds + tid, BT, LDSTOP ? top_lds : nullptr);
            if (COUNT) { cn += reads; cs += 1; }
        }
        if (COUNT) { ct = (unsigned long long)(r.base + r.n); cr = r.phase == PH_SHADOW ? 1u : 0u; cl = c.loads; }   // shade() left the lane in PH_SHADOW
    }
    if (COUNT) flush_counters(P, cn, cs, ct, cr, cl);
}

// =====================================================================================
// The default kernel for trees behind a full-depth or split grid: k_march.
//
//   One lane per pixel, one 8x8 tile per wave, the plain kernel's XCD-interleaved tile rows.  Three phases, the wave
//   converged between them:
//     1. the primary march of Compute.hlsl:194-203 -- a loop of loop-header test, escape test, find, sample, advance with no
//        phase logic, no shading code and no stores in it; one exit, the reason read off the lane's registers afterwards;
//     2. sky pixels store their colour (coalesced); the lanes that ended on the surface take the shading step of
//        Compute.hlsl:205-213 together; a pixel that faces away from the light is finished there, black;
//     3. the others march their shadow rays (Compute.hlsl:214-230), again a tight loop with one exit, and store their colour.
//
// Why: in the one-kernel form of round 1 (k_plain) the ~200 instructions of the shading step run whenever ANY
// lane of a wave finishes its primary march -- for three or four lanes at a time, about 130 000 times
// per 1080p frame, a quarter of all VALU instructions issued -- and every iteration pays the phase
// dispatch of a lane state machine.  Here shading runs once per wave, and both loops carry only their own exits.
// Per-pixel arithmetic, its order and the cursor a pixel carries from the primary into the shadow march are unchanged:
// images and counters stay bit-identical.
//
// QUEUE = true (SDFHIP_FLAG_COMPACT, BASELINE cfg-3's "wavefront ray compaction"): after the shading step a wave counts its shadow
// rays (one ballot).  A wave that holds at least P.hit_min of them marches them in place, as above.  A sparser wave -- a silhouette,
// a shadow edge: a handful of live lanes that would keep 64 marching -- appends its rays (position, prox, step count, cursor, light
// direction: 64 bytes) to a queue, slots by ballot + mbcnt prefix and one atomic per wave, and ends; a second kernel, k_shadow,
// marches the queued rays 64 consecutive records to a wave.  Measured (scripts/shadow_hybrid_ab.sh): with every ray queued
// (hit_min = 65, round 2's A/B form) 0.095 ms per 1080p frame against 0.087 and 0.341 against 0.311 at 4K -- the coherent primary
// rays of these frames leave most waves either full of shadow rays or empty, and the queue's 128 bytes per ray and the second
// launch cost more than the denser waves return; with hit_min = 32 0.0894 and 0.3168: within 2-3 % of no compaction at all,
// against 0.285 / 0.917 for the persistent-wave lane refill (k_compact) that carried the flag until round 4.
// =====================================================================================
enum { OUT_RGBA32F = 0, OUT_GAMMA8 = 1, OUT_HEAT8 = 2, OUT_WIRE = 3, OUT_SPARSE = 4 };

// OUT_SPARSE: what a rank of a sharded frame puts on xGMI, written by the march kernel itself.  A wave of k_march IS one
// 8x8 tile, so at its end it knows the tile's 64 wire pixels (a, code) in registers: the mask of pixels whose a has any bit
// set is one ballot, the tile's slots in the packed float array one atomic add of its popcount (the order of the tiles in
// that array is whatever order the waves finish in; each tile records where its floats start), and the code bytes are one
// 64-byte store, in tile order.  No dense wire buffer, no mask / scan / scatter kernels behind the render (sdfhip_wire_compact_device:
// 265 us per group of four 4K frames), and rank 0 expands the shares exactly as before.  One share holds the `frames` frames of a
// launch: header (word 0: float slots handed out SINCE THE BUFFER WAS ZEROED -- a launch counts on from where the last one stopped and is
// told that value; slots beyond the capacity are dropped and the count says so) | masks
// [frames][tiles] | slot bases [frames][tiles] | codes [frames][tiles][64] | floats [capacity], so that a gather is two
// contiguous copies: the fixed part, and as many floats as were used.
struct Sparse2Layout {
    uint32_t width, rows, tiles_x, tiles_y, tiles, frames, capacity;
    size_t off_masks, off_bases, off_codes, off_floats, bytes;
};
__host__ __device__ inline Sparse2Layout sparse2_layout(uint32_t width, uint32_t rows, uint32_t frames, uint32_t capacity)
{
    Sparse2Layout L;
    L.width = width; L.rows = rows; L.frames = frames; L.capacity = capacity;
    L.tiles_x = (width + 7) / 8; L.tiles_y = (rows + 7) / 8; L.tiles = L.tiles_x * L.tiles_y;
    const size_t ft = (size_t)frames * L.tiles;
    L.off_masks = 64;
    L.off_bases = L.off_masks + ft * 8;
    L.off_codes = (L.off_bases + ft * 4 + 63) & ~(size_t)63;
    L.off_floats = L.off_codes + ft * 64;
    L.bytes = (L.off_floats + (size_t)capacity * 4 + 63) & ~(size_t)63;
    return L;
}
struct SparseLane { float a; uint32_t code; };     // a lane's wire pixel, until the wave's flush
__device__ __forceinline__ void sparse2_flush(const RenderParams &P, uint32_t f, uint32_t tile, uint32_t lane, const SparseLane &px)
{
    const Sparse2Layout L = sparse2_layout(P.width, P.nrows_out, P.n_frames, P.sparse_cap);
    char *share = reinterpret_cast<char *>(P.out);
    const size_t ft = (size_t)f * L.tiles + tile;
    const bool lit = __float_as_uint(px.a) != 0u;
    const unsigned long long m = __ballot(lit);
    uint32_t base = 0;
    if (m) {
        // (the share's counter is never zeroed: it counts on from P.sparse_base, its value before this launch -- no memset in front of the march)
        if (lane == 0) base = atomicAdd(reinterpret_cast<uint32_t *>(share), (uint32_t)__popcll(m)) - P.sparse_base;
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    }
    reinterpret_cast<uint8_t *>(share + L.off_codes)[ft * 64 + lane] = (uint8_t)px.code;
    if (lane == 0) {
        reinterpret_cast<unsigned long long *>(share + L.off_masks)[ft] = m;
        reinterpret_cast<uint32_t *>(share + L.off_bases)[ft] = base;
    }
    const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (lit && slot < P.sparse_cap) reinterpret_cast<float *>(share + L.off_floats)[slot] = px.a;
}

// Where a finished pixel goes, by output mode (a template parameter: no dispatch at the store, and no
// display-pass code in the RGBA32F kernels).  idx = pixel index within the launch's frame f.
// A frame's pixels are written once and not read again by the kernel that writes them: k_march / k_shadow store them past the
// caches (round 6).  As plain stores the 33 MB (1080p) / 133 MB (4K) of a frame went through the XCDs' 4 MB L2s and took their lines
// from the grid cells the rays look up: with the stores non-temporal the 4K frame takes 0.292 instead of 0.304 ms (-4 %), the
// depth-10 stand-in 0.101-0.103 instead of 0.105-0.109, the mesh scene 0.055-0.057 instead of 0.058-0.063, 1080p 0.0847 instead of
// 0.0854 (A/B/A/B of two builds of the library in one run; bit-identical frames: the same values, another cache policy).
// NOT into page-locked HOST memory (sdfhip_render into a registered array: the march stores across PCIe): there the non-temporal
// form is slower -- 1080p RGBA32F 0.676 -> 0.701 ms, the RGBA8 display frame 0.256 -> 0.366 (scripts/host_frame.py --locked, the same
// A/B) -- so that launch takes an instantiation of its own (output mode | OUT_HOST) with plain stores.  The policy is a TEMPLATE
// parameter on purpose: a run-time "if (host) plain else non-temporal" is merged by the compiler into one store (both forms store
// the same value to the same address) and the hint is lost -- measured: the 4 % were gone; a volatile host store keeps them apart
// but is emitted as a system-scope write-through store that crosses PCIe at half the rate for the 4-byte display pixels (0.512 ms).
constexpr int OUT_HOST = 8;
template <bool HOST>
__device__ __forceinline__ void frame_store(float4 *p, const float4 &v)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    if (HOST) *p = v;
    else __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4 *>(p));
}
template <bool HOST>
__device__ __forceinline__ void frame_store(uint32_t *p, uint32_t v)
{
    if (HOST) *p = v;
    else __builtin_nontemporal_store(v, p);
}
template <int MODE_>
struct PixelSink {
    static constexpr int MODE = MODE_ & (OUT_HOST - 1);       // the output mode proper
    static constexpr bool HOST = (MODE_ & OUT_HOST) != 0;      // the frame lies in page-locked host memory: plain stores (frame_store)
    float4 *out;             // frame f of the launch: RGBA32F pixels / uint32 pixels / the frame's wire planes
    uint8_t *codes;          // OUT_WIRE: the byte plane behind the float plane
    uint32_t sky8;
    SparseLane *px;          // OUT_SPARSE: the lane's wire pixel stays in registers until the wave's flush (sparse2_flush)
    __device__ __forceinline__ PixelSink(const RenderParams &P, uint32_t f, SparseLane *lane_px = nullptr)
    {
        const size_t npx = (size_t)P.nrows_out * P.width;
        sky8 = P.sky8;
        px = lane_px;
        if (MODE == OUT_SPARSE) { out = nullptr; codes = nullptr; }
        else if (MODE == OUT_RGBA32F) { out = P.out + f * npx; codes = nullptr; }
        else if (MODE == OUT_WIRE) {
            char *base = reinterpret_cast<char *>(P.out) + f * npx * 5;
            out = reinterpret_cast<float4 *>(base); codes = reinterpret_cast<uint8_t *>(base) + 4 * npx;
        } else { out = reinterpret_cast<float4 *>(reinterpret_cast<uint32_t *>(P.out) + f * npx); codes = nullptr; }
    }
    __device__ __forceinline__ void wire(size_t idx, float a, uint32_t code) const
    {
        if (MODE == OUT_SPARSE) { px->a = a; px->code = code; return; }
        reinterpret_cast<float *>(out)[idx] = a;
        codes[idx] = (uint8_t)code;
    }
    __device__ __forceinline__ void sky(size_t idx, float steps) const
    {
        if (MODE == OUT_RGBA32F) frame_store<HOST>(&out[idx], make_float4(0.005f, 0.01f, 0.2f, steps));
        else if (MODE == OUT_WIRE || MODE == OUT_SPARSE) wire(idx, 0.0f, 255u - (uint32_t)steps);
        else frame_store<HOST>(&reinterpret_cast<uint32_t *>(out)[idx], MODE == OUT_HEAT8 ? heat8(steps) : (sky8 | alpha8(steps)));
    }
    __device__ __forceinline__ void grey(size_t idx, float a, float steps) const
    {
        if (MODE == OUT_RGBA32F) frame_store<HOST>(&out[idx], make_float4(a, a, a, steps));
        else if (MODE == OUT_WIRE || MODE == OUT_SPARSE) wire(idx, a, (uint32_t)steps);
        else if (MODE == OUT_HEAT8) frame_store<HOST>(&reinterpret_cast<uint32_t *>(out)[idx], heat8(steps));
        else { const uint32_t q = gamma8(a); frame_store<HOST>(&reinterpret_cast<uint32_t *>(out)[idx], q | (q << 8) | (q << 16) | alpha8(steps)); }
    }
    __device__ __forceinline__ void black(size_t idx, float steps) const
    {
        if (MODE == OUT_RGBA32F) frame_store<HOST>(&out[idx], make_float4(0.0f, 0.0f, 0.0f, steps));
        else if (MODE == OUT_WIRE || MODE == OUT_SPARSE) wire(idx, 0.0f, (uint32_t)steps);
        else frame_store<HOST>(&reinterpret_cast<uint32_t *>(out)[idx], MODE == OUT_HEAT8 ? heat8(steps) : alpha8(steps));
    }
};

// blockIdx -> 8x8 tile of the local rows, XCD-aware (see k_plain): blocks b and b+8 share an XCD and its
// L2; XCD k renders tile rows k, k+8, ...  Returns 0xFFFFFFFF for the padding blocks.
__device__ __forceinline__ uint32_t tile_of_block(const RenderParams &P, uint32_t bid, uint32_t frame)
{
    const uint32_t xcd = (bid + frame * (P.tiles_y & 7u)) & 7u, j = bid >> 3;   // j-th block of this label
    const uint32_t r = j / P.tiles_x, cx = j - r * P.tiles_x;
    const uint32_t row = r * 8 + xcd;
    return row < P.tiles_y ? row * P.tiles_x + cx : 0xFFFFFFFFu;
}
// Fill count of queue q of frame f, one 128-byte line each, in two sets: a launch pair uses set
// P.hit_set, and its k_shadow zeroes the other set for the next pair on this scratch -- nothing else
// touches that set meanwhile (launches that share a scratch run in stream order), so the queues are
// emptied without a "last workgroup" counter (8192 atomic adds on one word serialise at ~90 per
// microsecond: that alone took 90 us per frame in the first version).
__device__ __forceinline__ uint32_t *hit_count(const RenderParams &P, uint32_t set, uint32_t f, uint32_t q)
{
    return P.hit_ctl + (((size_t)set * MAX_BATCH + f) * HIT_QUEUES + q) * 32u;
}

// The shadow march of Compute.hlsl:214-230 for a lane whose RayState holds the shading step's results (pos, dir, prox,
// dist; n = 0).  Every exit is black (:223, :229) except the one that reaches the light (:215-219): returns that.
// One loop exit, as in the primary march: the three tests are combined without a branch (the gradient of :221-223 only for
// the lanes close to the surface), and which one ended the march is read off the lane's final state afterwards.
template <bool COUNT, class CursorT>
__device__ __forceinline__ bool shadow_march(const RenderParams &P, const FrameInfo &I, RayState &r, CursorT &c,
                                             unsigned long long &cn, unsigned long long &cs, StepClasses *classes = nullptr)
{
    // any(pos < 0) || any(pos > 1) as min3 / max3: v_min3_f32 and v_max3_f32 skip NaN operands, and a
    // comparison with NaN is false either way, so the two forms agree for every input
    auto header = [&]() { return r.n < 40 && r.prox > -I.margin; };
    auto at_light = [&]() {
        const float lo = __builtin_fminf(__builtin_fminf(r.px, r.py), r.pz), hi = __builtin_fmaxf(__builtin_fmaxf(r.px, r.py), r.pz);
        return r.prox > r.dist || lo < 0.0f || hi > 1.0f;
    };
    for (;;) {
        bool go = ((int)header() & (int)!at_light()) != 0;
        if (go && r.prox < I.margin) {
            float gx, gy, gz;
            gradient(cell_of(c, P), r.px, r.py, r.pz, gx, gy, gz);
            go = !(dot3(gx, gy, gz, r.dx, r.dy, r.dz) < 0.0f);
        }
        if (!go) break;
        const float qx = r.px, qy = r.py, qz = r.pz;
        uint32_t reads = march_step(P, I, r, c, nullptr, 0);
        if (COUNT) { cn += reads; cs += 1; if (classes) classify_step(*classes, c, P, qx, qy, qz); }
    }
    asm volatile("" : "+v"(r.prox), "+v"(r.n));       // (see k_march: keeps the exit reasons out of the loop)
    return header() && at_light();
}

// QUEUE = false (the default): the wave marches its own shadow rays after the shading step.  QUEUE = true
// (SDFHIP_FLAG_COMPACT): a wave with fewer than P.hit_min of them appends them to the queue that k_shadow marches 64 to a wave.
template <int CUR, bool COUNT, int MODE, bool QUEUE = false>
__global__ __launch_bounds__(64, PLAIN_WAVES_PER_SIMD) void k_march(RenderParams P)
{
    typedef typename CursorOf<CUR, COUNT>::type CursorT;
    // grid: x = 8 tiles_x (XCD label in the low three bits, tile column above), y = frame of the batch, z = groups of eight tile rows:
    // tile_of_block's mapping (XCD k renders tile rows k, k + 8, ...) read off the block's coordinates, without its division
    // (round 5: the frame is y and the row group z, not the other way round -- the dispatcher walks x, then y, then z, so the
    // frames of a batch are neighbours in the dispatch order and a row group of ALL frames starts before the next one: the last
    // frame's long waves no longer start when seven eighths of the launch have been dispatched.  A rank's 20-step burst at 8 ranks
    // 5.43 -> 5.79x.)  Or, for a launch in tile order (P.tile_perm): x = XCD label, y = frame, z = order slot.
    const uint32_t f = blockIdx.y;
    FrameInfo I = P.frames[f];
    // the scalars every march step reads stay in SGPRs: left alone, the compiler reloads them
    // from the kernel arguments (s_load + s_waitcnt) in every iteration
    asm volatile("" : "+s"(I.margin2), "+s"(I.limit));
    // the kernel arguments this wave's set-up reads, asked for together: left to itself the compiler fetches each where it is first
    // used -- a dozen scalar loads, each waited for by itself, in a row at the start of every wave
    {
        const uint32_t a0 = P.width, a1 = P.height, a2 = P.nrows_out, a3 = P.tiles_x, a4 = P.tiles_y, a5 = P.n_band_list, a6 = P.band_first,
                       a7 = P.band_stride, a8 = P.n_tiles;
        const int32_t b0 = P.top_level, b1 = P.fine_bits;
        const void *p0 = P.nodes, *p1 = P.top, *p2 = P.fine, *p3 = P.out, *p4 = P.tile_perm, *p5 = P.tile_cost;
        asm volatile("" :: "s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(a5), "s"(a6), "s"(a7), "s"(a8), "s"(b0), "s"(b1),
                     "s"(p0), "s"(p1), "s"(p2), "s"(p3), "s"(p4), "s"(p5));
    }
    uint32_t tile, tx, ty;
    if (P.tile_perm) {
        // a launch in tile order.  (label-major: the slots of one XCD label are neighbours in memory)
        const uint32_t wg = blockIdx.z * gridDim.x + blockIdx.x;
        const uint32_t e = P.tile_perm[(wg & 7u) * P.perm_per_label + (wg >> 3)];  // tile row << 16 | tile column; all ones: an idle workgroup
        tx = e & 0xFFFFu; ty = e >> 16;
        if (ty >= P.tiles_y || tx >= P.tiles_x) return;
        tile = ty * P.tiles_x + tx;
    } else {
        tx = blockIdx.x >> 3;
        ty = blockIdx.z * 8u + ((blockIdx.x + f * (P.tiles_y & 7u)) & 7u);
        if (ty >= P.tiles_y) return;
        tile = ty * P.tiles_x + tx;
    }
    const uint32_t lane = threadIdx.x;
    const uint32_t x = tx * 8 + (lane & 7u), yl = ty * 8 + (lane >> 3);
    unsigned long long cn = 0, cs = 0, ct = 0;   // nodes, samples, steps (of the pixels that end here)
    StepClasses classes;                         // (counting builds only)
    SparseLane wire_px{0.0f, 0u};                // (OUT_SPARSE only; a lane without a pixel keeps a = +0, code 0)
    bool live = x < P.width && yl < P.nrows_out;
    uint32_t y = 0;
    if (live) { y = global_row(P, yl); live = y < P.height; }
    RayState r;
    CursorT c;
    c.loads = 0;
    const NodeRec root = P.nodes[0];
    start_pixel(I, root, live ? x : 0u, live ? y : 0u, r, c);
    // Compute.hlsl:194-203.  One exit: the loop-header test and the escape test are evaluated together and
    // told apart after the loop (a lane that left keeps its state), header first, as the shader orders them.
    auto marching = [&]() { return (r.prox > I.margin2 || r.prox < 0.0f) && r.n < 100; };
    // (Flat and non-flat lanes share an iteration.  Separating them in time -- flat lanes step while the others wait, then all sample
    // together -- was built bit-identical in round 5 and is 1.5-2.8x slower: profiles/r05_flat_run_ab.txt, commit 972d54c.)
    if (live) {
        // (both tests every time, combined without a branch: the escape test is three instructions)
        auto go_on = [&]() { return ((int)marching() & (int)!(dot3(r.px, r.py, r.pz, r.px, r.py, r.pz) > I.limit)) != 0; };
        if (go_on()) {                               // the first step, from the root, apart: see find_fresh
            float qx = r.px, qy = r.py, qz = r.pz;
            uint32_t reads = march_step<CursorT, true>(P, I, r, c, nullptr, 0);
            if (COUNT) { cn += reads; cs += 1; classify_step(classes, c, P, qx, qy, qz); }
            while (go_on()) {
                if (COUNT) { qx = r.px; qy = r.py; qz = r.pz; }
                reads = march_step(P, I, r, c, nullptr, 0);
                if (COUNT) { cn += reads; cs += 1; classify_step(classes, c, P, qx, qy, qz); }
            }
        }
    }
    // (the lane's own registers say why it left; hiding them from the optimiser here keeps the loop from carrying
    // one more exec-mask per exit reason through every iteration)
    asm volatile("" : "+v"(r.prox), "+v"(r.n));
    const int end = !live ? 3 : marching() ? 2 : 1;   // 1 on the surface (or out of steps), 2 escaped, 3 no pixel
    if (P.tile_cost) {                                  // iterations this wave's primary loop ran = its longest lane
        int m = live ? r.n : 0;
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
        if (lane == 0) P.tile_cost[(size_t)f * P.n_tiles + tile] = (uint16_t)m;
    }
    // ---- the wave is converged again ----
    const size_t lidx = (size_t)yl * P.width + x;
    if (end == 2) {
        const PixelSink<MODE> dst(P, f, &wire_px);
        dst.sky(lidx, (float)r.n);
        if (COUNT) ct = (unsigned long long)r.n;
    }
    // The shading step (Compute.hlsl:205-213) for the lanes that ended on the surface -- all of them at once, after the
    // loop, not whenever one of them gets there: a pixel that faces away from the light is finished (black); the others
    // go to the queue of shadow rays with what the shadow march needs.
    bool shadow = false;
    if (__ballot(end == 1)) {
        if (end == 1) {
            float lx = I.lightx - r.px, ly = I.lighty - r.py, lz = I.lightz - r.pz;
            const float rl = 1.0f / sqrtf(dot3(lx, ly, lz, lx, ly, lz));
            r.dx = lx * rl; r.dy = ly * rl; r.dz = lz * rl;
            r.px = __builtin_fmaf(r.dx, I.margin, r.px);
            r.py = __builtin_fmaf(r.dy, I.margin, r.py);
            r.pz = __builtin_fmaf(r.dz, I.margin, r.pz);
            float gx, gy, gz;
            gradient(cell_of(c, P), r.px, r.py, r.pz, gx, gy, gz);
            const float rg = 1.0f / sqrtf(dot3(gx, gy, gz, gx, gy, gz));
            r.angle = dot3(r.dx, r.dy, r.dz, gx * rg, gy * rg, gz * rg);
            if (r.angle < 0.0f) {
                const PixelSink<MODE> dst(P, f, &wire_px);
                dst.black(lidx, (float)r.n);
                if (COUNT) ct = (unsigned long long)r.n;
            } else {
                shadow = true;
            }
        }
    }
    // QUEUE: a wave that holds at least P.hit_min shadow rays marches them in place -- the queue (and the second kernel) take the
    // rays of the sparse waves only, the silhouettes and shadow edges, where a wave of 64 lanes would march a handful
    bool in_place = !QUEUE;
    if constexpr (QUEUE) in_place = (uint32_t)__popcll(__ballot(shadow)) >= P.hit_min;
    if (in_place) {
        if (shadow) {
            const PixelSink<MODE> dst(P, f, &wire_px);
            const float lx = I.lightx - r.px, ly = I.lighty - r.py, lz = I.lightz - r.pz;      // Compute.hlsl:212
            r.dist = sqrtf(dot3(lx, ly, lz, lx, ly, lz)) / 2.0f;
            r.base = r.n; r.n = 0; r.phase = PH_SHADOW;           // i stays, j starts; steps are prox + margin from here
            const bool lit = shadow_march<COUNT>(P, I, r, c, cn, cs, &classes);
            if (lit) dst.grey(lidx, r.angle / (r.dist * r.dist) * I.k_strength, (float)(r.base + r.n));
            else dst.black(lidx, (float)(r.base + r.n));
            if (COUNT) ct = (unsigned long long)(r.base + r.n);
        }
        if (P.tile_cost) {                              // ... and, in the high byte, the iterations of its shadow loop
            int m = shadow ? r.n : 0;
            for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
            if (lane == 0) P.tile_cost[(size_t)f * P.n_tiles + tile] |= (uint16_t)(m << 8);
        }
        if (MODE == OUT_SPARSE) sparse2_flush(P, f, tile, lane, wire_px);     // the tile's share of the sparse wire share
        if (COUNT) { flush_counters(P, cn, cs, ct, shadow ? 1u : 0u, c.loads, 0u); flush_classes(P, classes); }
        return;
    }
    if constexpr (QUEUE) {
    const unsigned long long hits = __ballot(shadow);
    if (hits) {
        const uint32_t q = (blockIdx.z * gridDim.x + blockIdx.x) & (HIT_QUEUES - 1u);     // (the workgroup's number in the frame: what the queues' capacity counts)
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(hit_count(P, P.hit_set, f, q), (uint32_t)__popcll(hits));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(hits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hits, 0u));
        if (shadow && base + rank < P.hit_cap) {       // always true while the fill counts start at zero (k_shadow leaves them so)
            const size_t slot = ((size_t)f * HIT_QUEUES + q) * P.hit_cap + base + rank;
            P.hit_a[slot] = make_float4(r.px, r.py, r.pz, r.prox);
            P.hit_b[slot] = c.pack();
            P.hit_c[slot] = make_uint4((uint32_t)lidx, (uint32_t)r.n, c.v0, c.v1);
            P.hit_d[slot] = make_float4(r.dx, r.dy, r.dz, r.angle);
        }
    }
    if (COUNT) flush_counters(P, cn, cs, ct, shadow ? 1u : 0u, c.loads, shadow ? 1u : 0u);
    }
}

// ---- the second kernel of SDFHIP_FLAG_COMPACT (k_march<..., QUEUE = true> fills the queue) ----
// One lane per queued shadow ray; a wave takes 64 consecutive records of one queue at a time (chunks are
// numbered over the frame's queues: a wave-wide scan of the 64 fill counts, once per wave).
template <int CUR, bool COUNT, int MODE>
__global__ __launch_bounds__(64, PLAIN_WAVES_PER_SIMD) void k_shadow(RenderParams P)
{
    typedef typename CursorOf<CUR, COUNT>::type CursorT;
    const uint32_t f = blockIdx.y, lane = threadIdx.x;
    FrameInfo I = P.frames[f];
    asm volatile("" : "+s"(I.margin));
    const PixelSink<MODE> dst(P, f);
    // chunks of queue `lane`, and their running sum over the queues (inclusive scan across the wave)
    const uint32_t fill = min(*hit_count(P, P.hit_set, f, lane), P.hit_cap);
    if (blockIdx.x == 0 && f == 0)                       // empty the other set -- every frame of it -- for the next launch pair
        for (uint32_t ff = 0; ff < (uint32_t)MAX_BATCH; ff++) *hit_count(P, P.hit_set ^ 1u, ff, lane) = 0u;
    const uint32_t chunks = (fill + 63u) >> 6;
    uint32_t incl = chunks;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if ((int)lane >= o) incl += v; }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    unsigned long long cn = 0, cs = 0, ct = 0, cl = 0;
    for (uint32_t t = blockIdx.x; t < total; t += gridDim.x) {
        // queue of chunk t = the number of queues whose inclusive sum is <= t
        const uint32_t q = (uint32_t)__popcll(__ballot(incl <= t));
        const uint32_t q_incl = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)q), q_chunks = (uint32_t)__builtin_amdgcn_readlane((int)chunks, (int)q);
        const uint32_t q_fill = (uint32_t)__builtin_amdgcn_readlane((int)fill, (int)q);
        const uint32_t i = (t - (q_incl - q_chunks)) * 64u + lane;
        if (i < q_fill) {
            const size_t slot = ((size_t)f * HIT_QUEUES + q) * P.hit_cap + i;
            const float4 a = P.hit_a[slot], d = P.hit_d[slot];
            const int4 b = P.hit_b[slot];
            const uint4 e = P.hit_c[slot];
            RayState r;
            CursorT c;
            r.px = a.x; r.py = a.y; r.pz = a.z; r.prox = a.w; r.dx = d.x; r.dy = d.y; r.dz = d.z; r.angle = d.w;
            r.base = (int)e.y; r.n = 0; r.phase = PH_SHADOW;          // i stays, j starts
            c.unpack(b, CursorT::units_shift(P.top_level + (CUR == CUR_STACK_SPLIT ? P.fine_bits : 0)));
            c.v0 = e.z; c.v1 = e.w; c.loads = 0;
            const size_t lidx = e.x;
            const float lx = I.lightx - r.px, ly = I.lighty - r.py, lz = I.lightz - r.pz;      // Compute.hlsl:212
            r.dist = sqrtf(dot3(lx, ly, lz, lx, ly, lz)) / 2.0f;
            const bool lit = shadow_march<COUNT>(P, I, r, c, cn, cs);
            if (lit) dst.grey(lidx, r.angle / (r.dist * r.dist) * I.k_strength, (float)(r.base + r.n));
            else dst.black(lidx, (float)(r.base + r.n));
            if (COUNT) { ct += (unsigned long long)(r.base + r.n); cl += c.loads; }
        }
    }
    if (COUNT) flush_counters(P, cn, cs, ct, 0, cl);
}

// ---- persistent waves with lane refill and state batching (wavefront ray compaction) --
// One wave per workgroup, as many workgroups as the chip holds.  Lane states:
// PRIMARY / SHADOW (marching), SHADE (march over, shading pending), IDLE (no pixel).
//   * refill: pixels are numbered in 8x8-tile order, p = tile*64 + (y&7)*8 + (x&7).  A wave
//     owns the range [cur, end) of one tile at a time and hands the next pixels of it to its
//     idle lanes: rank = number of idle lanes below me (ballot + mbcnt), lane gets pixel
//     cur + rank.  Tiles come from 8 atomic queues, one per XCD label (blockIdx & 7): queue q
//     holds tile rows q, q+8, ... (the plain kernel's mapping), and a wave whose queue is
//     empty steals from the next ones.
//   * batching: refill runs only once REFILL_MIN lanes are idle, shading only once SHADE_MIN
//     lanes wait for it (or nothing else can run), so those long divergent blocks execute
//     for many lanes at a time instead of once per straggler.
// A pixel's result depends on its coordinates only, so the image is the plain kernel's
// bit for bit.  Exit: every wave leaves once all queues are exhausted and its lanes idle.
constexpr int REFILL_MIN = 12;
constexpr int SHADE_MIN = 8;

template <int CUR, bool COUNT>
__global__ __launch_bounds__(64, COMPACT_WAVES_PER_SIMD) void k_compact(RenderParams P)
{
    __shared__ int32_t stack_lds[CUR ? MAX_STACK * 64 : 1];
    __shared__ float4 out_lds[64];
    const uint32_t lane = threadIdx.x;
    FrameInfo I = P.frames[0];
    asm volatile("" : "+s"(I.margin), "+s"(I.margin2), "+s"(I.limit));     // see k_plain
    const NodeRec root = P.nodes[0];
    const LdsSink dst{&out_lds[lane]};
    unsigned long long cn = 0, cs = 0, ct = 0, cr = 0;   // nodes, samples, steps, shadow rays
    RayState r;
    typename CursorOf<CUR, COUNT>::type c;
    uint32_t pix = 0;           // x | yl << 16
    uint32_t cur = 0, end = 0;  // wave-uniform: pixel range of the current tile
    uint32_t q = blockIdx.x & 7u, tried = 0;   // wave-uniform: queue in use, queues found empty
    bool more = true, first = true;   // wave-uniform
    r.px = r.py = r.pz = r.dx = r.dy = r.dz = r.prox = r.angle = r.dist = 0.0f;
    r.n = r.base = 0;
    r.phase = PH_IDLE;
    c.loads = 0;
    c.reset(root);
    const uint32_t rows_q = (P.tiles_y + 7u) >> 3;         // tile rows per queue (upper bound)

    for (;;) {
        unsigned long long m_idle = __ballot(r.phase >= PH_IDLE);
        int n_idle = __popcll(m_idle);
        int n_shade = __popcll(__ballot(r.phase == PH_SHADE));
        int n_march = 64 - n_idle - n_shade;
        if (n_idle >= REFILL_MIN || (n_march == 0 && n_shade == 0)) {
            // flush the colours of the pixels finished since the last refill: one batch of
            // stores, whose completion the wave waits for once (behind start_pixel's work)
            if (r.phase == PH_DONE) {
                const size_t pidx = (size_t)(pix >> 16) * P.width + (pix & 0xFFFFu);
                if (P.out_mode == 0u) P.out[pidx] = out_lds[lane];
                else reinterpret_cast<uint32_t *>(P.out)[pidx] = display8(out_lds[lane], P.out_mode);
                r.phase = PH_IDLE;
            }
            while (more && n_idle > 0) {
                if (cur == end) {
                    // first tile of a wave: its own index within its queue, no atomic (all
                    // waves start together; thousands of simultaneous adds on one word
                    // serialise at ~90 per microsecond).  Later tiles: the queue head, one
                    // 128-byte line per queue, counts on from where the static ones end.
                    uint32_t t = blockIdx.x >> 3;
                    if (!first) {
                        if (lane == 0) t = atomicAdd(P.queue + q * 32u, 1u);
                        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t) + ((gridDim.x + 7u - q) >> 3);
                    }
                    first = false;
                    const uint32_t trow = (t / P.tiles_x) * 8u + q;
                    if (t >= rows_q * P.tiles_x || trow >= P.tiles_y) {   // this queue is empty: steal
                        q = (q + 1u) & 7u;
                        if (++tried >= 8u) more = false;
                        continue;
                    }
                    cur = (trow * P.tiles_x + (t % P.tiles_x)) * 64u;
                    end = cur + 64u;
                }
                uint32_t take = min((uint32_t)n_idle, end - cur);
                uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_idle >> 32),
                                    __builtin_amdgcn_mbcnt_lo((uint32_t)m_idle, 0u));
                if (r.phase >= PH_IDLE && rank < take) {
                    uint32_t p = cur + rank, tile = p >> 6, qq = p & 63u;
                    uint32_t x = (tile % P.tiles_x) * 8 + (qq & 7u);
                    uint32_t yl = (tile / P.tiles_x) * 8 + (qq >> 3);
                    if (x < P.width && yl < P.nrows_out) {
                        uint32_t y = global_row(P, yl);
                        if (y < P.height) {
                            start_pixel(I, root, x, y, r, c);
                            pix = x | (yl << 16);
                        }
                    }
                }
                cur += take;
                m_idle = __ballot(r.phase >= PH_IDLE);
                n_idle = __popcll(m_idle);
            }
            n_march = 64 - n_idle - n_shade;
        }
        if (n_march == 0 && n_shade == 0) {
            if (!more) break;
            continue;
        }
        if (n_shade >= SHADE_MIN || n_march == 0) {
            if (r.phase == PH_SHADE) {
                r.phase = PH_PRIMARY;
                if (shade(I, r, c, dst)) {
                    if (COUNT) ct += (unsigned long long)(r.base + r.n);
                    r.phase = PH_DONE;
                } else if (COUNT) {
                    cr += 1;
                }
            }
        }
        if (r.phase <= PH_SHADOW) {
            bool done = false;
            if (r.phase == PH_PRIMARY) {
                int s = check_primary(I, r, dst);
                if (s == 1) r.phase = PH_SHADE;
                done = s == 2;
            } else {
                done = check_shadow(I, r, c, dst);
            }
            if (done) {
                if (COUNT) ct += (unsigned long long)(r.base + r.n);
                r.phase = PH_DONE;
            } else if (r.phase <= PH_SHADOW) {
                uint32_t reads = march_step(P, I, r, c, stack_lds + lane, 64);
                if (COUNT) { cn += reads; cs += 1; }
            }
        }
    }
    if (COUNT) flush_counters(P, cn, cs, ct, cr, c.loads);
}

// ---- path-traced mode (BASELINE config 5) -----------------------------------------------
// Not in the reference (README "plans" only); defined by o_pixel_pt in oracle/sdf_oracle.c:
// per pixel spp samples, each a jittered camera ray followed by up to 1 + max_bounces
// segments built from the reference's own pieces (primary march, shading, shadow march)
// chained by cosine-weighted diffuse bounces; PCG-hash RNG; no transcendental function.
// The oracle writes that as four nested loops.  After the first bounce every lane of a wave
// is somewhere else in them, and nested loops would run one lane group at a time; here, as
// in k_plain, a lane is a state machine -- MARCH (a segment's march) or SHADOW (its shadow
// march) -- around ONE find + sample + advance body, so whatever phase the 64 lanes are in,
// they share the instruction stream of the expensive part.  The arithmetic and its order per
// pixel are the oracle's; only the control flow differs.  One lane per pixel, one 8x8 tile
// per wave, the plain kernel's XCD-interleaved tile rows.
template <int CUR, bool COUNT>
__global__ __launch_bounds__(64, PATH_WAVES_PER_SIMD) void k_path(RenderParams P)
{
    __shared__ int32_t stack_lds[CUR ? MAX_STACK * 64 : 1];
    const uint32_t bid = blockIdx.x, xcd = bid & 7u, jb = bid >> 3;
    const uint32_t rr = jb / P.tiles_x, cxx = jb - rr * P.tiles_x, row = rr * 8 + xcd;
    if (row >= P.tiles_y) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t x = cxx * 8 + (lane & 7u), yl = row * 8 + (lane >> 3);
    unsigned long long cn = 0, cs = 0, ct = 0, cr = 0, cl = 0;   // nodes, samples, steps, shadow rays, loads
    bool live = x < P.width && yl < P.nrows_out;
    uint32_t y = 0;
    if (live) { y = global_row(P, yl); live = y < P.height; }
    if (live) {
        typedef typename CursorOf<CUR, COUNT>::type CursorT;
        FrameInfo I = P.frames[0];
        asm volatile("" : "+s"(I.margin), "+s"(I.limit));                      // see k_plain
        const NodeRec root = P.nodes[0];
        const uint32_t p = y * P.width + x;
        const float margin = I.margin;
        int32_t *stack = stack_lds + lane;
        float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
        uint32_t steps = 0;
        // lane state
        CursorT c;
        c.loads = 0;
        float mx, my, mz;        // the position being marched (segment, then shadow ray)
        float ux, uy, uz;        // its direction (segment direction, then direction to the light)
        float hx = 0, hy = 0, hz = 0;   // the hit point, kept while the shadow ray marches
        float n0 = 0, n1 = 0, n2 = 0;   // the hit normal, facing the incoming ray
        float T = 1.0f, prox = 1.0f, angle = 0.0f, dist = 0.0f;
        uint32_t s = 0, b = 0;
        int it = 0;              // i of the segment march, or j of the shadow march
        bool shadow = false;

        // start of sample s: reset cursor, jittered camera ray (o_pixel_pt's sample loop head)
        c.reset(root);
        mx = I.posx; my = I.posy; mz = I.posz;
        ray_f(I, (float)x + rnd(P.pt_seed, p, 0, 0, 0), (float)y + rnd(P.pt_seed, p, 0, 0, 1), ux, uy, uz);

        for (;;) {
            // ---- everything between two march steps --------------------------------------
            bool bounce = false, end_sample = false;
            if (!shadow) {
                if ((prox > margin * 2.0f || prox < 0.0f) && it < 100) {
                    if (dot3(mx, my, mz, mx, my, mz) > I.limit) {            // escaped: sky
                        steps += (uint32_t)it;
                        acc0 = __builtin_fmaf(T, 0.005f, acc0); acc1 = __builtin_fmaf(T, 0.01f, acc1); acc2 = __builtin_fmaf(T, 0.2f, acc2);
                        end_sample = true;
                    }
                } else {                                                     // hit: shade (Compute.hlsl:205-213)
                    steps += (uint32_t)it;
                    float lx = I.lightx - mx, ly = I.lighty - my, lz = I.lightz - mz;
                    const float rl = 1.0f / sqrtf(dot3(lx, ly, lz, lx, ly, lz));
                    const float L0 = lx * rl, L1 = ly * rl, L2 = lz * rl;
                    mx = __builtin_fmaf(L0, margin, mx); my = __builtin_fmaf(L1, margin, my); mz = __builtin_fmaf(L2, margin, mz);
                    float g0, g1, g2;
                    gradient(c.cell(), mx, my, mz, g0, g1, g2);
                    const float rg = 1.0f / sqrtf(dot3(g0, g1, g2, g0, g1, g2));
                    n0 = g0 * rg; n1 = g1 * rg; n2 = g2 * rg;
                    angle = dot3(L0, L1, L2, n0, n1, n2);
                    // the bounce needs the normal facing the incoming ray; decide now, while
                    // the incoming direction is still in (ux, uy, uz)
                    const bool flip = dot3(n0, n1, n2, ux, uy, uz) > 0.0f;
                    hx = mx; hy = my; hz = mz;
                    if (!(angle < 0.0f)) {
                        lx = I.lightx - mx; ly = I.lighty - my; lz = I.lightz - mz;
                        dist = sqrtf(dot3(lx, ly, lz, lx, ly, lz)) / 2.0f;
                        ux = L0; uy = L1; uz = L2;
                        shadow = true;
                        if (COUNT) cr += 1;
                        it = 0;
                    } else {
                        bounce = true;
                    }
                    if (flip) { n0 = -n0; n1 = -n1; n2 = -n2; }
                }
            }
            if (shadow) {                                                    // Compute.hlsl:214-223
                bool over = false;
                if (!(it < 40 && prox > -margin)) {
                    over = true;
                } else if (prox > dist || (mx < 0.0f || my < 0.0f || mz < 0.0f) ||
                           (mx > 1.0f || my > 1.0f || mz > 1.0f)) {
                    // angle was computed with the unflipped normal; n is stored flipped
                    const float e = T * (P.pt_albedo * (angle / (dist * dist) * I.k_strength));
                    acc0 += e; acc1 += e; acc2 += e;
                    over = true;
                } else if (prox < margin) {
                    float q0, q1, q2;
                    gradient(c.cell(), mx, my, mz, q0, q1, q2);
                    if (dot3(q0, q1, q2, ux, uy, uz) < 0.0f) over = true;
                }
                if (over) {
                    steps += (uint32_t)it;
                    shadow = false;
                    bounce = true;
                }
            }
            if (bounce) {
                if (b == P.pt_bounces) {
                    end_sample = true;
                } else {                                                     // o_pixel_pt "diffuse bounce"
                    float u0 = n0, u1 = n1, u2 = n2, q = 1.0f;
                    for (uint32_t a = 0; a < 8; a++) {
                        const float c0 = rnd(P.pt_seed, p, s, b + 1, 3 * a) * 2.0f - 1.0f;
                        const float c1 = rnd(P.pt_seed, p, s, b + 1, 3 * a + 1) * 2.0f - 1.0f;
                        const float c2 = rnd(P.pt_seed, p, s, b + 1, 3 * a + 2) * 2.0f - 1.0f;
                        const float qq = dot3(c0, c1, c2, c0, c1, c2);
                        if (qq <= 1.0f && qq > 1e-12f) { u0 = c0; u1 = c1; u2 = c2; q = qq; break; }
                    }
                    const float ru = 1.0f / sqrtf(q);
                    float d0 = __builtin_fmaf(u0, ru, n0), d1 = __builtin_fmaf(u1, ru, n1), d2 = __builtin_fmaf(u2, ru, n2);
                    float qd = dot3(d0, d1, d2, d0, d1, d2);
                    if (!(qd > 1e-12f)) { d0 = n0; d1 = n1; d2 = n2; qd = dot3(n0, n1, n2, n0, n1, n2); }
                    const float rd = 1.0f / sqrtf(qd);
                    ux = d0 * rd; uy = d1 * rd; uz = d2 * rd;
                    const float off = margin * 4.0f;
                    mx = __builtin_fmaf(n0, off, hx); my = __builtin_fmaf(n1, off, hy); mz = __builtin_fmaf(n2, off, hz);
                    T *= P.pt_albedo;
                    b++;
                    prox = 1.0f;
                    it = 0;
                    continue;                   // the new segment starts with its loop-header checks
                }
            }
            if (end_sample) {
                s++;
                if (s == P.pt_spp) break;
                c.reset(root);
                mx = I.posx; my = I.posy; mz = I.posz;
                ray_f(I, (float)x + rnd(P.pt_seed, p, s, 0, 0), (float)y + rnd(P.pt_seed, p, s, 0, 1), ux, uy, uz);
                T = 1.0f; b = 0; prox = 1.0f; it = 0;
                continue;
            }
            // ---- one march step of the segment or of the shadow ray -----------------------
            typename CursorT::Pos u;
            uint32_t reads = find(c, P.nodes, grid_of(P), P.n_nodes, stack, 64, mx, my, mz, u);
            prox = sample_after_find(c, u, mx, my, mz);
            if (COUNT) { cn += reads; cs += 1; }
            const float st = shadow ? prox + margin : prox;
            mx = __builtin_fmaf(ux, st, mx);
            my = __builtin_fmaf(uy, st, my);
            mz = __builtin_fmaf(uz, st, mz);
            it++;
        }
        const float inv = (float)P.pt_spp;
        P.out[(size_t)yl * P.width + x] = make_float4(acc0 / inv, acc1 / inv, acc2 / inv, (float)steps);
        if (COUNT) { ct = steps; cl = c.loads; }
    }
    if (COUNT) flush_counters(P, cn, cs, ct, cr, cl);
}

// =====================================================================================
// Path-traced mode as a pipeline of kernels -- the default wherever a find is a grid lookup.
//
// k_path (above) gives a lane a pixel and walks its 16 samples x up to 4 segments: on a silhouette tile the
// sky lanes (80 % of the frame) finish their 16 one-segment samples in a fifth of the time the surface lanes
// need, and whoever bounces marches alone: lanes were busy in 31 % of the VALU thread-cycles.  Here a lane is
// a PATH (pixel, sample), and a kernel is one level of it:
//   k_pt_primary   the camera segment of every path: 8x8 pixels per wave, the samples one after another (all
//                  64 lanes march sample s together: neighbouring, nearly parallel rays).  A path that escapes
//                  is finished: its sky throughput and steps go to the path results.  A path that hits appends
//                  its state to hit queue 0.
//   k_pt_bounce    level b = 0 .. bounces: one lane per queued hit, 64 consecutive entries per wave: the shading
//                  step and the shadow march (the light the vertex receives -> pt_e[b]), then -- unless b is the last
//                  level -- the diffuse bounce and the march of the next segment, which ends in the sky (-> pt_t)
//                  or in hit queue b + 1.
//   k_pt_resolve   per pixel, the oracle's accumulation replayed in its order: for every sample, the vertices'
//                  light in bounce order, then the sky term; the mean and the step total.
// Float addition is not associative, so the paths do not add into a shared pixel: they leave their addends
// (adding the +0 of an unlit vertex, or fma(0, sky, acc), changes no bit of a non-negative or NaN sum) and
// k_pt_resolve adds them as o_pixel_pt does.  Per path the arithmetic, the RNG draws and the cursor carried
// from segment to shadow ray to next segment are the oracle's: images and counters stay bit-identical.
// (Measured and dropped: k_pt_bounce with lane refill -- a wave owning a share of the level's hits, lanes as
// SHADOW / MARCH / parked state machines around one shared march step, hits loaded, shaded and bounced in batches
// once 16 / 32 / 48 lanes are free.  Bit-identical, 30.8 / 29.1 / 28.6 ms per cfg-5 frame against 26.3 ms for the
// plain form below: every batch stalls the marching lanes behind the queue loads and the shading of the new ones.)
// =====================================================================================
__device__ __forceinline__ uint32_t *pt_count(const RenderParams &P, uint32_t queue, uint32_t q) { return P.pt_ctl + ((size_t)queue * HIT_QUEUES + q) * 32u; }

// a path's state at a surface hit, three 16-byte records (SoA over the queue: record r of entry i at [r * total + i]):
//   {pos, prox}   where the march ended, its last sample
//   {cursor coordinates and level in two words (pt_pack_cursor), the cursor's two value words}
//   {path = pixel * spp + sample (pixel = row * width + x of the LOCAL rows), incoming direction}
// The throughput is not stored: it is albedo multiplied level times onto 1, in that order, and the next level does just that.
// (Round 2's entry had a fourth record: the cursor in four words, the throughput, and a step count nobody read.)
constexpr uint32_t PT_RECORDS = 3;
// (ax, ay, az, s) of CursorFT::pack() -- coordinates below 2^LM = 4096 or the root mark on all three, s = LM - level | FLAT_BIT -- in 64 bits
__device__ __forceinline__ uint2 pt_pack_cursor(const int4 &k)
{
    const bool fresh = k.x == 0x40000000;
    const uint32_t x = fresh ? 0u : (uint32_t)k.x, y = fresh ? 0u : (uint32_t)k.y, z = fresh ? 0u : (uint32_t)k.z, s = (uint32_t)k.w;
    return make_uint2(x | (y << 13) | (z << 26), (z >> 6) | ((s & 31u) << 7) | ((s >> 31) << 12) | ((fresh ? 1u : 0u) << 13));
}
__device__ __forceinline__ int4 pt_unpack_cursor(const uint2 &w)
{
    const bool fresh = ((w.y >> 13) & 1u) != 0u;
    const int32_t x = (int32_t)(w.x & 0x1FFFu), y = (int32_t)((w.x >> 13) & 0x1FFFu), z = (int32_t)((w.x >> 26) | ((w.y & 0x7Fu) << 6));
    const uint32_t s = ((w.y >> 7) & 31u) | (((w.y >> 12) & 1u) << 31);
    return make_int4(fresh ? 0x40000000 : x, fresh ? 0x40000000 : y, fresh ? 0x40000000 : z, (int32_t)s);
}
// the hit queues are written once and read once, a level later: streamed past the caches (nt), so that the L2 lines they
// would take stay with the grid cells the bounce rays look up
__device__ __forceinline__ void nt_store(float4 *p, const float4 &v)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4 *>(p));
}
__device__ __forceinline__ float4 nt_load(const float4 *p)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
template <class CursorT>
__device__ __forceinline__ void pt_push(const RenderParams &P, uint32_t queue, uint32_t q, bool hit, uint32_t lane,
                                        float px, float py, float pz, float prox, const CursorT &c, uint32_t pid,
                                        float ux, float uy, float uz)
{
    const unsigned long long hits = __ballot(hit);
    if (!hits) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(pt_count(P, queue, q), (uint32_t)__popcll(hits));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(hits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hits, 0u));
    // (the capacity covers the worst case of every sub-queue -- sdfhip_device.hip sizes it so -- and a hit that found no room
    // anyway is not lost silently: the word behind the fill counts says so, and the synchronous entry points return an error)
    if (hit && base + rank >= P.pt_cap) atomicOr(P.pt_ctl + (size_t)2 * HIT_QUEUES * 32u, 1u);
    if (hit && base + rank < P.pt_cap) {
        const size_t total = (size_t)HIT_QUEUES * P.pt_cap, i = (size_t)q * P.pt_cap + base + rank;
        float4 *Q = P.pt_q[queue];
        nt_store(&Q[i], make_float4(px, py, pz, prox));
        const uint2 k = pt_pack_cursor(c.pack());
        nt_store(&Q[total + i], make_float4(__uint_as_float(k.x), __uint_as_float(k.y), __uint_as_float(c.v0), __uint_as_float(c.v1)));
        nt_store(&Q[2 * total + i], make_float4(__uint_as_float(pid), ux, uy, uz));
    }
}

// the march of one path segment (o_pixel_pt's inner loop, Compute.hlsl:194-203): true = escaped to the sky
// FRESH: the cursor comes straight from reset() (a camera segment), see find_fresh
template <bool COUNT, bool FRESH = false, class CursorT>
__device__ __forceinline__ bool pt_march(const RenderParams &P, const FrameInfo &I, RayState &r, CursorT &c,
                                         unsigned long long &cn, unsigned long long &cs)
{
    auto marching = [&]() { return (r.prox > I.margin2 || r.prox < 0.0f) && r.n < 100; };
    auto go_on = [&]() { return marching() && !(dot3(r.px, r.py, r.pz, r.px, r.py, r.pz) > I.limit); };
    if (FRESH && go_on()) {
        uint32_t reads = march_step<CursorT, true>(P, I, r, c, nullptr, 0);
        if (COUNT) { cn += reads; cs += 1; }
    }
    while (go_on()) {
        uint32_t reads = march_step(P, I, r, c, nullptr, 0);
        if (COUNT) { cn += reads; cs += 1; }
    }
    return marching();
}

template <int CUR, bool COUNT>
__global__ __launch_bounds__(64, PLAIN_WAVES_PER_SIMD) void k_pt_primary(RenderParams P)
{
    typedef typename CursorOf<CUR, COUNT>::type CursorT;
    FrameInfo I = P.frames[0];
    asm volatile("" : "+s"(I.margin2), "+s"(I.limit));
    const uint32_t tile = tile_of_block(P, blockIdx.x, 0);
    if (tile >= P.n_tiles) return;
    const uint32_t tx = tile % P.tiles_x, ty = tile / P.tiles_x, lane = threadIdx.x;
    const uint32_t x = tx * 8 + (lane & 7u), yl = ty * 8 + (lane >> 3);
    unsigned long long cn = 0, cs = 0, cl = 0;
    bool live = x < P.width && yl < P.nrows_out;
    uint32_t y = 0;
    if (live) { y = global_row(P, yl); live = y < P.height; }
    const NodeRec root = P.nodes[0];
    const size_t npx = (size_t)P.nrows_out * P.width, lidx = (size_t)yl * P.width + x;
    const uint32_t p = y * P.width + x;                       // the pixel index that seeds the RNG: of the FRAME
    const uint32_t q = blockIdx.x & (HIT_QUEUES - 1u);
    for (uint32_t s = 0; s < P.pt_spp; s++) {
        RayState r;
        CursorT c;
        c.loads = 0;
        c.reset(root);
        r.px = I.posx; r.py = I.posy; r.pz = I.posz;
        ray_f(I, (float)x + rnd(P.pt_seed, p, s, 0, 0), (float)y + rnd(P.pt_seed, p, s, 0, 1), r.dx, r.dy, r.dz);
        r.prox = 1.0f; r.n = 0; r.base = 0; r.phase = PH_PRIMARY; r.angle = 0.0f; r.dist = 0.0f;
        bool escaped = false;
        if (live) escaped = pt_march<COUNT, true>(P, I, r, c, cn, cs);
        if (COUNT) cl += c.loads;
        if (live) {
            const size_t o = (size_t)s * npx + lidx;
            __builtin_nontemporal_store(escaped ? 1.0f : 0.0f, &P.pt_t[o]);   // the camera ray's throughput is 1
            __builtin_nontemporal_store((uint32_t)r.n, &P.pt_n[o]);           // no vertex yet
        }
        pt_push(P, 0, q, live && !escaped, lane, r.px, r.py, r.pz, r.prox, c, (uint32_t)lidx * P.pt_spp + s, r.dx, r.dy, r.dz);
    }
    if (COUNT) flush_counters(P, cn, cs, 0, 0, cl);
}

// chunk t of 64 entries of queue `queue` -> (sub-queue, first entry, fill): the bounce kernel's numbering (a wave-wide scan of the fills)
struct PtChunks {
    uint32_t fill, chunks, incl, nchunks;
    __device__ __forceinline__ PtChunks(const RenderParams &P, uint32_t queue, uint32_t lane)
    {
        fill = min(*pt_count(P, queue, lane), P.pt_cap);
        chunks = (fill + 63u) >> 6;
        incl = chunks;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if ((int)lane >= o) incl += v; }
        nchunks = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    // the entry lane `lane` takes of chunk t: its index in the queue's arrays, or 0xFFFFFFFF
    __device__ __forceinline__ uint32_t entry(const RenderParams &P, uint32_t t, uint32_t lane) const
    {
        const uint32_t q = (uint32_t)__popcll(__ballot(incl <= t));
        const uint32_t q_incl = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)q), q_chunks = (uint32_t)__builtin_amdgcn_readlane((int)chunks, (int)q);
        const uint32_t q_fill = (uint32_t)__builtin_amdgcn_readlane((int)fill, (int)q);
        const uint32_t i = (t - (q_incl - q_chunks)) * 64u + lane;
        return i < q_fill ? q * P.pt_cap + i : 0xFFFFFFFFu;
    }
};

#ifdef SDFHIP_EXPERIMENTS
// ---- LABORATORY (experiments build): the bounce levels' queue entries ordered by (region of the hit, octant of the outgoing direction) ----
// MEASURED: NO GAIN (profiles/r04_cfg5_sort_ab.txt in the history (commit 53ee955); profiles/r06_cfg5_xcd_order_ab.txt): the camera level's queue is already in screen-tile order, which
// no key of this kind beats (level 0: 9.5 -> 10.6 ms ordered), and the deeper levels gain 0.5 ms of 13 for 0.7 ms of key + scatter:
// what the bounce rays fetch is decided where they END, which no order of their starts can know.  Kept as a bit-identical A/B
// (SDFHIP_PT_SORT=R, SDFHIP_PT_SORT_FROM=first level).
// ROUND 6 (profiles/r06_cfg5_xcd_order_ab.txt): the order alone moves no byte because every XCD still takes every eighth chunk of it;
// with each XCD walking a CONTIGUOUS eighth (SDFHIP_PT_SORT_XCD=1, k_pt_bounce) the bounce kernels read 77 GB instead of 113 per frame
// and the lines summed over the XCDs halve -- and the frame is 10 % SLOWER (key + scatter 1.4 ms; the levels gain 0.9 ms, level 0 loses
// 1.1 through the permutation): the levels sit at the latency of their dependent lookups, not at the bandwidth roof.  Dropped again.
// The bounce levels are bound by the 64-byte sectors their incoherent rays fetch.  The queues hold the hits in
// the order the waves of the level before pushed them; here a level's entries get a KEY -- where the ray starts and which way it
// will go: the bounce direction is a function of the entry alone (the hit's normal from its cursor's cell, the counter-based RNG) --
// and a permutation that orders them by it, so that the 64 rays of a wave, and the waves in flight, walk the same blocks of the
// grid.  The entries stay where they are (a gathered read through the permutation); results do not depend on the order of the
// queue (every path writes its own slots of pt_e / pt_t / pt_n), so the frame is bit for bit the unsorted pipeline's.
//   k_pt_key      one lane per entry: the key -> pt_key[entry], counts per key -> pt_hist (an LDS histogram per workgroup)
//   k_pt_scan     one workgroup: exclusive prefix of the counts -> the keys' first slots; pt_hist[keys] = the entry total
//   k_pt_scatter  the same entry -> workgroup map again: a workgroup claims its share of every key's slots with ONE atomic per key
//                 and hands them to its entries through LDS counters -> pt_perm[slot] = entry
constexpr int PT_SORT_MAX_BITS = 3;                       // 8^3 regions x 8 octants = 4096 keys: two 16 KB LDS arrays
constexpr int PT_SORT_THREADS = 256;
__device__ __forceinline__ uint32_t pt_morton3(uint32_t x, uint32_t y, uint32_t z, int bits)
{
    uint32_t m = 0;
    for (int b = 0; b < bits; b++) m |= (((x >> b) & 1u) | (((y >> b) & 1u) << 1) | (((z >> b) & 1u) << 2)) << (3 * b);
    return m;
}
template <int CUR>
__global__ __launch_bounds__(PT_SORT_THREADS) void k_pt_key(RenderParams P)
{
    typedef typename ScatterCursorOf<CUR, false>::type CursorT;
    __shared__ uint32_t hist[8u << (3 * PT_SORT_MAX_BITS)];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, b = P.pt_level, qin = b & 1u;
    const int R = (int)P.pt_sort_bits;
    const uint32_t nkeys = 8u << (3 * R);
    for (uint32_t k = tid; k < nkeys; k += PT_SORT_THREADS) hist[k] = 0u;
    __syncthreads();
    const FrameInfo &I = P.frames[0];
    const size_t total = (size_t)HIT_QUEUES * P.pt_cap;
    const float4 *Q = P.pt_q[qin];
    const PtChunks C(P, qin, lane);
    constexpr uint32_t WAVES = PT_SORT_THREADS / 64;
    for (uint32_t t = blockIdx.x * WAVES + wave; t < C.nchunks; t += gridDim.x * WAVES) {
        const uint32_t e = C.entry(P, t, lane);
        if (e == 0xFFFFFFFFu) continue;
        const float4 a = nt_load(&Q[e]), k = nt_load(&Q[total + e]), d = nt_load(&Q[2 * total + e]);
        // the vertex's normal and outgoing direction, as k_pt_bounce computes them (only their signs are used here)
        CursorT c;
        c.unpack(pt_unpack_cursor(make_uint2(__float_as_uint(k.x), __float_as_uint(k.y))),
                 CursorT::units_shift(P.top_level + (CUR == CUR_STACK_SPLIT ? P.fine_bits : 0)));
        c.v0 = __float_as_uint(k.z); c.v1 = __float_as_uint(k.w);
        float lx = I.lightx - a.x, ly = I.lighty - a.y, lz = I.lightz - a.z;
        const float rl = 1.0f / sqrtf(dot3(lx, ly, lz, lx, ly, lz));
        const float px = __builtin_fmaf(lx * rl, I.margin, a.x), py = __builtin_fmaf(ly * rl, I.margin, a.y), pz = __builtin_fmaf(lz * rl, I.margin, a.z);
        uint32_t octant = 0;
        if (b < P.pt_bounces) {
            float g0, g1, g2;
            gradient(c.cell(), px, py, pz, g0, g1, g2);
            const float rg = 1.0f / sqrtf(dot3(g0, g1, g2, g0, g1, g2));
            float n0 = g0 * rg, n1 = g1 * rg, n2 = g2 * rg;
            if (dot3(n0, n1, n2, d.y, d.z, d.w) > 0.0f) { n0 = -n0; n1 = -n1; n2 = -n2; }
            const uint32_t pid = __float_as_uint(d.x), pix = pid / P.pt_spp, s = pid - pix * P.pt_spp;
            const uint32_t y_l = pix / P.width, x_l = pix - y_l * P.width;
            const uint32_t p = global_row(P, y_l) * P.width + x_l;
            float u0 = n0, u1 = n1, u2 = n2, qq1 = 1.0f;
            for (uint32_t a2 = 0; a2 < 8; a2++) {
                const float c0 = rnd(P.pt_seed, p, s, b + 1, 3 * a2) * 2.0f - 1.0f;
                const float c1 = rnd(P.pt_seed, p, s, b + 1, 3 * a2 + 1) * 2.0f - 1.0f;
                const float c2 = rnd(P.pt_seed, p, s, b + 1, 3 * a2 + 2) * 2.0f - 1.0f;
                const float qq = dot3(c0, c1, c2, c0, c1, c2);
                if (qq <= 1.0f && qq > 1e-12f) { u0 = c0; u1 = c1; u2 = c2; qq1 = qq; break; }
            }
            const float ru = 1.0f / sqrtf(qq1);
            const float d0 = __builtin_fmaf(u0, ru, n0), d1 = __builtin_fmaf(u1, ru, n1), d2 = __builtin_fmaf(u2, ru, n2);
            octant = (d0 < 0.0f ? 1u : 0u) | (d1 < 0.0f ? 2u : 0u) | (d2 < 0.0f ? 4u : 0u);
        }
        const float scale = (float)(1u << R), top = scale - 1.0f;
        const uint32_t rx = (uint32_t)__builtin_amdgcn_fmed3f(px * scale, 0.0f, top), ry = (uint32_t)__builtin_amdgcn_fmed3f(py * scale, 0.0f, top),
                       rz = (uint32_t)__builtin_amdgcn_fmed3f(pz * scale, 0.0f, top);
        const uint32_t key = (pt_morton3(rx, ry, rz, R) << 3) | octant;
        P.pt_key[e] = (uint16_t)key;
        atomicAdd(&hist[key], 1u);
    }
    __syncthreads();
    for (uint32_t k = tid; k < nkeys; k += PT_SORT_THREADS) { const uint32_t v = hist[k]; if (v) atomicAdd(&P.pt_hist[k], v); }
}

// exclusive prefix of the keys' counts in place (one workgroup); the entry total goes to pt_hist[nkeys]
template <int = 0>      // (a template so that the header may be included by several translation units)
__global__ __launch_bounds__(1024) void k_pt_scan(uint32_t *__restrict__ v, uint32_t n)
{
    __shared__ uint32_t part[1024];
    const uint32_t per = (n + 1023u) / 1024u, lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += v[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {
        const uint32_t add = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t i = lo; i < hi; i++) { const uint32_t x = v[i]; v[i] = run; run += x; }
    if (threadIdx.x == 1023u) v[n] = part[1023];
}

template <int = 0>
__global__ __launch_bounds__(PT_SORT_THREADS) void k_pt_scatter(RenderParams P)
{
    __shared__ uint32_t cnt[8u << (3 * PT_SORT_MAX_BITS)], base[8u << (3 * PT_SORT_MAX_BITS)];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, qin = P.pt_level & 1u;
    const uint32_t nkeys = 8u << (3 * P.pt_sort_bits);
    for (uint32_t k = tid; k < nkeys; k += PT_SORT_THREADS) cnt[k] = 0u;
    __syncthreads();
    const PtChunks C(P, qin, lane);
    constexpr uint32_t WAVES = PT_SORT_THREADS / 64;
    for (uint32_t t = blockIdx.x * WAVES + wave; t < C.nchunks; t += gridDim.x * WAVES) {       // this workgroup's entries by key
        const uint32_t e = C.entry(P, t, lane);
        if (e != 0xFFFFFFFFu) atomicAdd(&cnt[P.pt_key[e]], 1u);
    }
    __syncthreads();
    for (uint32_t k = tid; k < nkeys; k += PT_SORT_THREADS) {                                     // its share of every key's slots
        const uint32_t v = cnt[k];
        base[k] = v ? atomicAdd(&P.pt_hist[k], v) : 0u;
        cnt[k] = 0u;
    }
    __syncthreads();
    for (uint32_t t = blockIdx.x * WAVES + wave; t < C.nchunks; t += gridDim.x * WAVES) {
        const uint32_t e = C.entry(P, t, lane);
        if (e == 0xFFFFFFFFu) continue;
        const uint32_t key = P.pt_key[e];
        P.pt_perm[base[key] + atomicAdd(&cnt[key], 1u)] = e;
    }
}
#endif

template <int CUR, bool COUNT>
__global__ __launch_bounds__(64, PATH_WAVES_PER_SIMD) void k_pt_bounce(RenderParams P)
{
    typedef typename ScatterCursorOf<CUR, COUNT>::type CursorT;
    const uint32_t lane = threadIdx.x, b = P.pt_level, qin = b & 1u, qout = qin ^ 1u;
    FrameInfo I = P.frames[0];
    asm volatile("" : "+s"(I.margin), "+s"(I.margin2), "+s"(I.limit));
    const float margin = I.margin;
    const size_t npx = (size_t)P.nrows_out * P.width, total = (size_t)HIT_QUEUES * P.pt_cap;
    const PtChunks C(P, qin, lane);
#ifdef SDFHIP_EXPERIMENTS
    // ordered (P.pt_sort_bits, an A/B): chunk t = slots 64 t .. of the permutation; else the queue's own chunks
    const uint32_t n_sorted = P.pt_sort_bits ? P.pt_hist[8u << (3 * P.pt_sort_bits)] : 0u;
    const uint32_t nchunks = P.pt_sort_bits ? (n_sorted + 63u) >> 6 : C.nchunks;
    // ... and (P.pt_sort_xcd, round 6's attempt at the bounce levels' refetch factor) every XCD walks a contiguous eighth of that order
    // front to back instead of every eighth chunk of all of it: the workgroups that share an L2 work in the same part of the scene
    // (blockIdx.x & 7 names the workgroups that share an XCD under round-robin placement: speed only)
    const bool by_xcd = P.pt_sort_bits && P.pt_sort_xcd;
    const uint32_t per_xcd = (nchunks + 7u) >> 3, t_first = by_xcd ? (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t t_end = by_xcd ? min(((blockIdx.x & 7u) + 1u) * per_xcd, nchunks) : nchunks, t_step = by_xcd ? gridDim.x >> 3 : gridDim.x;
#else
    const uint32_t nchunks = C.nchunks;
    const uint32_t t_first = blockIdx.x, t_end = nchunks, t_step = gridDim.x;
#endif
    unsigned long long cn = 0, cs = 0, cr = 0, cl = 0, ch = 0;     // ch: queue entries this level took (counting builds: sdfhip_stats.n_hits)
    const float4 *Q = P.pt_q[qin];
    for (uint32_t t = t_first; t < t_end; t += t_step) {
        uint32_t e32;
#ifdef SDFHIP_EXPERIMENTS
        if (P.pt_sort_bits) { const uint32_t slot = t * 64u + lane; e32 = slot < n_sorted ? P.pt_perm[slot] : 0xFFFFFFFFu; }
        else
#endif
        e32 = C.entry(P, t, lane);
        const bool have = e32 != 0xFFFFFFFFu;
        RayState r;
        CursorT c;
        c.loads = 0;
        uint32_t pid = 0;
        float ux = 0, uy = 0, uz = 1, T = 0;
        bool next = false, escaped = false;       // a next segment was marched; it escaped
        if (have) {
            const size_t e = e32;
            const float4 a = nt_load(&Q[e]), k = nt_load(&Q[total + e]), d = nt_load(&Q[2 * total + e]);
            r.px = a.x; r.py = a.y; r.pz = a.z; r.prox = a.w;
            c.unpack(pt_unpack_cursor(make_uint2(__float_as_uint(k.x), __float_as_uint(k.y))),
                     CursorT::units_shift(P.top_level + (CUR == CUR_STACK_SPLIT ? P.fine_bits : 0)));
            c.v0 = __float_as_uint(k.z); c.v1 = __float_as_uint(k.w); pid = __float_as_uint(d.x);
            ux = d.y; uy = d.z; uz = d.w;
            T = 1.0f;
            for (uint32_t lv = 0; lv < b; lv++) T *= P.pt_albedo;     // what the levels before this one multiplied onto 1, in their order
            const uint32_t pix = pid / P.pt_spp, s = pid - pix * P.pt_spp;
            const size_t o = (size_t)s * npx + pix;
            // shade (Compute.hlsl:205-213)
            float lx = I.lightx - r.px, ly = I.lighty - r.py, lz = I.lightz - r.pz;
            const float rl = 1.0f / sqrtf(dot3(lx, ly, lz, lx, ly, lz));
            const float L0 = lx * rl, L1 = ly * rl, L2 = lz * rl;
            r.px = __builtin_fmaf(L0, margin, r.px); r.py = __builtin_fmaf(L1, margin, r.py); r.pz = __builtin_fmaf(L2, margin, r.pz);
            float g0, g1, g2;
            gradient(c.cell(), r.px, r.py, r.pz, g0, g1, g2);
            const float rg = 1.0f / sqrtf(dot3(g0, g1, g2, g0, g1, g2));
            float n0 = g0 * rg, n1 = g1 * rg, n2 = g2 * rg;
            const float angle = dot3(L0, L1, L2, n0, n1, n2);
            const float hx = r.px, hy = r.py, hz = r.pz;      // the hit point: the bounce leaves from here
            float e_light = 0.0f;
            uint32_t shadow_steps = 0;
            if (!(angle < 0.0f)) {
                // shadow march (Compute.hlsl:213-230)
                lx = I.lightx - r.px; ly = I.lighty - r.py; lz = I.lightz - r.pz;
                const float dist = sqrtf(dot3(lx, ly, lz, lx, ly, lz)) / 2.0f;
                r.dx = L0; r.dy = L1; r.dz = L2; r.n = 0; r.phase = PH_SHADOW;
                if (COUNT) cr += 1;
                r.dist = dist;
                const bool lit = shadow_march<COUNT>(P, I, r, c, cn, cs);
                shadow_steps = (uint32_t)r.n;
                if (lit) e_light = T * (P.pt_albedo * (angle / (dist * dist) * I.k_strength));
            }
            __builtin_nontemporal_store(e_light, &P.pt_e[(size_t)b * P.pt_spp * npx + o]);
            uint32_t nsteps = (__builtin_nontemporal_load(&P.pt_n[o]) & 0xFFFFu) + shadow_steps;
            if (b < P.pt_bounces) {
                // diffuse bounce (o_pixel_pt): the normal facing the incoming ray, a direction by rejection in the cube
                if (dot3(n0, n1, n2, ux, uy, uz) > 0.0f) { n0 = -n0; n1 = -n1; n2 = -n2; }
                const uint32_t y_l = pix / P.width, x_l = pix - y_l * P.width;
                const uint32_t p = global_row(P, y_l) * P.width + x_l;          // the frame's pixel index seeds the RNG
                float u0 = n0, u1 = n1, u2 = n2, qq1 = 1.0f;
                for (uint32_t a2 = 0; a2 < 8; a2++) {
                    const float c0 = rnd(P.pt_seed, p, s, b + 1, 3 * a2) * 2.0f - 1.0f;
                    const float c1 = rnd(P.pt_seed, p, s, b + 1, 3 * a2 + 1) * 2.0f - 1.0f;
                    const float c2 = rnd(P.pt_seed, p, s, b + 1, 3 * a2 + 2) * 2.0f - 1.0f;
                    const float qq = dot3(c0, c1, c2, c0, c1, c2);
                    if (qq <= 1.0f && qq > 1e-12f) { u0 = c0; u1 = c1; u2 = c2; qq1 = qq; break; }
                }
                const float ru = 1.0f / sqrtf(qq1);
                float d0 = __builtin_fmaf(u0, ru, n0), d1 = __builtin_fmaf(u1, ru, n1), d2 = __builtin_fmaf(u2, ru, n2);
                float qd = dot3(d0, d1, d2, d0, d1, d2);
                if (!(qd > 1e-12f)) { d0 = n0; d1 = n1; d2 = n2; qd = dot3(n0, n1, n2, n0, n1, n2); }
                const float rd = 1.0f / sqrtf(qd);
                ux = d0 * rd; uy = d1 * rd; uz = d2 * rd;
                const float off = margin * 4.0f;
                r.px = __builtin_fmaf(n0, off, hx); r.py = __builtin_fmaf(n1, off, hy); r.pz = __builtin_fmaf(n2, off, hz);
                T *= P.pt_albedo;
                // the next segment
                r.dx = ux; r.dy = uy; r.dz = uz; r.prox = 1.0f; r.n = 0; r.phase = PH_PRIMARY;
                next = true;
                escaped = pt_march<COUNT>(P, I, r, c, cn, cs);
                nsteps += (uint32_t)r.n;
                if (escaped) __builtin_nontemporal_store(T, &P.pt_t[o]);
            }
            __builtin_nontemporal_store(nsteps | ((b + 1u) << 16), &P.pt_n[o]);
            if (COUNT) { cl += c.loads; ch += 1; }
        }
        pt_push(P, qout, blockIdx.x & (HIT_QUEUES - 1u), have && next && !escaped, lane, r.px, r.py, r.pz, r.prox, c, pid, ux, uy, uz);
    }
    if (COUNT) flush_counters(P, cn, cs, 0, cr, cl, ch);
}

#ifdef SDFHIP_EXPERIMENTS
// ---- LABORATORY (round 6): a bounce level with LANE REFILL ------------------------------------------------------------------------
// k_pt_bounce gives a lane one queue entry and the wave waits for its slowest lane twice (the shadow march, then the next segment):
// lanes are on in 44 % of the VALU thread-cycles.  Here a wave is PERSISTENT over its chunks and a lane that has finished its
// entry takes the next one (its wave's next unread queue slot) while the others march on: one loop, every active lane makes one
// march step per iteration (of its shadow ray or of its segment), and the code between two marches -- loading an entry and shading
// it, the light term and the bounce, the stores and the push -- runs under wave-uniform "does any lane need it" tests.
// Every path's arithmetic is k_pt_bounce's, statement for statement (the cursor goes from the shadow march into the segment, as
// there); only the ORDER of the pushes into the next queue differs, which no result depends on.  SDFHIP_PT_REFILL=1.
// MEASURED: 66 % SLOWER (scripts/pt_refill_ab.py, profiles/r06_cfg5_refill_ab.txt: 20.8-21.1 -> 34.7-35.5 ms per cfg-5 frame, frames and
// counters identical; same 72 VGPRs, occupancy 7).  What k_march taught in round 2 holds for incoherent rays too: the lane state
// machine pays its phase tests and ballots in every iteration and its transition code whenever ANY lane needs it, and that costs more
// than the idle lanes of two tight loops.  Kept as the record of the attempt.
template <int CUR, bool COUNT>
__global__ __launch_bounds__(64, PATH_WAVES_PER_SIMD) void k_pt_bounce_refill(RenderParams P)
{
    typedef typename ScatterCursorOf<CUR, COUNT>::type CursorT;
    enum { ST_IDLE = 0, ST_SHADOW = 1, ST_SEGMENT = 2 };
    const uint32_t lane = threadIdx.x, b = P.pt_level, qin = b & 1u, qout = qin ^ 1u;
    FrameInfo I = P.frames[0];
    asm volatile("" : "+s"(I.margin), "+s"(I.margin2), "+s"(I.limit));
    const float margin = I.margin;
    const size_t npx = (size_t)P.nrows_out * P.width, total = (size_t)HIT_QUEUES * P.pt_cap;
    const PtChunks C(P, qin, lane);
    const uint32_t nchunks = C.nchunks;
    unsigned long long cn = 0, cs = 0, cr = 0, cl = 0, ch = 0;
    const float4 *Q = P.pt_q[qin];
    float Tb = 1.0f;
    for (uint32_t lv = 0; lv < b; lv++) Tb *= P.pt_albedo;             // what the levels before this one multiplied onto 1, in their order
    // the wave's reading position: slot `pos` of chunk `t` (64 slots per chunk; a chunk's last slots may hold no entry)
    uint32_t t = blockIdx.x, pos = 0;
    // lane state
    int st = ST_IDLE;
    RayState r;
    CursorT c;
    c.loads = 0;
    r.px = r.py = r.pz = 0.0f; r.dx = r.dy = r.dz = 0.0f; r.prox = 1.0f; r.angle = 0.0f; r.dist = 1.0f; r.n = 0; r.base = 0; r.phase = PH_PRIMARY;
    uint32_t pid = 0, nsteps = 0;
    size_t o = 0;
    float ux = 0, uy = 0, uz = 1, T = 0, angle = 0, hx = 0, hy = 0, hz = 0, n0 = 0, n1 = 0, n2 = 1;
    bool flip = false;
    for (;;) {
        // ---- (A) idle lanes take the wave's next unread slots ---------------------------------------------------------------
        bool fresh = false;
        {
            unsigned long long idle = __ballot(st == ST_IDLE);
            while (idle && t < nchunks) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const uint32_t room = 64u - pos, want = (uint32_t)__popcll(idle), take = want < room ? want : room;
                const bool mine = st == ST_IDLE && !fresh && rank < take;
                const uint32_t e32 = C.entry(P, t, mine ? pos + rank : 0u);          // (wave-wide inside: every lane calls it)
                if (mine && e32 != 0xFFFFFFFFu) {
                    const size_t e = e32;
                    const float4 a = nt_load(&Q[e]), k = nt_load(&Q[total + e]), d = nt_load(&Q[2 * total + e]);
                    r.px = a.x; r.py = a.y; r.pz = a.z; r.prox = a.w;
                    c.loads = 0;
                    c.unpack(pt_unpack_cursor(make_uint2(__float_as_uint(k.x), __float_as_uint(k.y))),
                             CursorT::units_shift(P.top_level + (CUR == CUR_STACK_SPLIT ? P.fine_bits : 0)));
                    c.v0 = __float_as_uint(k.z); c.v1 = __float_as_uint(k.w); pid = __float_as_uint(d.x);
                    ux = d.y; uy = d.z; uz = d.w;
                    fresh = true;
                    if (COUNT) ch += 1;
                }
                pos += take;
                if (pos == 64u) { pos = 0; t += gridDim.x; }
                idle = __ballot(st == ST_IDLE && !fresh);
            }
        }
        // ---- (B) a fresh entry: shade it (Compute.hlsl:205-213), start its shadow ray -----------------------------------------
        bool to_bounce = false;                       // this lane's shadow march is over (or there was none): light term, bounce
        if (__ballot(fresh)) {
            if (fresh) {
                T = Tb;
                const uint32_t pix = pid / P.pt_spp, s = pid - pix * P.pt_spp;
                o = (size_t)s * npx + pix;
                float lx = I.lightx - r.px, ly = I.lighty - r.py, lz = I.lightz - r.pz;
                const float rl = 1.0f / sqrtf(dot3(lx, ly, lz, lx, ly, lz));
                const float L0 = lx * rl, L1 = ly * rl, L2 = lz * rl;
                r.px = __builtin_fmaf(L0, margin, r.px); r.py = __builtin_fmaf(L1, margin, r.py); r.pz = __builtin_fmaf(L2, margin, r.pz);
                float g0, g1, g2;
                gradient(c.cell(), r.px, r.py, r.pz, g0, g1, g2);
                const float rg = 1.0f / sqrtf(dot3(g0, g1, g2, g0, g1, g2));
                n0 = g0 * rg; n1 = g1 * rg; n2 = g2 * rg;
                angle = dot3(L0, L1, L2, n0, n1, n2);
                flip = dot3(n0, n1, n2, ux, uy, uz) > 0.0f;        // (decided now: the incoming direction is overwritten below)
                hx = r.px; hy = r.py; hz = r.pz;
                nsteps = 0;
                r.n = 0;
                if (!(angle < 0.0f)) {
                    lx = I.lightx - r.px; ly = I.lighty - r.py; lz = I.lightz - r.pz;
                    r.dist = sqrtf(dot3(lx, ly, lz, lx, ly, lz)) / 2.0f;
                    r.dx = L0; r.dy = L1; r.dz = L2; r.phase = PH_SHADOW;
                    if (COUNT) cr += 1;
                    st = ST_SHADOW;
                } else {
                    st = ST_SHADOW; to_bounce = true;              // no shadow ray: e_light = 0, zero shadow steps
                }
            }
        }
        // ---- (C) between two march steps: is this lane's march over? --------------------------------------------------------
        bool lit = false;
        if (st == ST_SHADOW && !to_bounce) {                       // Compute.hlsl:214-223, as shadow_march's loop head
            const bool header = r.n < 40 && r.prox > -I.margin;
            const float lo = __builtin_fminf(__builtin_fminf(r.px, r.py), r.pz), hi = __builtin_fmaxf(__builtin_fmaxf(r.px, r.py), r.pz);
            const bool at_light = r.prox > r.dist || lo < 0.0f || hi > 1.0f;
            bool go = header && !at_light;
            if (go && r.prox < I.margin) {
                float gx, gy, gz;
                gradient(cell_of(c, P), r.px, r.py, r.pz, gx, gy, gz);
                go = !(dot3(gx, gy, gz, r.dx, r.dy, r.dz) < 0.0f);
            }
            if (!go) { to_bounce = true; lit = header && at_light; }
        }
        bool seg_over = false, escaped = false;
        if (__ballot(to_bounce)) {
            if (to_bounce) {
                const bool had_ray = !(angle < 0.0f);
                const uint32_t shadow_steps = had_ray ? (uint32_t)r.n : 0u;
                float e_light = 0.0f;
                if (had_ray && lit) e_light = T * (P.pt_albedo * (angle / (r.dist * r.dist) * I.k_strength));
                __builtin_nontemporal_store(e_light, &P.pt_e[(size_t)b * P.pt_spp * npx + o]);
                nsteps = (__builtin_nontemporal_load(&P.pt_n[o]) & 0xFFFFu) + shadow_steps;
                if (b < P.pt_bounces) {
                    // diffuse bounce (o_pixel_pt): the normal facing the incoming ray, a direction by rejection in the cube
                    if (flip) { n0 = -n0; n1 = -n1; n2 = -n2; }
                    const uint32_t pix = pid / P.pt_spp, s = pid - pix * P.pt_spp;
                    const uint32_t y_l = pix / P.width, x_l = pix - y_l * P.width;
                    const uint32_t p = global_row(P, y_l) * P.width + x_l;          // the frame's pixel index seeds the RNG
                    float u0 = n0, u1 = n1, u2 = n2, qq1 = 1.0f;
                    for (uint32_t a2 = 0; a2 < 8; a2++) {
                        const float c0 = rnd(P.pt_seed, p, s, b + 1, 3 * a2) * 2.0f - 1.0f;
                        const float c1 = rnd(P.pt_seed, p, s, b + 1, 3 * a2 + 1) * 2.0f - 1.0f;
                        const float c2 = rnd(P.pt_seed, p, s, b + 1, 3 * a2 + 2) * 2.0f - 1.0f;
                        const float qq = dot3(c0, c1, c2, c0, c1, c2);
                        if (qq <= 1.0f && qq > 1e-12f) { u0 = c0; u1 = c1; u2 = c2; qq1 = qq; break; }
                    }
                    const float ru = 1.0f / sqrtf(qq1);
                    float d0 = __builtin_fmaf(u0, ru, n0), d1 = __builtin_fmaf(u1, ru, n1), d2 = __builtin_fmaf(u2, ru, n2);
                    float qd = dot3(d0, d1, d2, d0, d1, d2);
                    if (!(qd > 1e-12f)) { d0 = n0; d1 = n1; d2 = n2; qd = dot3(n0, n1, n2, n0, n1, n2); }
                    const float rd = 1.0f / sqrtf(qd);
                    ux = d0 * rd; uy = d1 * rd; uz = d2 * rd;
                    const float off = margin * 4.0f;
                    r.px = __builtin_fmaf(n0, off, hx); r.py = __builtin_fmaf(n1, off, hy); r.pz = __builtin_fmaf(n2, off, hz);
                    T *= P.pt_albedo;
                    r.dx = ux; r.dy = uy; r.dz = uz; r.prox = 1.0f; r.n = 0; r.phase = PH_PRIMARY;
                    st = ST_SEGMENT;
                } else {
                    seg_over = true;                               // the last level: no next segment, nothing to push
                    st = ST_IDLE;
                    __builtin_nontemporal_store(nsteps | ((b + 1u) << 16), &P.pt_n[o]);
                    if (COUNT) cl += c.loads;
                }
            }
        }
        bool push = false;
        if (st == ST_SEGMENT) {                                    // pt_march's loop head
            const bool marching = (r.prox > I.margin2 || r.prox < 0.0f) && r.n < 100;
            const bool go_on = marching && !(dot3(r.px, r.py, r.pz, r.px, r.py, r.pz) > I.limit);
            if (!go_on) {
                escaped = marching;
                nsteps += (uint32_t)r.n;
                if (escaped) __builtin_nontemporal_store(T, &P.pt_t[o]);
                __builtin_nontemporal_store(nsteps | ((b + 1u) << 16), &P.pt_n[o]);
                if (COUNT) cl += c.loads;
                push = !escaped;
                seg_over = true;
                st = ST_IDLE;
            }
        }
        if (__ballot(seg_over))
            pt_push(P, qout, blockIdx.x & (HIT_QUEUES - 1u), push, lane, r.px, r.py, r.pz, r.prox, c, pid, ux, uy, uz);
        // ---- (D) done? else one march step of every lane that is marching ------------------------------------------------------
        const unsigned long long active = __ballot(st != ST_IDLE);
        if (!active) {
            if (t >= nchunks) break;
            continue;                                              // every lane idle, entries left: refill
        }
        if (st != ST_IDLE) {
            uint32_t reads = march_step(P, I, r, c, nullptr, 0);
            if (COUNT) { cn += reads; cs += 1; }
        }
    }
    if (COUNT) flush_counters(P, cn, cs, 0, cr, cl, ch);
}
#endif

// Per pixel: o_pixel_pt's accumulation over samples and bounces, in its order; alpha = the march steps of all paths.
template <bool COUNT>
__global__ __launch_bounds__(256) void k_pt_resolve(RenderParams P)
{
    const size_t npx = (size_t)P.nrows_out * P.width;
    unsigned long long ct = 0;
    for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npx; pix += (size_t)gridDim.x * blockDim.x) {
        const uint32_t yl = (uint32_t)(pix / P.width);
        if (global_row(P, yl) >= P.height) continue;          // a padding row of the last band: not a pixel
        float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
        uint32_t steps = 0;
        for (uint32_t s = 0; s < P.pt_spp; s++) {
            const size_t o = (size_t)s * npx + pix;
            const uint32_t n = __builtin_nontemporal_load(&P.pt_n[o]), nv = n >> 16;
            steps += n & 0xFFFFu;
            for (uint32_t b = 0; b < nv; b++) {
                const float e = __builtin_nontemporal_load(&P.pt_e[(size_t)b * P.pt_spp * npx + o]);
                acc0 += e; acc1 += e; acc2 += e;
            }
            const float T = __builtin_nontemporal_load(&P.pt_t[o]);
            acc0 = __builtin_fmaf(T, 0.005f, acc0); acc1 = __builtin_fmaf(T, 0.01f, acc1); acc2 = __builtin_fmaf(T, 0.2f, acc2);
        }
        const float inv = (float)P.pt_spp;
        P.out[pix] = make_float4(acc0 / inv, acc1 / inv, acc2 / inv, (float)steps);
        if (COUNT) ct += steps;
    }
    if (COUNT) flush_counters(P, 0, 0, ct, 0, 0);
}

}  // namespace sdfhip
