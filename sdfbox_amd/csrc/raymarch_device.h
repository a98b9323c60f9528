// Device-side building blocks of the ray-march kernels (gfx950 only).
//
// Arithmetic contract (DESIGN.md "Numerics"; identical to oracle/sdf_oracle.c):
// every fp32 operation is individually rounded in the order written (the
// translation unit is built with -ffp-contract=off: the compiler fuses nothing
// on its own), IEEE divide and sqrt.  The multiply-adds HLSL compiles to
// mad / dp3 are explicit fused __builtin_fmaf on both sides:
//     lerp(a,b,t) = fma(t, b - a, a)      dot = fma(az,bz, fma(ay,by, ax*bx))
//     pos += dir*s = fma(dir, s, pos)     normalize(v) = v * (1 / sqrt(dot(v,v)))
//
// Exact rewrites relative to the HLSL text, all bit-identical:
//   * x / box.scale  ->  x * box.inv   (scale is a power of two, inv = 1/scale:
//     multiplying by an exact power of two rounds exactly like dividing by it);
//   * (int) saturate(v * 2)  ->  v >= 0.5f   (v*2 is exact; saturate maps NaN
//     to 0, and NaN >= 0.5f is false);
//   * (float)b / 255.0f  ->  fma(b, fl(1/255), b * fl(1/255 - fl(1/255))) (unorm8 below,
//     checked for all 256 bytes);
//   * the cursor-stack kernel walks the tree on integer cell coordinates
//     (find_stack below states why that visits the same cells).
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


// The camera part of the Info block, unpacked (Logic.cs:407-420).
struct FrameInfo {
    float h0x, h0y, h0z, h1x, h1y, h1z, h2x, h2y, h2z;  // heading rows
    float posx, posy, posz, margin;
    float screen_w, screen_h, limit;
    float lightx, lighty, lightz;
    float fov, k_strength;     // k_strength = exp2f(strength) - 1, evaluated on the host
    float margin2;             // margin * 2 (exact), Compute.hlsl:194
    float half_aspect;         // screen_w / screen_h * 0.5f (Compute.hlsl:165), the same two IEEE operations, once per frame on the host
};
constexpr int MAX_BATCH = 8;   // frames one k_plain launch can render (grid.y)
constexpr int MAX_BAND_LIST = 512;   // bands one launch can be handed as an explicit list
constexpr int INLINE_BAND_LIST = 64; // ... of which this many travel in the kernel arguments; a longer list is read from device memory

struct TopCell { uint32_t level, v0, v1; int32_t children; };   // a cell of the top grid, see below (cursor-stack kernels)
// the grid as find() sees it: dense cells of level `level`; for a split grid, blocks of 8^fine_bits finer cells
// fine_order 1: within a block the cells are stored 2x2x2 sub-cube by sub-cube (the 8 cells of a sub-cube share a 128-byte
// line), not in x-y-z order: for the kernels whose every lookup is a cache miss (fine_cell_index)
// (experiments build: d4 / recs = the same grid again as a dense array of 4-byte words + 64-byte sample records of the non-flat
// leaves, for the default kernel's loop through CursorFF, lab_device.h; or null)
struct GridRef {
    const TopCell *top; const TopCell *fine; int level; int fine_bits; int fine_order;
#ifdef SDFHIP_EXPERIMENTS
    const uint32_t *d4 = nullptr; const uint4 *recs = nullptr;
    // sdfhip_debug_touch_*: one bit per 128-byte line of `top` / `fine` and XCD ([8][words]), set by the COUNTING kernels' lookups
    uint32_t *touch_top = nullptr, *touch_fine = nullptr; uint32_t touch_top_words = 0, touch_fine_words = 0;
#endif
};
#ifdef SDFHIP_EXPERIMENTS
// The line of cell `cell` (16-byte cells, 8 to a 128-byte line: the arrays are hipMalloc'ed, so line-aligned) is marked in the
// bitmap of the XCD this wave runs on (HW_REG_XCC_ID, hwreg 20 bits 3:0).  Counting builds only: what a frame MUST fetch is
// the distinct lines it touches -- chip-wide with one ideal cache, per XCD with eight ideal L2s that share nothing.
__device__ __forceinline__ void touch_line(uint32_t *bits, uint32_t words, uint32_t cell)
{
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;
    const uint32_t line = cell >> 3;
    uint32_t *w = bits + (size_t)xcc * words + (line >> 5);
    const uint32_t bit = 1u << (line & 31u);
    if (!(*reinterpret_cast<volatile uint32_t *>(w) & bit)) atomicOr(w, bit);
}
#endif
__host__ __device__ __forceinline__ uint32_t fine_cell_index(uint32_t x, uint32_t y, uint32_t z, int FB, int order)
{
    if (order == 0) return x | (y << FB) | (z << (2 * FB));
    const int H = FB - 1;                                  // sub-cube coordinates have FB - 1 bits
    return ((((x >> 1) | ((y >> 1) << H) | ((z >> 1) << (2 * H))) << 3) | (x & 1u) | ((y & 1u) << 1) | ((z & 1u) << 2));
}

// Kernel parameters: scene, frame geometry, and the camera block of every frame of the launch.
struct RenderParams {
    const NodeRec *nodes;
    uint32_t n_nodes;
    const TopCell *top;        // top grid (cursor-stack kernels) of level top_level, or null
    int32_t top_level;
    // split grid (CUR_STACK_SPLIT kernels): a cell of `top` whose node is internal has level 15 and names,
    // in `children`, the first cell of a block of `fine`: the 8^fine_bits cells of the next fine_bits levels below it
    int32_t fine_bits;
    int32_t fine_order;        // see GridRef
    const TopCell *fine;
#ifdef SDFHIP_EXPERIMENTS
    const uint32_t *d4;        // the grid as 4-byte words (CursorFF): a flat leaf's distance, or where a non-flat leaf's sample record is; or null
    const uint4 *recs;         // ... the records, biased by the words' tag (see find_sample_ff)
    uint32_t *touch_top, *touch_fine;            // sdfhip_debug_touch_*: see GridRef (null: off)
    uint32_t touch_top_words, touch_fine_words;
#endif
    float4 *out;               // compact rows: nrows_out x width
    uint32_t width, height;    // full frame
    uint32_t band_rows, band_first, band_stride, nrows_out;
    uint32_t band_shift;       // log2(band_rows) when that is a power of two (a rank's 16-row bands: a shift, not a division, per pixel), else 32
    // n_band_list != 0: local band i is band band_list[i] of the frame (an explicit list instead of
    // every band_stride-th band: layouts that give the ranks unequal shares).  Up to INLINE_BAND_LIST bands travel
    // in the kernel arguments (read-only: the kernels index them there); a longer list is in device memory at band_ptr.
    uint32_t n_band_list;
    uint16_t band_list[INLINE_BAND_LIST];
    const uint16_t *band_ptr;
    uint32_t tiles_x, tiles_y, n_tiles;   // 8x8 (compact) or 16x16 (plain) tiles of the local rows
    // frames[f]: camera of frame f of a batched launch (k_plain: f = blockIdx.y, output at
    // out + f * nrows_out * width pixels; the other kernels render frames[0] only).  A rank's
    // share of a sharded frame is mostly the serial tail of its longest pixels; several frames
    // in one grid share that tail.  The kernels read these through a const reference into the
    // kernel-argument segment (scalar loads), never through a modified copy.
    uint32_t n_frames;
    FrameInfo frames[MAX_BATCH];
    unsigned long long *counters;  // [0] nodes [1] samples [2] steps [3] shadow rays [4] records / cells loaded [5] queued hits (COUNT builds)
    uint32_t *queue;           // tile queue head (compact kernels)
#ifdef SDFHIP_EXPERIMENTS
    uint32_t tile_order;       // k_plain: blockIdx -> tile mapping (tuning knob, flags bits 8..11)
#endif
    // fused display pass (DisplayFrag.hlsl): out_mode 0 = RGBA32F frame, 1 = gamma RGBA8,
    // 2 = step-count heat map RGBA8 (`out` aliased as one uint32 per pixel), 3 (experiments build) = wire pixels of round 1's
    // tile gather (per frame a float plane and a byte plane, 5 bytes per pixel)
    uint32_t out_mode;         // 4: the sparse wire share of sdfhip_render_sparse_device (`out` = the share of all frames of the launch)
    uint32_t sparse_cap;       // ... and the float slots it holds
    uint32_t sparse_base;      // ... and the value of its counter (header word 0) before this launch
    uint32_t sky8;             // the sky constant through the display pass (host-computed, alpha excluded)
    uint32_t out_host;         // `out` is page-locked HOST memory (sdfhip_render into a registered array): read by the HOST's launch code only,
                               // which picks the instantiation that stores the frame with plain stores (frame_store, raymarch_kernels.h)
    // path-traced mode (k_path): samples per pixel, diffuse bounces, RNG seed, albedo
    uint32_t pt_spp, pt_bounces, pt_seed;
    float pt_albedo;
    // SDFHIP_FLAG_COMPACT on a scene with a full-depth grid (k_march<..., QUEUE> -> k_shadow): the shadow rays of the waves that hold
    // fewer than hit_min of them, as records {position, prox | cursor | pixel, steps, values | direction to the light, Lambert term}
    // in HIT_QUEUES queues per frame of the launch: arrays [n_frames][HIT_QUEUES][hit_cap]; hit_ctl = two sets of
    // [MAX_BATCH][HIT_QUEUES] fill counts (a 128-byte line each); this launch pair uses set hit_set and leaves the other one zeroed
    float4 *hit_a;
    int4 *hit_b;
    uint4 *hit_c;
    float4 *hit_d;                 // direction to the light, Lambert term
    uint32_t *hit_ctl;
    uint32_t hit_cap, hit_set;
    uint32_t hit_min;              // a wave with at least this many shadow rays marches them itself (65: every ray is queued)
    // path-traced mode as a pipeline of kernels (k_pt_primary -> k_pt_bounce x (bounces + 1) -> k_pt_resolve):
    // two hit queues used alternately (a level's kernel reads one and fills the other), HIT_QUEUES sub-queues of
    // pt_cap entries each, an entry = four 16-byte records; per-path results [sample][pixel] that k_pt_resolve sums
    // in the oracle's order: pt_e[bounce] = the light a path's vertex of that bounce received (0 when unlit), pt_t =
    // the throughput it escaped to the sky with (0 when it did not), pt_n = its march steps | vertices << 16
    float4 *pt_q[2];
    uint32_t *pt_ctl;               // [2 queues][HIT_QUEUES] fill counts, a 128-byte line each
    uint32_t pt_cap, pt_level;      // entries per sub-queue; the bounce level this launch shades
    float *pt_e;                    // [bounces + 1][spp][pixels]
    float *pt_t;                    // [spp][pixels]
    uint32_t *pt_n;                 // [spp][pixels]
#ifdef SDFHIP_EXPERIMENTS
    // A/B (measured: no gain, profiles/r04_cfg5_sort_ab.txt in the history (commit 53ee955); profiles/r06_cfg5_xcd_order_ab.txt): bounce levels with pt_sort_bits = R > 0: before a level's kernel its queue entries are ordered by the key
    // (Morton code of the hit's region, 2^R regions per axis) << 3 | octant of the OUTGOING direction -- the rays a wave marches
    // start in the same part of the scene and head the same way (k_pt_key, k_pt_scatter): pt_key[entry] = the key, pt_perm[slot] =
    // the entry a lane takes, pt_hist = the keys' counts, then their running slots ([keys + 1]: the last word is the entry total)
    uint16_t *pt_key;
    uint32_t *pt_perm;
    uint32_t *pt_hist;
    uint32_t pt_sort_bits;
    uint32_t pt_sort_xcd;          // ... and each XCD walks a contiguous eighth of that order (SDFHIP_PT_SORT_XCD=1)
#endif
    // k_march, optional: tile_perm[b] = the tile workgroup b renders, as tile row << 16 | tile column (a permutation of the
    // default order: the previous frame's expensive tiles first; all ones = idle); tile_cost[tile] = march iterations the tile's wave ran
    const uint32_t *tile_perm;       // [XCD label][slot]: perm_per_label slots per label
    uint32_t perm_per_label;
    uint16_t *tile_cost;
};
constexpr uint32_t HIT_QUEUES = 64;   // one fill counter (its own 128-byte line) per queue: a wave's append is one atomic, spread over 64 words

__device__ __forceinline__ float sat(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f); }
__device__ __forceinline__ float lerp(float a, float b, float t) { return __builtin_fmaf(t, b - a, a); }
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, ax * bx));
}

// R8_UNorm decode, bit-identical to (float)b / 255.0f for b = 0..255 (checked
// exhaustively by tests/test_gpu_parity.py::test_unorm_table): 1/255 split into
// fl(1/255) + a correction term, summed by one fma -- one rounding, like the division.
__device__ __forceinline__ float unorm8(float b)
{
    const float r_hi = 0x1.010102p-8f;                   // fl(1/255)
    const float r_lo = __uint_as_float(0xaf7efeffu);     // fl(1/255 - r_hi) = -2.3191758e-10
    return __builtin_fmaf(b, r_hi, b * r_lo);
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

// A cell as the sampling code sees it: Cube (Compute.hlsl:31-58) plus the 8 value
// bytes of the node the cursor sits on.  inv == 1/scale exactly.
struct Cell {
    float lx, ly, lz, scale, inv;
    uint32_t v0, v1;
};
// What find() hands to the sample that follows it at the same position: nothing (generic cursor),
// or the position in units of 2^-LM -- an exact power-of-two scaling the cursor-stack find computes
// anyway (sample_after_find below).
struct Unscaled {};
struct Scaled { float x, y, z; };

// ---- generic cursor: float box, links followed through memory ----------------
// The shader's static `index` + `box` (Compute.hlsl:12,61) plus the record of the
// node it sits on, kept in registers.
struct CursorG {
    typedef Unscaled Pos;
    float lx, ly, lz, scale, inv;
    int32_t parent, children;
    uint32_t v0, v1;
    uint32_t index;
    uint32_t loads;          // 16-byte records this lane has loaded (the kernels zero it; only the counting ones read it)

    __device__ __forceinline__ void reset(const NodeRec &root)
    {
        lx = ly = lz = 0.0f; scale = 1.0f; inv = 1.0f; index = 0;
        parent = (int32_t)root.x; children = (int32_t)root.y; v0 = root.z; v1 = root.w;
    }
    __device__ __forceinline__ Cell cell() const { return Cell{lx, ly, lz, scale, inv, v0, v1}; }
};

// Cube::inside, Compute.hlsl:50-53
__device__ __forceinline__ bool inside(const CursorG &c, float px, float py, float pz)
{
    float hx = c.lx + c.scale, hy = c.ly + c.scale, hz = c.lz + c.scale;
    return (c.lx <= px && c.ly <= py && c.lz <= pz) && (px <= hx && py <= hy && pz <= hz);
}
// Cube::scale_up, Compute.hlsl:36-40
__device__ __forceinline__ void scale_up(CursorG &c)
{
    c.scale *= 2.0f;
    c.inv *= 0.5f;
    c.lx = floorf(c.lx * c.inv) * c.scale;
    c.ly = floorf(c.ly * c.inv) * c.scale;
    c.lz = floorf(c.lz * c.inv) * c.scale;
}
// child octant of pos in the current cell: (int3) saturate((pos - lower) / scale * 2),
// Compute.hlsl:100, then Cube::scale_down, Compute.hlsl:41-45.  Returns p = x + 2y + 4z.
__device__ __forceinline__ int descend_box(CursorG &c, float px, float py, float pz)
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

// find, Compute.hlsl:88-108 -- generic form: follows parent and children links
// through memory exactly as the shader does (the entry read of data[index] is
// served from the registers that already hold that record).  Returns the number
// of node records the *reference* reads in this call.
__device__ __forceinline__ uint32_t find(CursorG &c, const NodeRec *__restrict__ nodes, const GridRef &,
                                         uint32_t n_nodes, int32_t *, uint32_t, float px, float py, float pz, Unscaled &)
{
    uint32_t reads = 1;
    while (!inside(c, px, py, pz) && c.parent >= 0) {
        c.index = (uint32_t)c.parent;
        NodeRec r = nodes[c.index];
        c.parent = (int32_t)r.x; c.children = (int32_t)r.y; c.v0 = r.z; c.v1 = r.w;
        scale_up(c);
        reads++; c.loads++;
    }
    int iterations = 0;
    while (c.index < n_nodes && iterations < 12 && c.children >= 0) {
        int p = descend_box(c, px, py, pz);
        c.index = (uint32_t)(c.children + p);
        NodeRec r = nodes[c.index];
        c.parent = (int32_t)r.x; c.children = (int32_t)r.y; c.v0 = r.z; c.v1 = r.w;
        iterations++;
        reads++; c.loads++;
    }
    return reads;
}

// ---- cursor-stack form: integer cell coordinates, ancestors in LDS -----------
// For parent/child-consistent trees of depth <= LM = 12 (checked at upload).
//
// Why it visits the shader's cells.  Every box the shader can hold is a dyadic
// cell: lower = a * 2^-LM with integer a, scale = 2^-level.  All of Cube's float
// arithmetic on such boxes is exact (scale_up's floor(lower/scale)*scale, and
// scale_down's lower += p*scale), so a cell is fully described by (level, a).
//   * inside(pos), per axis lower <= pos <= lower + scale, with u = pos * 2^LM
//     (exact) is  a <= u <= a + 2^s,  s = LM - level.  With A = floor(u) and
//     B = ceil(u) - 1 (B = A unless u is an integer, then B = A - 1) this is
//     (B >> t) <= (a >> t) <= (A >> t) at t = s; the ancestor k levels up has
//     the same test at t = s + k.  Since A - B <= 1, the test holds exactly when
//     a >> t equals A >> t or B >> t, i.e. t >= min(bitlen(a^A), bitlen(a^B)).
//     So the ascend loop of Compute.hlsl:93-97 stops after
//         k = min(level, max(0, max_axes(min(bitlen(a^A), bitlen(a^B))) - s))
//     steps: no loop, no memory.  (A is clamped to [-1, 2^LM+1]; NaN gives -1:
//     then nothing matches and the cursor goes to the root, as in the shader,
//     where every comparison with NaN is false.)
//   * descent, per axis (int) saturate((pos - lower)/scale*2): pos - lower is
//     exact in fp32 here (both are multiples of ulp(pos), the difference is not
//     larger than the operands), so the bit is [pos >= cell midpoint].  Inside
//     the cell that is the next lower bit of A -- or of B when the cell matched
//     through B only (pos exactly on its upper face: the shader's saturate
//     gives 1 on every level below, and B's low bits are all ones).  Outside
//     the root cube saturate clamps: all zeros below 0, all ones above 1, which
//     is what clamping A/B to [0, 2^LM - 1] yields.
//   * `index < buffer_size` always holds for a validated tree and `iterations <
//     12` never binds, because one call descends at most depth <= 12 levels.
// The only thing the descent needs from an ancestor is its `children` field,
// which was pushed to an LDS stack (stack[level * stride + tid]: conflict-free
// for any mix of levels) when the path went down through it.
constexpr int LM = 12;

// Top grid.  Far from the surface the cells are coarse and a march step usually leaves its cell
// through a face of a shallow ancestor: half of all records a descent loads belong to levels 1-3,
// 87 % to levels 1-6 (scripts/descent_levels.py), each one a dependent load.  The grid holds, for
// every cell of level TG, the record of the deepest node of level <= TG that contains it:
// {level, values, children}.  A descent that restarts above level
// TG takes ONE load from the grid instead of up to TG dependent loads, and ends in the same node.
// TG is chosen per scene at upload (RenderParams::top_level); top == nullptr disables it.  When the
// grid is as deep as the tree the kernels use CursorF (below) instead of CursorS, and the cells hold
// LM - level instead of the level (and CursorF loads only the first 12 bytes of a cell).
//
// Cell (x, y, z) of the level-TG grid -> index: plain x-y-z order.  Blocked orders (2^B cells per
// axis contiguous; B = 1 puts the 8 cells of one parent in one 128-byte line) were measured and are
// slower, 0.183 vs 0.172 ms per 1080p frame for B = 1, 2, 3: the index arithmetic costs more than
// the better locality returns.  Kept as a build knob.
#ifndef TOP_BLOCK_BITS
#define TOP_BLOCK_BITS 0
#endif
__host__ __device__ __forceinline__ uint32_t top_index(uint32_t x, uint32_t y, uint32_t z, int TG)
{
    constexpr int B = TOP_BLOCK_BITS;
    if (B == 0 || TG <= B) return (((z << TG) | y) << TG) | x;       // two v_lshl_or_b32
    constexpr uint32_t m = (1u << B) - 1u;
    const int H = TG - B;
    const uint32_t hi = (x >> B) | ((y >> B) << H) | ((z >> B) << (2 * H));
    const uint32_t lo = (x & m) | ((y & m) << B) | ((z & m) << (2 * B));
    return (hi << (3 * B)) | lo;
}

// ---- cursor-stack cursor: level, integer anchor, children index, values ------------------------
struct CursorS {
    typedef Scaled Pos;
    int32_t ax, ay, az;      // lower * 2^LM
    int32_t level;
    int32_t children;
    uint32_t v0, v1;
    uint32_t loads;          // 16-byte records / grid cells this lane has loaded (the kernels zero it; only the counting ones read it)

    __device__ __forceinline__ void reset(const NodeRec &root)
    {
        ax = ay = az = 0; level = 0;
        children = (int32_t)root.y; v0 = root.z; v1 = root.w;
    }
    __device__ __forceinline__ Cell cell() const
    {
        Cell k;
        const float q = 1.0f / 4096.0f;
        k.lx = (float)ax * q; k.ly = (float)ay * q; k.lz = (float)az * q;   // exact
        k.scale = __int_as_float((127 - level) << 23);                      // 2^-level
        k.inv = __int_as_float((127 + level) << 23);                        // 2^level
        k.v0 = v0; k.v1 = v1;
        return k;
    }
};

// per axis: u = pos * 2^LM, its floor, and A = floor clamped to [-2, 2^LM + 1].  Every
// negative A means "below the cube" (it matches no cell and clamps to octant bits 0).
// v_med3_f32 returns the minimum of its non-NaN operands when one is NaN, so NaN gives
// -2: nothing matches, the cursor goes to the root and descends with octant bits 0 --
// what the shader does when every comparison with NaN is false and saturate(NaN) is 0.
__device__ __forceinline__ int32_t axis_a(float p, float &u, float &f)
{
    u = p * 4096.0f;
    f = floorf(u);
    return (int32_t)__builtin_amdgcn_fmed3f(f, -2.0f, 4097.0f);
}
__device__ __forceinline__ int bitlen(uint32_t x) { return 32 - __clz((int)x); }   // bitlen(0) = 0

// ON_GRID = false: no lane of the wave has a coordinate exactly on the 2^-LM grid, so
// B == A on every axis and the B terms drop out (the common case; the wave-uniform
// branch in find() picks it).
template <bool ON_GRID>
__device__ __forceinline__ uint32_t find_s(CursorS &c, const NodeRec *__restrict__ nodes,
                                           const TopCell *__restrict__ top, const int TG,
                                           int32_t *__restrict__ stack, uint32_t stride,
                                           int32_t Ax, int32_t Ay, int32_t Az, bool gx, bool gy, bool gz)
{
    const int32_t Bx = Ax - ((ON_GRID && gx) ? 1 : 0);
    const int32_t By = Ay - ((ON_GRID && gy) ? 1 : 0);
    const int32_t Bz = Az - ((ON_GRID && gz) ? 1 : 0);
    const int s = LM - c.level;
    int t;
    if (ON_GRID) {
        const int tx = min(bitlen((uint32_t)(c.ax ^ Ax)), bitlen((uint32_t)(c.ax ^ Bx)));
        const int ty = min(bitlen((uint32_t)(c.ay ^ Ay)), bitlen((uint32_t)(c.ay ^ By)));
        const int tz = min(bitlen((uint32_t)(c.az ^ Az)), bitlen((uint32_t)(c.az ^ Bz)));
        t = max(max(tx, ty), max(tz, s));
    } else {
        // the longest of the three differences = the length of their OR
        t = max(bitlen((uint32_t)(c.ax ^ Ax) | (uint32_t)(c.ay ^ Ay) | (uint32_t)(c.az ^ Az)), s);
    }
    int k = min(t - s, c.level);                 // ascents (Compute.hlsl:93-97)
    uint32_t reads = 1u + (uint32_t)k;
    if (k > 0) {
        c.level -= k;
        // an ancestor has children (consistent tree); above the top grid their index is not needed
        c.children = (top && c.level < TG) ? 0 : stack[(uint32_t)c.level * stride];
    }
    if (c.children >= 0) {
        // where the descent reads its octant bits from (see above), clamped into the cube
        int32_t Dx = Ax, Dy = Ay, Dz = Az;
        if (ON_GRID) {
            const int tt = LM - c.level;         // <= LM
            Dx = (((uint32_t)(c.ax ^ Ax) >> tt) == 0u) ? Ax : Bx;
            Dy = (((uint32_t)(c.ay ^ Ay) >> tt) == 0u) ? Ay : By;
            Dz = (((uint32_t)(c.az ^ Az) >> tt) == 0u) ? Az : Bz;
        }
        Dx = min(max(Dx, 0), 4095); Dy = min(max(Dy, 0), 4095); Dz = min(max(Dz, 0), 4095);
        if (top && c.level < TG) {
            // through the top grid: the node this descent reaches at level TG (or the leaf above it)
            const int sh = LM - TG;
            const uint32_t cellidx = top_index((uint32_t)Dx >> sh, (uint32_t)Dy >> sh, (uint32_t)Dz >> sh, TG);
            const uint4 e = reinterpret_cast<const uint4 *>(top)[cellidx];
            const int lvl = (int)e.x;
            c.loads++;
            reads += (uint32_t)(lvl - c.level);
            c.level = lvl;
            c.children = (int32_t)e.w;
            c.v0 = e.y;
            c.v1 = e.z;
        }
        while (c.children >= 0) {
            stack[(uint32_t)c.level * stride] = c.children;
            const int sb = LM - 1 - c.level;
            uint32_t p = ((uint32_t)Dx >> sb & 1u) | (((uint32_t)Dy >> sb & 1u) << 1) | (((uint32_t)Dz >> sb & 1u) << 2);
            NodeRec r = nodes[(uint32_t)c.children + p];
            c.children = (int32_t)r.y;
            c.v0 = r.z;
            c.v1 = r.w;
            // keep the value bytes live in every iteration: otherwise the compiler loads only
            // `children` in the loop and fetches the leaf's values in a second, dependent
            // round trip after it
            asm volatile("" : "+v"(c.v0), "+v"(c.v1));
            c.level++;
            reads++; c.loads++;
        }
        const int32_t keep = ~((1 << (LM - c.level)) - 1);
        c.ax = Dx & keep; c.ay = Dy & keep; c.az = Dz & keep;
    } else if (k > 0) {
        // unreachable for a consistent tree (an ancestor always has children); keep the
        // anchor coherent anyway
        const int32_t keep = ~((1 << (LM - c.level)) - 1);
        c.ax &= keep; c.ay &= keep; c.az &= keep;
    }
    return reads;
}

// ---- cursor when the top grid is as deep as the tree ---------------------------------------------
// Every leaf is a grid cell or a block of cells, so the grid says where a position's leaf is and the
// cursor shrinks to the leaf's anchor, size and values: no children index, no ancestor stack.  It
// keeps s = LM - level (what the shifts need); such a grid stores s instead of the level (k_top_grid).
// After reset the anchor is a mark that no position matches -- the first find of a pixel always looks
// its cell up -- while the values are the root's, and cell() maps the mark back to the root box: a
// gradient taken before any find (Compute.hlsl:207 with margin >= 0.5) sees node 0 in [0,1]^3, as the
// shader's cursor does.
// FLAT cells.  Far from the surface a leaf holds 8 equal bytes (the quantiser clamps at 1.5 s and
// -0.5 s); its trilinear blend is unorm8(byte) exactly (every lerp of equal operands returns the
// operand: fma(t, a - a, a) = a for finite t, and d is saturated, hence finite), so the distance such
// a cell returns depends on the cell alone.  k_top_grid / k_fine_blocks evaluate it once, with the
// operations of sample_after_find in their order, and store {s | FLAT_BIT, bytes, distance bits}: a
// march step through empty space -- two thirds of all steps of the bench frame -- is one sign test and
// a move instead of a decode and seven lerps.  (The shifts that take s use its low five bits, as the
// hardware does.)
// EXACT: the kernel reports the algorithmic read count, so a NaN coordinate must behave exactly as in
// find_s (it matches no cell: ascents up to the root).  Without it NaN converts to 0 and the find
// lands in the same leaf -- the cell at the origin -- without the clamp instructions.
// SPLIT: the grid is a coarse dense level whose internal cells point at blocks of finer cells (one more
// dependent load for the positions near the surface, a fraction of the memory of a dense grid of the
// tree's depth: the form for trees of depth 10-12).
constexpr uint32_t FLAT_BIT = 0x80000000u;
// ORDERED: honour GridRef::fine_order (the path tracer's scatter grid); the other kernels' grids are in x-y-z order
template <bool EXACT, bool SPLIT, bool ORDERED = false>
struct CursorFT {
    typedef Scaled Pos;
    static constexpr int32_t ROOT_MARK = 0x40000000;
    // The coordinates of the last lookup, in units of 2^-(LM - sh): the cell's lower corner * 2^LM is these shifted left by
    // sh, with the bits below s cleared (masking is left to the readers: cell(), sample_after_find).  The counting
    // cursors (EXACT) work in units of 2^-LM, sh = 0; the others in units of the grid's own cells (sh = LM - its full level;
    // find() sets it), which saves the shifts between the two in every step.
    int32_t ax, ay, az;
    uint32_t s;              // LM - level, | FLAT_BIT
    uint32_t v0, v1;         // the 8 value bytes; a flat cell: v0 = its byte four times, v1 = its distance (float bits)
    uint32_t sh;             // wave-uniform: see ax
    uint32_t loads;          // grid cells this lane has loaded (the kernels zero it; only the counting ones read it)

    __device__ __forceinline__ void reset(const NodeRec &root)
    {
        ax = ay = az = ROOT_MARK; s = (uint32_t)LM; v0 = root.z; v1 = root.w; sh = 0;
    }
    // for the queues between kernels: coordinates in units of 2^-LM whatever the cursor works in, and back
    __device__ __forceinline__ int4 pack() const
    {
        const bool fresh = ax == ROOT_MARK;
        return make_int4(fresh ? ax : (int32_t)((uint32_t)ax << sh), fresh ? ay : (int32_t)((uint32_t)ay << sh), fresh ? az : (int32_t)((uint32_t)az << sh), (int32_t)s);
    }
    __device__ __forceinline__ void unpack(const int4 &b, uint32_t shift)
    {
        const bool fresh = b.x == ROOT_MARK;
        ax = fresh ? b.x : b.x >> shift; ay = fresh ? b.y : b.y >> shift; az = fresh ? b.z : b.z >> shift;
        s = (uint32_t)b.w; sh = fresh ? 0u : shift;
    }
    // the unit shift of a grid as deep as the tree (full level = top_level, + fine_bits for a split grid)
    __device__ __forceinline__ static uint32_t units_shift(int full_level) { return EXACT ? 0u : (uint32_t)(LM - full_level); }
    __device__ __forceinline__ Cell cell() const
    {
        Cell k;
        const float q = 1.0f / 4096.0f;
        const bool fresh = ax == ROOT_MARK;                                  // never looked up: the root box
        const int32_t keep = (int32_t)(0xFFFFFFFFu << (s & 31u));
        const int32_t x = (int32_t)((uint32_t)ax << sh) & keep, y = (int32_t)((uint32_t)ay << sh) & keep, z = (int32_t)((uint32_t)az << sh) & keep;
        k.lx = fresh ? 0.0f : (float)x * q; k.ly = fresh ? 0.0f : (float)y * q; k.lz = fresh ? 0.0f : (float)z * q;   // exact
        const uint32_t sc = s & 15u;
        k.scale = __uint_as_float((127u - LM + sc) << 23);                  // 2^-level
        k.inv = __uint_as_float((127u + LM - sc) << 23);                    // 2^level
        k.v0 = v0; k.v1 = (s & FLAT_BIT) ? v0 : v1;
        return k;
    }
};

// One lookup in a grid as deep as the tree (dense, or coarse level + fine blocks): the leaf of the clamped
// cell coordinates D, into the cursor.  Returns s of that leaf.
template <bool EXACT, bool SPLIT, bool ORDERED>
__device__ __forceinline__ int load_cell(CursorFT<EXACT, SPLIT, ORDERED> &c, const GridRef &g, int32_t Dx, int32_t Dy, int32_t Dz)
{
    const int TG = g.level, sh = LM - TG;
    const uint32_t ti = top_index((uint32_t)Dx >> sh, (uint32_t)Dy >> sh, (uint32_t)Dz >> sh, TG);
    uint4 e = reinterpret_cast<const uint4 *>(g.top)[ti];
    c.loads++;
#ifdef SDFHIP_EXPERIMENTS
    if (EXACT && g.touch_top) touch_line(g.touch_top, g.touch_top_words, ti);
#endif
    if (SPLIT) {
        // one 16-byte load for the whole cell (left alone, the compiler fetches level and children first
        // and the values in a second, dependent load)
        asm volatile("" : "+v"(e.y), "+v"(e.z), "+v"(e.w));
        if (e.x == 15u) {                         // internal at the coarse level: its block of fine cells
            c.loads++;
            const int FB = g.fine_bits, sh2 = sh - FB;
            const uint32_t m = (1u << FB) - 1u;
            const uint32_t local = fine_cell_index(((uint32_t)Dx >> sh2) & m, ((uint32_t)Dy >> sh2) & m, ((uint32_t)Dz >> sh2) & m, FB, ORDERED ? g.fine_order : 0);
#ifdef SDFHIP_EXPERIMENTS
            if (EXACT && g.touch_fine) touch_line(g.touch_fine, g.touch_fine_words, e.w + local);
#endif
            e = reinterpret_cast<const uint4 *>(g.fine)[e.w + local];      // children = the block's first cell (32-bit: < 2^31 fine cells)
        }
    }
    c.s = e.x;                                    // LM - level of the leaf, | FLAT_BIT
    c.v0 = e.y;
    c.v1 = e.z;
    c.ax = Dx; c.ay = Dy; c.az = Dz;              // the anchor is these with the low s bits cleared: see CursorFT
    return (int)(e.x & 15u);
}

// The same lookup with D in units of the grid's full level (level, + fine_bits for a split grid): no shifts.
template <bool SPLIT, bool ORDERED>
__device__ __forceinline__ void load_cell_units(CursorFT<false, SPLIT, ORDERED> &c, const GridRef &g, int32_t Dx, int32_t Dy, int32_t Dz)
{
    uint4 e;
    c.loads++;
    if (!SPLIT) {
        e = reinterpret_cast<const uint4 *>(g.top)[top_index((uint32_t)Dx, (uint32_t)Dy, (uint32_t)Dz, g.level)];
    } else {
        const int FB = g.fine_bits;
        e = reinterpret_cast<const uint4 *>(g.top)[top_index((uint32_t)Dx >> FB, (uint32_t)Dy >> FB, (uint32_t)Dz >> FB, g.level)];
        asm volatile("" : "+v"(e.y), "+v"(e.z), "+v"(e.w));       // one 16-byte load, see load_cell
        if (e.x == 15u) {
            c.loads++;
            const uint32_t m = (1u << FB) - 1u;
            const uint32_t local = fine_cell_index((uint32_t)Dx & m, (uint32_t)Dy & m, (uint32_t)Dz & m, FB, ORDERED ? g.fine_order : 0);
            e = reinterpret_cast<const uint4 *>(g.fine)[e.w + local];
            asm volatile("" : "+v"(e.y), "+v"(e.z), "+v"(e.w));   // the same shape as the coarse load: one register tuple for both
        }
    }
    c.s = e.x; c.v0 = e.y; c.v1 = e.z;
    c.ax = Dx; c.ay = Dy; c.az = Dz;
}

// The reference's choice between the two cells that meet in a position exactly on a cell face (the same A/B rule as
// find_s): (ax, ay, az, s) the cursor in units of 2^-LM (or the root mark), D the floor coordinates, g* "this axis is
// on the 2^-LM grid".  Replaces D by the coordinates of the cell the descent ends in; k = the ascents.
__device__ __forceinline__ int on_face_choice(int32_t cax, int32_t cay, int32_t caz, int s, int32_t &Dx, int32_t &Dy, int32_t &Dz,
                                              bool gx, bool gy, bool gz, bool &moved)
{
    const int level = LM - s;
    // the A/B arithmetic wants coordinates that do not wrap: far outside the cube is -2 or 2^LM + 1
    Dx = min(max(Dx, -2), 4097); Dy = min(max(Dy, -2), 4097); Dz = min(max(Dz, -2), 4097);
    const bool root = cax == 0x40000000;
    const int32_t ax = root ? 0 : cax, ay = root ? 0 : cay, az = root ? 0 : caz;
    const int32_t Bx = Dx - (gx ? 1 : 0), By = Dy - (gy ? 1 : 0), Bz = Dz - (gz ? 1 : 0);
    const int tx = min(bitlen((uint32_t)(ax ^ Dx)), bitlen((uint32_t)(ax ^ Bx)));
    const int ty = min(bitlen((uint32_t)(ay ^ Dy)), bitlen((uint32_t)(ay ^ By)));
    const int tz = min(bitlen((uint32_t)(az ^ Dz)), bitlen((uint32_t)(az ^ Bz)));
    const int k = min(max(max(tx, ty), max(tz, s)) - s, level);
    moved = k > 0 || root;
    const int tt = s + k;                        // LM - (level the descent restarts at), <= LM
    Dx = (((uint32_t)(ax ^ Dx) >> tt) == 0u) ? Dx : Bx;
    Dy = (((uint32_t)(ay ^ Dy) >> tt) == 0u) ? Dy : By;
    Dz = (((uint32_t)(az ^ Dz) >> tt) == 0u) ? Dz : Bz;
    return k;
}

// find(): is the position still in the current cell, and -- on exact cell boundaries (any_on_grid,
// wave-uniform) -- which of the two adjacent cells does the reference's descent pick (the same A/B
// rule as find_s).  The ascent count k feeds the algorithmic read count only.  One place updates the
// cursor, so the two branches join on D and k, not on the cursor.
template <bool EXACT, bool SPLIT, bool ORDERED>
__device__ __forceinline__ uint32_t find_full(CursorFT<EXACT, SPLIT, ORDERED> &c, const GridRef &g,
                                              int32_t Dx, int32_t Dy, int32_t Dz, bool gx, bool gy, bool gz,
                                              const bool any_on_grid)
{
    const int s = (int)(c.s & 15u), level = LM - s;
    bool moved;
    int k;
    if (!any_on_grid) {
        const uint32_t diff = (uint32_t)(c.ax ^ Dx) | (uint32_t)(c.ay ^ Dy) | (uint32_t)(c.az ^ Dz);
        moved = (diff >> s) != 0u;                    // on the root: the mark differs from every A in bit 30
        k = min(max(bitlen(diff), s) - s, level);
    } else {
        k = on_face_choice(c.ax, c.ay, c.az, s, Dx, Dy, Dz, gx, gy, gz, moved);
    }
    uint32_t reads = 1u;
    if (moved) {
        Dx = min(max(Dx, 0), 4095); Dy = min(max(Dy, 0), 4095); Dz = min(max(Dz, 0), 4095);
        const int ns = load_cell(c, g, Dx, Dy, Dz);   // LM - level of the leaf
        reads = 1u + (uint32_t)k + (uint32_t)((LM - ns) - (level - k));
    }
    return reads;
}
// floor(u) as an integer in one instruction.  A float-to-int cast of NaN or of a value outside the int range is
// undefined in C++; v_cvt_flr_i32_f32 is not: it saturates, and NaN gives INT_MAX (scripts/micro/cvt_check.hip).
__device__ __forceinline__ int32_t cvt_floor(float u)
{
    int32_t a;
    asm("v_cvt_flr_i32_f32_e32 %0, %1" : "=v"(a) : "v"(u));
    return a;
}
// The kernels that do not count work in units of the grid's own cells, 2^-F (F = the grid's full level = the tree's depth):
//   * every step looks its cell up -- no "still in my cell?" test: the march steps are as long as the cells are wide, 98 %
//     of the lane-steps leave their cell, so the wave's load is issued anyway, and the lookup of a position in the cell
//     it came from returns that cell;
//   * the cell coordinates are floor(clamp(pos * 2^F)): outside the cube the shader's descent clamps to the boundary leaf
//     at every step (saturate, Compute.hlsl:100).  The clamp is a float v_med3 in front of the conversion: it returns the
//     smaller bound when its first operand is NaN, so a NaN coordinate selects cell 0 on that axis, as the shader does
//     (every comparison with NaN is false: up to the root; saturate(NaN) = 0: down along the low cells);
//   * a position exactly on a cell face needs the reference's choice between the two cells (on_face_choice).  Every face
//     lies on the 2^-F lattice, so "no coordinate of any lane is on that lattice" (a fractional part of zero) rules it out
//     for the whole wave; otherwise the wave takes the exact rule, in the 2^-LM coordinates the counting kernels use.
//   * FRESH: the cursor has not looked anything up yet (the first step of a pixel).  From the root the exact rule picks
//     the floor cell on every axis whose coordinate is inside the cube and the clamped one otherwise (on_face_choice with
//     a = 0, s = LM: the descent restarts at level 0 and "same cell above level 0" holds for 0 <= D < 2^LM) -- what the
//     plain lookup does, so the first step skips the lattice test.  (The reference's default camera sits at x = y = 0.5:
//     without this every pixel's first step takes the exact rule.)
// (Round 4 A/B'd three forms of this function with PMC -- the lattice branch told which way it usually goes, kept: the
// __builtin_expect below; the coarse cell's load issued by hand before the lattice test: waits 5 % less, issues 13 % more, slower;
// waves strictly inside the cube skipping the clamps: slower -- profiles/r04_k_march_ab.txt (history: commit 53ee955), docs/history.md section 4.6.)
template <bool SPLIT, bool FRESH, bool ORDERED>
__device__ __forceinline__ uint32_t find_units(CursorFT<false, SPLIT, ORDERED> &c, const GridRef &g, float px, float py, float pz, Scaled &u)
{
    const int F = g.level + (SPLIT ? g.fine_bits : 0), sh = LM - F;
    const float unit = __uint_as_float((uint32_t)(127 + F) << 23), top = unit - 1.0f;      // 2^F, and the last cell
    u.x = px * unit; u.y = py * unit; u.z = pz * unit;
    const float fx = __builtin_amdgcn_fractf(u.x), fy = __builtin_amdgcn_fractf(u.y), fz = __builtin_amdgcn_fractf(u.z);
    const float fm = __builtin_fminf(__builtin_fminf(fx, fy), fz);                           // NaN: never zero; the three are >= 0
    int32_t Dx, Dy, Dz;
    if (FRESH || __builtin_expect(__ballot(fm == 0.0f) == 0ull, 1)) {
        Dx = cvt_floor(__builtin_amdgcn_fmed3f(u.x, 0.0f, top)); Dy = cvt_floor(__builtin_amdgcn_fmed3f(u.y, 0.0f, top));
        Dz = cvt_floor(__builtin_amdgcn_fmed3f(u.z, 0.0f, top));
    } else {
        float tx, ty, tz, qx, qy, qz;
        Dx = axis_a(px, tx, qx); Dy = axis_a(py, ty, qy); Dz = axis_a(pz, tz, qz);
        const int4 k = c.pack();
        bool moved;
        on_face_choice(k.x, k.y, k.z, (int)(c.s & 15u), Dx, Dy, Dz, tx == qx, ty == qy, tz == qz, moved);
        Dx = min(max(Dx, 0), 4095) >> sh; Dy = min(max(Dy, 0), 4095) >> sh; Dz = min(max(Dz, 0), 4095) >> sh;
    }
    c.sh = (uint32_t)sh;
    load_cell_units(c, g, Dx, Dy, Dz);
    return 0;
}
template <bool SPLIT, bool ORDERED>
__device__ __forceinline__ uint32_t find(CursorFT<false, SPLIT, ORDERED> &c, const NodeRec *__restrict__, const GridRef &g,
                                         uint32_t, int32_t *__restrict__, uint32_t, float px, float py, float pz,
                                         Scaled &u)
{
    return find_units<SPLIT, false, ORDERED>(c, g, px, py, pz, u);
}
// find() for a cursor straight from reset(): the same as find() for every cursor kind but the one above
template <class CursorT, class Pos>
__device__ __forceinline__ uint32_t find_fresh(CursorT &c, const NodeRec *__restrict__ nodes, const GridRef &g, uint32_t n,
                                               int32_t *__restrict__ stack, uint32_t stride, float px, float py, float pz, Pos &u)
{
    return find(c, nodes, g, n, stack, stride, px, py, pz, u);
}
template <bool SPLIT, bool ORDERED>
__device__ __forceinline__ uint32_t find_fresh(CursorFT<false, SPLIT, ORDERED> &c, const NodeRec *__restrict__, const GridRef &g,
                                               uint32_t, int32_t *__restrict__, uint32_t, float px, float py, float pz,
                                               Scaled &u)
{
    return find_units<SPLIT, true, ORDERED>(c, g, px, py, pz, u);
}
// The counting kernels: NaN must match no cell (ascents up to the root count as reads), see axis_a.
template <bool SPLIT, bool ORDERED>
__device__ __forceinline__ uint32_t find(CursorFT<true, SPLIT, ORDERED> &c, const NodeRec *__restrict__, const GridRef &g,
                                         uint32_t, int32_t *__restrict__, uint32_t, float px, float py, float pz,
                                         Scaled &u)
{
    float fx, fy, fz;
    const int32_t Ax = axis_a(px, u.x, fx), Ay = axis_a(py, u.y, fy), Az = axis_a(pz, u.z, fz);
    const bool gx = u.x == fx, gy = u.y == fy, gz = u.z == fz;    // on the 2^-LM grid (false for NaN)
    return find_full(c, g, Ax, Ay, Az, gx, gy, gz, __ballot(gx || gy || gz) != 0ull);
}
// u = the position in the cursor's units, 2^-(LM - sh) (find() leaves it so): with a = the anchor in the same units and
// k = s - sh, (u - a) * 2^-k == (pos - lower) * 2^level bit for bit -- rounding is invariant under power-of-two scaling.
template <bool EXACT, bool SPLIT, bool ORDERED>
__device__ __forceinline__ float sample_after_find(const CursorFT<EXACT, SPLIT, ORDERED> &c, const Scaled &u, float, float, float)
{
    if (c.s & FLAT_BIT) return __uint_as_float(c.v1);                            // see CursorFT
    const float scale = __uint_as_float((c.s + (uint32_t)(127 - LM)) << 23);     // 2^-level = 2^(s - LM)
    const uint32_t k = c.s - c.sh;                                               // non-flat here: s has no flag bit set
    const float inv = __uint_as_float((127u - k) << 23);                         // 2^-k
    const int32_t keep = (int32_t)(0xFFFFFFFFu << (k & 31u));
    float dx = sat((u.x - (float)(c.ax & keep)) * inv);
    float dy = sat((u.y - (float)(c.ay & keep)) * inv);
    float dz = sat((u.z - (float)(c.az & keep)) * inv);
    Texels t = decode(c.v0, c.v1);
    float loadL = bilerp(t.v[0], t.v[1], t.v[2], t.v[3], dx, dy);
    float loadH = bilerp(t.v[4], t.v[5], t.v[6], t.v[7], dx, dy);
    return (lerp(loadL, loadH, dz) - 0.25f) * scale * 2.0f;
}
// what k_top_grid / k_fine_blocks store in a flat cell: the line above on eight equal texels
__device__ __forceinline__ uint32_t flat_cell_distance_bits(uint32_t byte, uint32_t s)
{
    const float scale = __uint_as_float((s + (uint32_t)(127 - LM)) << 23);
    return __float_as_uint((unorm8((float)byte) - 0.25f) * scale * 2.0f);
}

__device__ __forceinline__ uint32_t find(CursorS &c, const NodeRec *__restrict__ nodes, const GridRef &g,
                                         uint32_t, int32_t *__restrict__ stack, uint32_t stride, float px, float py, float pz,
                                         Scaled &u)
{
    const TopCell *__restrict__ top = g.top;
    const int TG = g.level;
    float ux, uy, uz, fx, fy, fz;
    const int32_t Ax = axis_a(px, ux, fx), Ay = axis_a(py, uy, fy), Az = axis_a(pz, uz, fz);
    const bool gx = ux == fx, gy = uy == fy, gz = uz == fz;    // on the 2^-LM grid (false for NaN)
    u.x = ux; u.y = uy; u.z = uz;
    if (__ballot(gx || gy || gz) == 0ull)
        return find_s<false>(c, nodes, top, TG, stack, stride, Ax, Ay, Az, false, false, false);
    return find_s<true>(c, nodes, top, TG, stack, stride, Ax, Ay, Az, gx, gy, gz);
}

// Cube::interpol_world -> sample_at, Compute.hlsl:54-58,19-29
__device__ __forceinline__ float interpol_world(const Cell &c, float px, float py, float pz)
{
    // Cells far from the surface hold 8 equal bytes (the quantiser clamps at 1.5 s and
    // -0.5 s).  Every lerp of equal operands returns the operand exactly
    // (fma(t, a - a, a) = a for finite t; d is saturated, hence finite), so the
    // trilinear blend of such a cell is unorm8(byte) bit for bit.  When that holds for
    // every active lane of the wave -- whole wavefronts of sky rays -- skip the blend.
    // (Four equal bytes <=> the word equals itself rotated by 8 bits: one full-rate instruction.)
    const bool flat = c.v0 == c.v1 && c.v0 == __builtin_amdgcn_alignbit(c.v0, c.v0, 8);
    if (__ballot(!flat) == 0ull)
        return (unorm8((float)(c.v0 & 0xFFu)) - 0.25f) * c.scale * 2.0f;
    float dx = sat((px - c.lx) * c.inv);
    float dy = sat((py - c.ly) * c.inv);
    float dz = sat((pz - c.lz) * c.inv);
    Texels t = decode(c.v0, c.v1);
    float loadL = bilerp(t.v[0], t.v[1], t.v[2], t.v[3], dx, dy);
    float loadH = bilerp(t.v[4], t.v[5], t.v[6], t.v[7], dx, dy);
    return (lerp(loadL, loadH, dz) - 0.25f) * c.scale * 2.0f;
}

// interpol_world at the position the preceding find() was called with.
__device__ __forceinline__ float sample_after_find(const CursorG &c, const Unscaled &, float px, float py, float pz)
{
    return interpol_world(c.cell(), px, py, pz);
}
// Cursor-stack form: with u = 2^LM * pos and a = 2^LM * lower (both exact),
// (u - a) * 2^(level - LM) == (pos - lower) * 2^level bit for bit -- rounding is invariant under
// power-of-two scaling -- which saves the conversion of the anchor back to world units.
__device__ __forceinline__ float sample_after_find(const CursorS &c, const Scaled &u, float, float, float)
{
    const float scale = __int_as_float((127 - c.level) << 23);                 // 2^-level
    const bool flat = c.v0 == c.v1 && c.v0 == __builtin_amdgcn_alignbit(c.v0, c.v0, 8);    // see interpol_world
    if (__ballot(!flat) == 0ull)
        return (unorm8((float)(c.v0 & 0xFFu)) - 0.25f) * scale * 2.0f;
    const float inv = __int_as_float((127 + c.level - LM) << 23);              // 2^(level - LM)
    float dx = sat((u.x - (float)c.ax) * inv);
    float dy = sat((u.y - (float)c.ay) * inv);
    float dz = sat((u.z - (float)c.az) * inv);
    Texels t = decode(c.v0, c.v1);
    float loadL = bilerp(t.v[0], t.v[1], t.v[2], t.v[3], dx, dy);
    float loadH = bilerp(t.v[4], t.v[5], t.v[6], t.v[7], dx, dy);
    return (lerp(loadL, loadH, dz) - 0.25f) * scale * 2.0f;
}

// gradient, Compute.hlsl:112-130 (taps at integer x / y have bilinear weight 0
// towards the neighbouring texel: lerp(a, b, 0) == a)
__device__ __forceinline__ void gradient(const Cell &c, float px, float py, float pz,
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
__device__ __forceinline__ void ray(const FrameInfo &I, uint32_t cx, uint32_t cy, float &dx,
                                    float &dy, float &dz)
{
    float sx = (float)cx / I.screen_h - I.half_aspect;
    float sy = (float)cy / I.screen_h - 0.5f;
    float vx = sx * I.fov, vy = sy * I.fov, vz = 0.5f;
    float d0 = dot3(vx, vy, vz, I.h0x, I.h0y, I.h0z);
    float d1 = dot3(vx, vy, vz, I.h1x, I.h1y, I.h1z);
    float d2 = dot3(vx, vy, vz, I.h2x, I.h2y, I.h2z);
    float rl = 1.0f / sqrtf(dot3(d0, d1, d2, d0, d1, d2));
    dx = d0 * rl;
    dy = d1 * rl;
    dz = d2 * rl;
}

// ---- path-traced mode helpers (oracle/sdf_oracle.c o_pcg / o_rnd / o_ray_f) ------------
__device__ __forceinline__ uint32_t pcg(uint32_t v)
{
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
// uniform in [0, 1): top 24 bits of the hash chained over (seed + pixel, sample, bounce, draw)
__device__ __forceinline__ float rnd(uint32_t seed, uint32_t p, uint32_t s, uint32_t b, uint32_t d)
{
    uint32_t h = pcg(pcg(pcg(pcg(seed + p) + s) + b) + d);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
// ray() through a fractional pixel coordinate
__device__ __forceinline__ void ray_f(const FrameInfo &I, float fx, float fy, float &dx, float &dy, float &dz)
{
    float sx = fx / I.screen_h - I.half_aspect;
    float sy = fy / I.screen_h - 0.5f;
    float vx = sx * I.fov, vy = sy * I.fov, vz = 0.5f;
    float d0 = dot3(vx, vy, vz, I.h0x, I.h0y, I.h0z);
    float d1 = dot3(vx, vy, vz, I.h1x, I.h1y, I.h1z);
    float d2 = dot3(vx, vy, vz, I.h2x, I.h2y, I.h2z);
    float rl = 1.0f / sqrtf(dot3(d0, d1, d2, d0, d1, d2));
    dx = d0 * rl;
    dy = d1 * rl;
    dz = d2 * rl;
}

}  // namespace sdfhip

#ifdef SDFHIP_EXPERIMENTS
#include "lab_device.h"
#endif
