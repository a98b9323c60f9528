// libsdfhip.so, device half, host side: the C-ABI entry points that own HBM (scene upload,
// render launches, band de-interleave) and the small helper kernels they use.  The ray-march
// kernels themselves: raymarch_kernels.h.
//
// Replaces the reference's dispatch and resource binding:
//   SdfBox/Program.cs:81,94               UpdateBuffer(info) + DispatchSized  -> sdfhip_render*
//   SdfBox/Program.cs:543-572,147-152     StructBuffer/ValueTexture + binding -> sdfhip_scene_upload
//   SdfBox/Program.cs:96-99               display pass draw                   -> sdfhip_render_display
#include "raymarch_kernels.h"
#include "sdfhip_internal.h"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <type_traits>
#include <vector>


namespace sdfhip {

// ---- small helper kernels -----------------------------------------------------
// {parent, children}[N] + bytes[N][8] -> fused 16-byte records (upload).
__global__ void k_fuse(const int2 *__restrict__ structs, const uint2 *__restrict__ values,
                       NodeRec *__restrict__ nodes, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int2 s = structs[i];
        uint2 v = values[i];
        nodes[i] = make_uint4((uint32_t)s.x, (uint32_t)s.y, v.x, v.y);
    }
}

// A leaf cell of a grid that serves CursorFT (t.level = LM - level): when its 8 bytes are equal, mark it flat and
// replace v1 by the distance it returns (raymarch_device.h, CursorFT)
__device__ __forceinline__ void flat_cell(TopCell &t)
{
    if (t.v0 == t.v1 && t.v0 == __builtin_amdgcn_alignbit(t.v0, t.v0, 8)) {
        t.v1 = flat_cell_distance_bits(t.v0 & 0xFFu, t.level);
        t.level |= FLAT_BIT;
    }
}

// The top grid of the cursor-stack kernels (raymarch_device.h): one thread per level-TG cell walks
// from the root by the cell's octant bits and stores the record it ends at.
// full = 0: cells hold the level (CursorS); 1: LM - level (CursorF, grid as deep as the tree); 2: like 1, but a
// cell whose node is still internal holds level 15 and the node's index: the coarse half of a split grid
__global__ void k_top_grid(const NodeRec *__restrict__ nodes, TopCell *__restrict__ top, int TG, int full)
{
    const uint32_t total = 1u << (3 * TG), mask = (1u << TG) - 1u;
    for (uint32_t cell = blockIdx.x * blockDim.x + threadIdx.x; cell < total; cell += gridDim.x * blockDim.x) {
        const uint32_t cx = cell & mask, cy = (cell >> TG) & mask, cz = cell >> (2 * TG);
        NodeRec r = nodes[0];
        uint32_t level = 0, index = 0;
        while (level < (uint32_t)TG && (int32_t)r.y >= 0) {
            const uint32_t sb = (uint32_t)TG - 1u - level;
            index = r.y + ((cx >> sb & 1u) | ((cy >> sb & 1u) << 1) | ((cz >> sb & 1u) << 2));
            r = nodes[index];
            level++;
        }
        TopCell t;
        // a grid as deep as the tree serves CursorFT, which wants LM - level
        t.level = full ? (uint32_t)LM - level : level; t.v0 = r.z; t.v1 = r.w; t.children = (int32_t)r.y;
        if (full == 2 && (int32_t)r.y >= 0) { t.level = 15u; t.children = (int32_t)index; }
        else if (full) flat_cell(t);
        top[top_index(cx, cy, cz, TG)] = t;
    }
}

// The fine half of a split grid: block b holds the 8^FB cells below the internal node block_node[b] of level TG,
// each the leaf that contains it (LM - level, values), in x-y-z order.
__global__ void k_fine_blocks(const NodeRec *__restrict__ nodes, const uint32_t *__restrict__ block_node,
                              TopCell *__restrict__ fine, uint32_t n_blocks, int TG, int FB, int order)
{
    const size_t total = (size_t)n_blocks << (3 * FB);
    const uint32_t mask = (1u << FB) - 1u;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (size_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)(j >> (3 * FB)), local = (uint32_t)j & ((1u << (3 * FB)) - 1u);
        const uint32_t cx = local & mask, cy = (local >> FB) & mask, cz = local >> (2 * FB);
        const size_t i = ((size_t)b << (3 * FB)) + fine_cell_index(cx, cy, cz, FB, order);    // where the cell is stored
        NodeRec r = nodes[block_node[b]];
        uint32_t level = (uint32_t)TG, down = 0;
        while (down < (uint32_t)FB && (int32_t)r.y >= 0) {
            const uint32_t sb = (uint32_t)FB - 1u - down;
            r = nodes[r.y + ((cx >> sb & 1u) | ((cy >> sb & 1u) << 1) | ((cz >> sb & 1u) << 2))];
            down++; level++;
        }
        TopCell t;
        t.level = (uint32_t)LM - level; t.v0 = r.z; t.v1 = r.w; t.children = -1;
        flat_cell(t);
        fine[i] = t;
    }
}

// ---- the grid's second form for the default kernel's loop: 4-byte words + sample records (CursorFF, raymarch_device.h) -------
// the 64-byte sample record of a non-flat leaf: s = LM - its level, its 8 value bytes, its lower corner a in units of 2^-F
__device__ __forceinline__ void write_sample_record(uint4 *__restrict__ rec, uint32_t s, uint32_t v0, uint32_t v1, uint32_t ax, uint32_t ay,
                                                    uint32_t az, int F)
{
    const uint32_t k = s - (uint32_t)(LM - F);                                              // the leaf is 2^k of the grid's cells wide
    const float inv = __uint_as_float((127u - k) << 23);
    const Texels q = decode(v0, v1);
    rec[0] = make_uint4(__float_as_uint(-((float)ax * inv)), __float_as_uint(-((float)ay * inv)), __float_as_uint(-((float)az * inv)), __float_as_uint(inv));
    rec[1] = make_uint4(__float_as_uint(q.v[0]), __float_as_uint(q.v[4]), __float_as_uint(q.v[2]), __float_as_uint(q.v[6]));
    rec[2] = make_uint4(__float_as_uint(q.v[1] - q.v[0]), __float_as_uint(q.v[5] - q.v[4]), __float_as_uint(q.v[3] - q.v[2]),
                        __float_as_uint(q.v[7] - q.v[6]));
    rec[3] = make_uint4(0u, 0u, 0u, 0u);
}
constexpr uint32_t D4_ANCHOR = 0x7FC00001u, D4_OTHER = 0x7FC00002u;       // pass-1 marks of a non-flat leaf's cells (NaNs: no distance)
// the leaf of the cell stored at word i of the level-F array, from the 16-byte cells; its anchor = its first cell, which will own the record
__device__ __forceinline__ uint4 d4_leaf(const GridRef &g, uint32_t i, int F, uint32_t &x, uint32_t &y, uint32_t &z, uint32_t &anchor)
{
    const uint32_t mask = (1u << F) - 1u;
    x = i & mask; y = (i >> F) & mask; z = i >> (2 * F);                  // (the inverse of top_index with TOP_BLOCK_BITS = 0)
    const uint4 e = cell_at(g, (int32_t)x, (int32_t)y, (int32_t)z);
    const uint32_t keep = 0xFFFFFFFFu << ((e.x & 15u) - (uint32_t)(LM - F));
    anchor = top_index(x & keep, y & keep, z & keep, F);
    return e;
}
// pass 1: a flat leaf's cells get its distance, the others a mark
__global__ __launch_bounds__(256) void k_d4_fill(const GridRef g, uint32_t *__restrict__ d4, int F)
{
    const size_t total = (size_t)1 << (3 * F);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x, y, z, a;
        const uint4 e = d4_leaf(g, (uint32_t)i, F, x, y, z, a);
        d4[i] = (e.x & FLAT_BIT) ? e.z : (a == (uint32_t)i ? D4_ANCHOR : D4_OTHER);
    }
}
// numbering the anchors in cell order (chunks of 256: count, k_split_scan over the chunk counts, then assign)
__global__ __launch_bounds__(256) void k_d4_count(const uint32_t *__restrict__ d4, size_t ncell, uint32_t n_chunks, uint32_t *__restrict__ chunk_sums)
{
    __shared__ uint32_t part[4];
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const size_t i = (size_t)c * 256u + threadIdx.x;
        const unsigned long long m = __ballot(i < ncell && d4[i] == D4_ANCHOR);
        if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) chunk_sums[c] = part[0] + part[1] + part[2] + part[3];
        __syncthreads();
    }
}
// pass 2: an anchor takes the next record, writes it, and its word becomes TAG + the record's index in 16-byte units
__global__ __launch_bounds__(256) void k_d4_anchor(const GridRef g, uint32_t *__restrict__ d4, size_t ncell, uint32_t n_chunks,
                                                   const uint32_t *__restrict__ chunk_offsets, uint4 *__restrict__ recs, int F)
{
    __shared__ uint32_t part[4];
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const size_t i = (size_t)c * 256u + threadIdx.x;
        const uint32_t w = threadIdx.x >> 6;
        const bool anchor = i < ncell && d4[i] == D4_ANCHOR;
        const unsigned long long m = __ballot(anchor);
        if ((threadIdx.x & 63u) == 0) part[w] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = chunk_offsets[c];
        for (uint32_t k = 0; k < w; k++) before += part[k];
        if (anchor) {
            const uint32_t id = before + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            uint32_t x, y, z, a;
            const uint4 e = d4_leaf(g, (uint32_t)i, F, x, y, z, a);
            write_sample_record(recs + (size_t)id * 4u, e.x & 15u, e.y, e.z, x, y, z, F);
            d4[i] = D4_TAG + id * 4u;
        }
        __syncthreads();
    }
}
// pass 3: the other cells of a non-flat leaf take their anchor's word (final since pass 2: no cell reads a cell this pass writes)
__global__ __launch_bounds__(256) void k_d4_share(const GridRef g, uint32_t *__restrict__ d4, int F)
{
    const size_t total = (size_t)1 << (3 * F);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        if (d4[i] != D4_OTHER) continue;
        uint32_t x, y, z, a;
        (void)d4_leaf(g, (uint32_t)i, F, x, y, z, a);
        d4[i] = d4[a];
    }
}

// sdfhip_octdata_validate (asdf_io.cpp) on the device, for the upload: the arrays are on their way to HBM anyway (8 ms for
// 451 MB) and the host's single-threaded pass over 28 M nodes takes 110 ms.  One thread per node; the same verdicts:
//   bad          a link out of range, or a parent chain of more than 64 links / a cycle (the host function then names the node)
//   inconsistent the root has a parent, a node's children block is block 0 or does not point back at it
//   depth        the most links from a node to the root among the nodes the root reaches (every link of the chain is mirrored by
//                its parent's children block): the host function's walk down from the root
// Reads nothing through a link it has not range-checked.
__global__ __launch_bounds__(256) void k_validate(const int2 *__restrict__ structs, uint32_t n, uint32_t *__restrict__ verdict)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    const int2 me = valid ? structs[i] : make_int2(-1, -1);
    bool bad = false, inconsistent = false;
    if (valid) {
        if (me.x >= 0 && (uint32_t)me.x >= n) bad = true;
        if (me.y >= 0 && (uint64_t)(uint32_t)me.y + 8u > (uint64_t)n) bad = true;
        if (i == 0 && me.x >= 0) inconsistent = true;
        if (!bad && me.y >= 0) {
            if (me.y == 0) inconsistent = true;
            for (int k = 0; k < 8; k++)
                if (structs[(uint32_t)me.y + (uint32_t)k].x != (int32_t)i) inconsistent = true;
        }
    }
    uint32_t links = 0;
    bool attached = valid && !bad;
    if (attached) {
        uint32_t j = i;
        int32_t p = me.x;
        while (p >= 0) {
            if ((uint32_t)p >= n) { attached = false; break; }            // that node's own thread reports the bad link
            const int2 up = structs[(uint32_t)p];
            if (!(up.y >= 0 && j >= (uint32_t)up.y && j - (uint32_t)up.y < 8u)) attached = false;
            j = (uint32_t)p; p = up.x;
            if (++links > 64u) { bad = true; attached = false; break; }
        }
        if (j != 0u) attached = false;
    }
    // one atomic per wave and verdict
    uint32_t m = attached ? links : 0u;
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    const uint32_t flags = (__ballot(bad) ? 1u : 0u) | (__ballot(inconsistent) ? 2u : 0u);
    if ((threadIdx.x & 63u) == 0) {
        if (flags) atomicOr(&verdict[0], flags);
        if (m) atomicMax(&verdict[1], m);
    }
}

// Numbering the internal cells of a split grid's coarse level in cell order (an exclusive prefix sum of "is internal"),
// on the device: chunks of 256 cells; count per chunk, scan of the chunk counts by one workgroup, then every internal cell
// gets its block -- children = the block's first fine cell -- and the block its node.
__global__ __launch_bounds__(256) void k_split_count(const TopCell *__restrict__ coarse, uint32_t ncell, uint32_t n_chunks,
                                                     uint32_t *__restrict__ chunk_sums)
{
    __shared__ uint32_t part[4];
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const uint32_t i = c * 256u + threadIdx.x;
        const bool internal = i < ncell && coarse[i].level == 15u;
        const unsigned long long m = __ballot(internal);
        if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) chunk_sums[c] = part[0] + part[1] + part[2] + part[3];
        __syncthreads();
    }
}
// exclusive scan of n values in place by ONE workgroup of 1024 threads; the total goes to v[n]
__global__ __launch_bounds__(1024) void k_split_scan(uint32_t *__restrict__ v, uint32_t n)
{
    __shared__ uint32_t part[1024];
    const uint32_t per = (n + 1023u) / 1024u, lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += v[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {           // inclusive scan of the 1024 partial sums
        const uint32_t add = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t i = lo; i < hi; i++) { const uint32_t x = v[i]; v[i] = run; run += x; }
    if (threadIdx.x == 1023u) v[n] = part[1023];
}
__global__ __launch_bounds__(256) void k_split_assign(TopCell *__restrict__ coarse, uint32_t ncell, uint32_t n_chunks,
                                                      const uint32_t *__restrict__ chunk_offsets, uint32_t *__restrict__ block_node, int FB)
{
    __shared__ uint32_t part[4];
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const uint32_t i = c * 256u + threadIdx.x, w = threadIdx.x >> 6;
        const bool internal = i < ncell && coarse[i].level == 15u;
        const unsigned long long m = __ballot(internal);
        if ((threadIdx.x & 63u) == 0) part[w] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = chunk_offsets[c];
        for (uint32_t k = 0; k < w; k++) before += part[k];
        if (internal) {
            const uint32_t id = before + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            block_node[id] = (uint32_t)coarse[i].children;              // the node the block hangs under
            coarse[i].children = (int32_t)(id << (3 * FB));             // the block's first fine cell
        }
        __syncthreads();
    }
}

// SDFHIP_FLAG_TILE_ORDER: the launch order of the next frame's tiles from the wave-iterations of this frame's (k_march's per-tile
// cost output).  A frame alone ends when its longest wave does -- a 100-step tile launched in the last round of workgroups adds
// its whole chain of dependent steps to the frame time -- so the expensive tiles go first.  Workgroup b runs on XCD b & 7 and the
// default order gives XCD x the tile rows x, x + 8, ...: every XCD keeps its own tiles (its L2 keeps seeing whole rows) and
// launches them in 8 classes of descending cost, in their old order within a class (a stable partition: neighbouring tiles of
// equal cost -- the sky -- still run together).  One workgroup per XCD label; thread t owns a contiguous span of the label's tiles.
// A tile's expected cost is the largest cost among the tiles up to ORDER_REACH away from it: the camera moves between frames,
// and a silhouette that crosses into a neighbouring tile must find that tile at the front too (with the camera turning round
// the scene by one degree per frame this takes a frame alone from 0.168 to 0.144 ms, the same as knowing the frame's own costs).
constexpr int ORDER_CLASSES = 8;
constexpr int ORDER_REACH = 2;
constexpr int ORDER_SPAN = 12;                                         // tiles per thread at most: 8 x 1024 x 12 order slots
// one thread per tile: its class, 0 = the longest (all CUs take part: the 25 reads per tile are the bulk of the work)
__global__ __launch_bounds__(256) void k_tile_class(const uint16_t *__restrict__ cost, uint8_t *__restrict__ cls, uint32_t tiles_x, uint32_t tiles_y)
{
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= tiles_x * tiles_y) return;
    const int tx = (int)(tile % tiles_x), ty = (int)(tile / tiles_x);
    uint32_t c = 0;                                                    // primary + shadow loop iterations, <= 140
    for (int y = max(ty - ORDER_REACH, 0); y <= min(ty + ORDER_REACH, (int)tiles_y - 1); y++)
        for (int x = max(tx - ORDER_REACH, 0); x <= min(tx + ORDER_REACH, (int)tiles_x - 1); x++) {
            const uint32_t v = cost[(size_t)y * tiles_x + x];
            c = max(c, (v & 0xFFu) + (v >> 8));
        }
    const uint32_t k = c / 18u;                                        // 0 .. 7
    cls[tile] = (uint8_t)((uint32_t)(ORDER_CLASSES - 1) - (k < (uint32_t)ORDER_CLASSES ? k : (uint32_t)(ORDER_CLASSES - 1)));
}
// one workgroup per XCD label: the stable partition of its tiles by class
__global__ __launch_bounds__(1024) void k_tile_order(const uint8_t *__restrict__ cls, uint32_t *__restrict__ perm,
                                                     uint32_t tiles_x, uint32_t tiles_y)
{
    __shared__ uint32_t wsum[ORDER_CLASSES][16];
    __shared__ uint32_t total[ORDER_CLASSES], base[ORDER_CLASSES + 1];
    const uint32_t x = blockIdx.x, t = threadIdx.x;
    const uint32_t per_label = ((tiles_y + 7u) >> 3) * tiles_x;       // workgroups (and order slots) per XCD label
    const uint32_t span = (per_label + 1023u) / 1024u, lo = min(per_label, t * span), hi = min(per_label, lo + span);
    auto tile_of = [&](uint32_t j) { const uint32_t r = j / tiles_x, row = r * 8u + x; return row < tiles_y ? row * tiles_x + (j - r * tiles_x) : 0xFFFFFFFFu; };
    // the classes of this thread's tiles, 4 bits each (15 = no tile), kept for the second pass
    unsigned long long packed = ~0ull;
    uint32_t mine[ORDER_CLASSES];
    for (int k = 0; k < ORDER_CLASSES; k++) mine[k] = 0;
#pragma unroll
    for (int i = 0; i < ORDER_SPAN; i++) {
        const uint32_t j = lo + (uint32_t)i;
        const uint32_t tile = j < hi ? tile_of(j) : 0xFFFFFFFFu;
        if (tile != 0xFFFFFFFFu) {
            const uint32_t c = cls[tile];
            packed = (packed & ~(0xFull << (4 * i))) | ((unsigned long long)c << (4 * i));
            for (int k = 0; k < ORDER_CLASSES; k++) mine[k] += c == (uint32_t)k ? 1u : 0u;      // (no indexed register array)
        }
    }
    // exclusive prefix of every class over the 1024 threads: within the wave by shuffles, across the 16 waves through LDS
    const uint32_t lane = t & 63u, wave = t >> 6;
    uint32_t before[ORDER_CLASSES];
    for (int k = 0; k < ORDER_CLASSES; k++) {
        uint32_t v = mine[k];
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(v, o); if ((int)lane >= o) v += y; }
        before[k] = v - mine[k];
        if (lane == 63u) wsum[k][wave] = v;
    }
    __syncthreads();
    if (t < (uint32_t)ORDER_CLASSES) {                                 // thread k: class k's waves, then the class bases
        uint32_t run = 0;
        for (int w = 0; w < 16; w++) { const uint32_t v = wsum[t][w]; wsum[t][w] = run; run += v; }
        total[t] = run;
    }
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (int k = 0; k < ORDER_CLASSES; k++) { base[k] = run; run += total[k]; }
        base[ORDER_CLASSES] = run;
    }
    __syncthreads();
    uint32_t at[ORDER_CLASSES];
    for (int k = 0; k < ORDER_CLASSES; k++) at[k] = base[k] + wsum[k][wave] + before[k];
#pragma unroll
    for (int i = 0; i < ORDER_SPAN; i++) {
        const uint32_t c = (uint32_t)(packed >> (4 * i)) & 0xFu;
        if (c == 0xFu) continue;
        uint32_t slot = 0;
        for (int k = 0; k < ORDER_CLASSES; k++) { const bool m = c == (uint32_t)k; slot = m ? at[k] : slot; at[k] += m ? 1u : 0u; }
        perm[(size_t)slot * 8u + x] = tile_of(lo + (uint32_t)i);
    }
    // the label's idle workgroups (rows past the frame's last tile row) behind its tiles
    const uint32_t real = base[ORDER_CLASSES];
    for (uint32_t j = real + t; j < per_label; j += 1024u) perm[(size_t)j * 8u + x] = 0xFFFFFFFFu;
}

// Gathered compact band buffers -> frame rows (rank-0 side of the tile gather).
// Which rank rendered a band, and where: round robin (n == 0), or an explicit map with
// src[band] = rank << 10 | local band (layouts with unequal shares).
struct BandMap {
    uint32_t n;
    uint16_t src[MAX_BAND_LIST];
};
struct WirePlanes {};   // In = WirePlanes: a rank's frame is the two planes SDFHIP_FLAG_WIRE renders make
template <class In, class Out>
__global__ void k_deinterleave(const In *__restrict__ gathered, Out *__restrict__ frame,
                               uint32_t width, uint32_t height, uint32_t band_rows, uint32_t world,
                               uint32_t rows_per_rank, uint32_t frames, const BandMap M, uint32_t only_rank)
{
    // only_rank != ~0: `gathered` is that one rank's buffer, and only its rows are written (the dense resend of a
    // rank whose sparse share overflowed)
    // gathered: [world][frames][rows_per_rank][width]  ->  frame: [frames][height][width]
    size_t per_frame = (size_t)width * height, total = per_frame * frames;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        uint32_t f = (uint32_t)(i / per_frame);
        size_t r = i - (size_t)f * per_frame;
        uint32_t y = (uint32_t)(r / width), x = (uint32_t)(r - (size_t)y * width);
        uint32_t band = y / band_rows, rank = band % world, lband = band / world;
        if (M.n) { const uint32_t e = M.src[band]; rank = e >> 10; lband = e & 1023u; }
        if (only_rank != 0xFFFFFFFFu) { if (rank != only_rank) continue; rank = 0; }
        uint32_t yl = lband * band_rows + (y - band * band_rows);
        if constexpr (std::is_same<In, WirePlanes>::value) {
            const size_t npx = (size_t)rows_per_rank * width, l = (size_t)yl * width + x;
            const char *base = reinterpret_cast<const char *>(gathered) + ((size_t)rank * frames + f) * npx * 5;
            frame[i] = wire_expand(reinterpret_cast<const float *>(base)[l], reinterpret_cast<const uint8_t *>(base)[4 * npx + l]);
        } else {
            frame[i] = gathered[(((size_t)rank * frames + f) * rows_per_rank + yl) * width + x];
        }
    }
}

// ---- sparse wire format of the tile gather ------------------------------------------------------
// A frame-share in the dense wire format (SDFHIP_FLAG_WIRE: rows*width floats, then rows*width code
// bytes) is mostly zeros in its float plane: sky and unlit pixels carry a = +0.  The sparse form keeps
// the code bytes, and per 8x8 tile a 64-bit mask of the pixels whose a has any bit set plus the index
// of the tile's first slot in a packed array of those floats (capacity slots; more are dropped and
// flagged).  Lossless within the capacity; about 1.2 bytes + 4 bytes per lit pixel instead of 5.
struct SparseLayout {
    uint32_t width, rows, tiles_x, tiles_y, tiles, capacity;
    size_t off_masks, off_bases, off_head, off_floats, bytes;
};
__host__ __device__ inline SparseLayout sparse_layout(uint32_t width, uint32_t rows, uint32_t capacity)
{
    SparseLayout L;
    L.width = width; L.rows = rows; L.capacity = capacity;
    L.tiles_x = (width + 7) / 8; L.tiles_y = (rows + 7) / 8; L.tiles = L.tiles_x * L.tiles_y;
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    L.off_masks = up((size_t)rows * width);
    L.off_bases = up(L.off_masks + (size_t)L.tiles * 8);
    L.off_head = up(L.off_bases + (size_t)L.tiles * 4);
    L.off_floats = L.off_head + 16;
    L.bytes = up(L.off_floats + (size_t)capacity * 4);
    return L;
}

// one wavefront per tile: the mask of pixels with a != +0 (bitwise), its popcount; the code bytes are copied
__global__ __launch_bounds__(256) void k_sparse_masks(const uint8_t *__restrict__ wire, uint8_t *__restrict__ sparse, SparseLayout L)
{
    const uint32_t f = blockIdx.y, tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (tile >= L.tiles) return;
    const size_t npx = (size_t)L.rows * L.width;
    const uint8_t *src = wire + (size_t)f * npx * 5;
    uint8_t *dst = sparse + (size_t)f * L.bytes;
    const uint32_t x = (tile % L.tiles_x) * 8 + (lane & 7u), y = (tile / L.tiles_x) * 8 + (lane >> 3);
    const bool in = x < L.width && y < L.rows;
    uint32_t bits = 0;
    if (in) {
        const size_t l = (size_t)y * L.width + x;
        bits = reinterpret_cast<const uint32_t *>(src)[l];
        dst[l] = src[4 * npx + l];
    }
    const unsigned long long m = __ballot(bits != 0u);
    if (lane == 0) {
        reinterpret_cast<unsigned long long *>(dst + L.off_masks)[tile] = m;
        reinterpret_cast<uint32_t *>(dst + L.off_bases)[tile] = (uint32_t)__popcll(m);
    }
}
// one workgroup per frame: exclusive scan of the tile counts in place, total and overflow flag to the header
__global__ __launch_bounds__(1024) void k_sparse_scan(uint8_t *__restrict__ sparse, SparseLayout L)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    uint8_t *dst = sparse + (size_t)blockIdx.x * L.bytes;
    uint32_t *bases = reinterpret_cast<uint32_t *>(dst + L.off_bases);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t b = 0; b < L.tiles; b += 1024) {
        const uint32_t i = b + tid;
        uint32_t v = i < L.tiles ? bases[i] : 0u, x = v;
        for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o); if ((int)lane >= o) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; w++) woff += wsum[w];
        const uint32_t c = carry;
        if (i < L.tiles) bases[i] = c + woff + x - v;
        __syncthreads();
        if (tid == 1023) carry = c + woff + x;
        __syncthreads();
    }
    if (tid == 0) {
        uint32_t *head = reinterpret_cast<uint32_t *>(dst + L.off_head);
        head[0] = carry; head[1] = carry > L.capacity ? 1u : 0u; head[2] = 0; head[3] = 0;
    }
}
// one wavefront per tile: the floats of its lit pixels to their slots
__global__ __launch_bounds__(256) void k_sparse_scatter(const uint8_t *__restrict__ wire, uint8_t *__restrict__ sparse, SparseLayout L)
{
    const uint32_t f = blockIdx.y, tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (tile >= L.tiles) return;
    const size_t npx = (size_t)L.rows * L.width;
    const uint8_t *src = wire + (size_t)f * npx * 5;
    uint8_t *dst = sparse + (size_t)f * L.bytes;
    const unsigned long long m = reinterpret_cast<const unsigned long long *>(dst + L.off_masks)[tile];
    if (!((m >> lane) & 1ull)) return;
    const uint32_t slot = reinterpret_cast<const uint32_t *>(dst + L.off_bases)[tile] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (slot >= L.capacity) return;
    const uint32_t x = (tile % L.tiles_x) * 8 + (lane & 7u), y = (tile / L.tiles_x) * 8 + (lane >> 3);
    reinterpret_cast<uint32_t *>(dst + L.off_floats)[slot] = reinterpret_cast<const uint32_t *>(src)[(size_t)y * L.width + x];
}
// rank 0: gathered sparse frame-shares -> RGBA32F frames in row order
__global__ void k_deinterleave_sparse(const uint8_t *__restrict__ gathered, float4 *__restrict__ frame, uint32_t width, uint32_t height,
                                      uint32_t band_rows, uint32_t world, uint32_t frames, SparseLayout L, const BandMap M,
                                      uint32_t *overflow)
{
    size_t per_frame = (size_t)width * height, total = per_frame * frames;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t f = (uint32_t)(i / per_frame);
        size_t r = i - (size_t)f * per_frame;
        uint32_t y = (uint32_t)(r / width), x = (uint32_t)(r - (size_t)y * width);
        uint32_t band = y / band_rows, rank = band % world, lband = band / world;
        if (M.n) { const uint32_t e = M.src[band]; rank = e >> 10; lband = e & 1023u; }
        const uint32_t yl = lband * band_rows + (y - band * band_rows);
        const uint8_t *src = gathered + ((size_t)rank * frames + f) * L.bytes;
        const uint32_t code = src[(size_t)yl * width + x];
        const uint32_t tile = (yl >> 3) * L.tiles_x + (x >> 3), bit = (yl & 7u) * 8 + (x & 7u);
        const unsigned long long m = reinterpret_cast<const unsigned long long *>(src + L.off_masks)[tile];
        float a = 0.0f;
        if ((m >> bit) & 1ull) {
            const uint32_t slot = reinterpret_cast<const uint32_t *>(src + L.off_bases)[tile] + (uint32_t)__popcll(m & ((1ull << bit) - 1ull));
            if (slot < L.capacity) a = __uint_as_float(reinterpret_cast<const uint32_t *>(src + L.off_floats)[slot]);
        }
        if (x == 0 && yl == 0 && overflow && reinterpret_cast<const uint32_t *>(src + L.off_head)[1]) atomicOr(overflow, 1u);
        frame[i] = wire_expand(a, code);
    }
}

// rank 0: the sparse shares the ranks' march kernels wrote themselves (OUT_SPARSE, raymarch_kernels.h) -> frames in row
// order.  One pointer per rank (a rank's share holds all frames of the group; rank 0's own is read where it was rendered).
// MODE: RGBA32F as Compute.hlsl writes it, or through the display pass (DisplayFrag.hlsl) as RGBA8.
constexpr uint32_t MULTI_MAX_RANKS = 16;
struct ShareTable { const uint8_t *p[MULTI_MAX_RANKS]; };
template <int MODE>
__global__ __launch_bounds__(256) void k_deinterleave_sparse2(const ShareTable S, void *__restrict__ frame, uint32_t width, uint32_t height,
                                                              uint32_t band_rows, uint32_t world, Sparse2Layout L, const BandMap M,
                                                              uint32_t only_rank, uint32_t sky8, uint32_t *__restrict__ counts)
{
    // the shares' counters (header word 0), for the host: `counts` may be pinned host memory -- no copy of its own behind the frame
    if (counts && blockIdx.x == 0 && threadIdx.x < world && S.p[threadIdx.x]) counts[threadIdx.x] = *reinterpret_cast<const uint32_t *>(S.p[threadIdx.x]);
    const size_t per_frame = (size_t)width * height, total = per_frame * L.frames;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t f = (uint32_t)(i / per_frame);
        const size_t r = i - (size_t)f * per_frame;
        const uint32_t y = (uint32_t)(r / width), x = (uint32_t)(r - (size_t)y * width);
        uint32_t band = y / band_rows, rank = band % world, lband = band / world;
        if (M.n) { const uint32_t e = M.src[band]; rank = e >> 10; lband = e & 1023u; }
        if (only_rank != 0xFFFFFFFFu && rank != only_rank) continue;
        const uint32_t yl = lband * band_rows + (y - band * band_rows);
        const uint8_t *src = S.p[rank];
        const size_t ft = (size_t)f * L.tiles + (size_t)(yl >> 3) * L.tiles_x + (x >> 3);
        const uint32_t bit = (yl & 7u) * 8u + (x & 7u);
        const uint32_t code = src[L.off_codes + ft * 64 + bit];
        const unsigned long long m = reinterpret_cast<const unsigned long long *>(src + L.off_masks)[ft];
        float a = 0.0f;
        if ((m >> bit) & 1ull) {
            const uint32_t slot = reinterpret_cast<const uint32_t *>(src + L.off_bases)[ft] + (uint32_t)__popcll(m & ((1ull << bit) - 1ull));
            if (slot < L.capacity) a = reinterpret_cast<const float *>(src + L.off_floats)[slot];
        }
        // (the frame is written once and read by somebody else: past the caches -- the lines stay with the grid cells of the groups
        // that march meanwhile; 4K, one rank 0.364 -> 0.351 ms per frame, four ranks 0.388 -> 0.382)
        if (MODE == OUT_RGBA32F) {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const float4 v = wire_expand(a, code);
            __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4 *>(frame) + i);
        }
        else {
            const float4 v = wire_expand(a, code);
            uint32_t q;
            if (MODE == OUT_HEAT8) q = heat8(v.w);
            else if (code > 140u) q = sky8 | alpha8(v.w);
            else { const uint32_t g = gamma8(a); q = g | (g << 8) | (g << 16) | alpha8(v.w); }
            __builtin_nontemporal_store(q, reinterpret_cast<uint32_t *>(frame) + i);
        }
    }
}

__global__ void k_unorm_table(float *out)
{
    out[threadIdx.x] = unorm8((float)threadIdx.x);
}

constexpr int MAX_TOP_LEVEL = 8;
#ifndef FULL_GRID_SHARE
#define FULL_GRID_SHARE 64          // a grid as deep as the tree may take 1/FULL_GRID_SHARE of the device's memory
#endif
#ifndef DENSE_GRID_MAX_BYTES
#define DENSE_GRID_MAX_BYTES (512ull << 20)   // ... and by default no more than this (depth <= 8): deeper trees get a split grid
#endif

}  // namespace sdfhip

// ===============================================================================
// Host side of the device half
// ===============================================================================
using namespace sdfhip;

struct sdfhip_scene {
    int device;
    uint32_t n, depth;
    int stack_ok;
    void *alloc;            // hipMalloc'ed block holding the records
    NodeRec *nodes;         // = alloc + 112: node 1 (first sibling block) starts a 128-B line
    hipStream_t stream;
    TopCell *d_top;                  // top grid of the cursor-stack kernels (raymarch_device.h), or null
    int top_level;
    TopCell *d_fine;                 // split grid: blocks of fine cells below the internal cells of d_top, or null
    int fine_bits;
    uint64_t fine_bytes;
    uint32_t *d_d4;                  // the grid's second form, for the default kernel's loop (CursorFF, raymarch_device.h): one word per
    uint4 *d_recs;                   // cell of the deepest level + the sample records of the non-flat leaves; or null
    uint64_t d4_bytes;
    // A second, split grid beside a dense full-depth one, for the kernels whose rays are incoherent (the bounce levels of
    // the path-traced pipeline are HBM-bound: every lookup in the 8^depth-cell dense grid is a cache miss, while a ray
    // that stays near the surface stays inside one block of fine cells).  Same cells, same cursor; built on the first
    // path-traced render.
    TopCell *d_top2, *d_fine2;
    int top2_level, fine2_bits, fine2_order, scatter_tried;
    uint64_t top2_bytes;
    size_t total_mem;
    uint32_t *d_verdict;             // k_validate's two words (upload)
    // Per-stream scratch of the render launches: the hit queues of the two-kernel pipeline and their
    // control words, and the tile-queue heads of the compact kernel.  Launches on one stream run in
    // order and may share a scratch; launches on different streams overlap (frames in flight) and
    // must not, so every stream that renders on this handle gets its own.
    struct Scratch {
        hipStream_t stream;
        char *hit_buf;               // hit_a | hit_b | hit_c, `records` each
        size_t records;
        uint32_t *ctl;               // hit fill counts (two sets), then the compact kernel's tile queues
        uint32_t launches;           // two-kernel launch pairs so far: its parity selects the set of fill counts
        uint64_t last_use;           // the handle's render count when this scratch was last handed out (the oldest idle one is recycled)
        hipEvent_t idle;             // the library's own event behind the last launch that used this scratch: "is it idle?" never asks the
                                     // caller's stream handle, which may have been destroyed since
        char *pt_buf;                // path-traced pipeline: two hit queues, then the per-path results
        size_t pt_bytes;
        // SDFHIP_FLAG_TILE_ORDER: the wave-iterations of every tile of the last frame rendered on this stream, the launch order
        // made from them for the next one, and the frame geometry both belong to
        uint16_t *ord_cost;
        uint8_t *ord_class;
        uint32_t *ord_perm;
        uint32_t ord_tiles, ord_blocks;      // capacity of the two arrays
        uint32_t ord_sig[8];                 // width, height, nrows_out, band_rows, band_first, band_stride, n_band_list, hash of the list
        bool ord_valid;
        sdfhip_info ord_info;                // the camera block of the frame the order was made from
    };
    static constexpr int MAX_SCRATCH = 16;
    // ... then the counters of SDFHIP_FLAG_COUNT renders on this stream, 16 x u64: nodes, samples, steps, shadow rays, loads,
    // hits, and from [6] the step classes of sdfhip_debug_step_classes (per stream: counting renders on two streams of one
    // handle do not add into each other's figures)
    static constexpr size_t CTL_HIT_WORDS = (size_t)2 * MAX_BATCH * HIT_QUEUES * 32, CTL_QUEUE_WORDS = 8 * 32,
                            CTL_PT_WORDS = (size_t)2 * HIT_QUEUES * 32 + 32 /* + the overflow word's line */, CTL_COUNTER_WORDS = 32;
    Scratch scratch[MAX_SCRATCH];
    int n_scratch;
    uint64_t uses;
    float4 *d_frame;        // grown on demand by sdfhip_render
    size_t frame_cap;
    hipEvent_t ev0, ev1;
    // sdfhip_render (the host frame): the frame in HOST_BANDS row bands, each on its own stream (its own scratch: the launch
    // order of SDFHIP_FLAG_TILE_ORDER is kept per stream) behind the band before it; a band's copy to the host runs while
    // the next bands march
    static constexpr int HOST_BANDS = 4;
    hipStream_t band_stream[HOST_BANDS];
    hipEvent_t band_done[HOST_BANDS];
    int cu_count;
    const uint32_t *dbg_tile_perm;   // sdfhip_debug_tile_order: experiment hooks for k_march
    uint16_t *dbg_tile_cost;
    std::mutex lock;        // render on one handle is single-caller; this makes misuse safe
};

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(SDFHIP_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

namespace {
// Keeps the caller's current device intact (the host process may be PyTorch).
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

extern "C" int sdfhip_device_count(int *count)
{
    if (!count) return fail(SDFHIP_ERR_ARG, "device_count: null argument");
    *count = 0;
    HIP_TRY(hipGetDeviceCount(count));
    return SDFHIP_OK;
}

extern "C" int sdfhip_scene_free(sdfhip_scene *s)
{
    if (!s) return SDFHIP_OK;
    {
        DeviceGuard g(s->device);
        if (s->stream) (void)hipStreamSynchronize(s->stream);
        if (s->alloc) (void)hipFree(s->alloc);
        if (s->d_verdict) (void)hipFree(s->d_verdict);
        if (s->d_top) (void)hipFree(s->d_top);
        if (s->d_fine) (void)hipFree(s->d_fine);
        if (s->d_d4) (void)hipFree(s->d_d4);
        if (s->d_recs) (void)hipFree(s->d_recs);
        if (s->d_top2) (void)hipFree(s->d_top2);
        if (s->d_fine2) (void)hipFree(s->d_fine2);
        for (int i = 0; i < s->n_scratch; i++) {
            if (s->scratch[i].hit_buf) (void)hipFree(s->scratch[i].hit_buf);
            if (s->scratch[i].pt_buf) (void)hipFree(s->scratch[i].pt_buf);
            if (s->scratch[i].ord_cost) (void)hipFree(s->scratch[i].ord_cost);
            if (s->scratch[i].ord_class) (void)hipFree(s->scratch[i].ord_class);
            if (s->scratch[i].ord_perm) (void)hipFree(s->scratch[i].ord_perm);
            if (s->scratch[i].ctl) (void)hipFree(s->scratch[i].ctl);
            if (s->scratch[i].idle) (void)hipEventDestroy(s->scratch[i].idle);
        }
        if (s->d_frame) (void)hipFree(s->d_frame);
        for (int b = 0; b < sdfhip_scene::HOST_BANDS; b++) {
            if (s->band_stream[b]) { (void)hipStreamSynchronize(s->band_stream[b]); (void)hipStreamDestroy(s->band_stream[b]); }
            if (s->band_done[b]) (void)hipEventDestroy(s->band_done[b]);
        }
        if (s->ev0) (void)hipEventDestroy(s->ev0);
        if (s->ev1) (void)hipEventDestroy(s->ev1);
        if (s->stream) (void)hipStreamDestroy(s->stream);
    }
    delete s;
    return SDFHIP_OK;
}

// A split grid over the scene's records: dense cells of level C whose internal cells (level word 15) name, in `children`,
// the first cell of a block of 8^FB fine cells; built on s->stream.  false (nothing allocated) when memory or the byte limit say no.
static bool build_split_grid(sdfhip_scene *s, int C, int FB, int order, uint64_t max_fine_bytes, TopCell **coarse_out, TopCell **fine_out,
                             uint64_t *fine_bytes_out)
{
    const size_t ncell = (size_t)1 << (3 * C);
    const uint32_t n_chunks = (uint32_t)((ncell + 255) / 256);
    uint32_t *d_block_node = nullptr, *d_chunks = nullptr;
    TopCell *d_coarse = nullptr, *d_fine = nullptr;
    bool ok = false;
    do {
        if (hipMalloc((void **)&d_coarse, ncell * sizeof(TopCell)) != hipSuccess) break;
        if (hipMalloc((void **)&d_chunks, ((size_t)n_chunks + 1) * 4) != hipSuccess) break;
        const uint32_t tb = (uint32_t)((ncell + 255) / 256 < 8192 ? (ncell + 255) / 256 : 8192);
        hipLaunchKernelGGL(k_top_grid, dim3(tb), dim3(256), 0, s->stream, s->nodes, d_coarse, C, 2);
        // number the internal cells in cell order (deterministic: a prefix sum) and count them
        const uint32_t cb = n_chunks < 16384u ? n_chunks : 16384u;
        hipLaunchKernelGGL(k_split_count, dim3(cb), dim3(256), 0, s->stream, d_coarse, (uint32_t)ncell, n_chunks, d_chunks);
        hipLaunchKernelGGL(k_split_scan, dim3(1), dim3(1024), 0, s->stream, d_chunks, n_chunks);
        uint32_t nb32 = 0;
        if (hipMemcpyAsync(&nb32, d_chunks + n_chunks, 4, hipMemcpyDeviceToHost, s->stream) != hipSuccess) break;
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) break;
        const size_t nblocks = nb32;
        const uint64_t fine_bytes = (uint64_t)(nblocks << (3 * FB)) * sizeof(TopCell);
        if (fine_bytes > max_fine_bytes || (nblocks << (3 * FB)) >= ((size_t)1 << 31)) break;
        if (nblocks) {
            if (hipMalloc((void **)&d_fine, fine_bytes) != hipSuccess) break;
            if (hipMalloc((void **)&d_block_node, nblocks * 4) != hipSuccess) break;
            hipLaunchKernelGGL(k_split_assign, dim3(cb), dim3(256), 0, s->stream, d_coarse, (uint32_t)ncell, n_chunks, d_chunks, d_block_node, FB);
            const size_t nfine = nblocks << (3 * FB);
            const uint32_t fb = (uint32_t)((nfine + 255) / 256 < 16384 ? (nfine + 255) / 256 : 16384);
            hipLaunchKernelGGL(k_fine_blocks, dim3(fb), dim3(256), 0, s->stream, s->nodes, d_block_node, d_fine,
                               (uint32_t)nblocks, C, FB, order);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) break;
        }
        *coarse_out = d_coarse; d_coarse = nullptr;
        *fine_out = d_fine; d_fine = nullptr;
        *fine_bytes_out = nblocks ? fine_bytes : 0;
        ok = true;
    } while (false);
    (void)hipGetLastError();
    if (d_block_node) (void)hipFree(d_block_node);
    if (d_chunks) (void)hipFree(d_chunks);
    if (d_coarse) (void)hipFree(d_coarse);
    if (d_fine) (void)hipFree(d_fine);
    return ok;
}

// The grid's second form (CursorFF, raymarch_device.h), made from the 16-byte cells once they exist -- a dense grid as deep as the
// tree, or a split one: a word per cell of the deepest level and a 64-byte sample record per non-flat leaf.  An accelerator of an
// accelerator: trees deeper than 10 levels (4 GB of words at depth 10), a grid that is not as deep as the tree, or too little
// memory do without it, and the default kernel reads the 16-byte cells.
// MEASURED SLOWER than the 16-byte cells on the bench frames (30 % fewer VALU instructions, 1.7 x the L1 tag
// lookups and 2.7 x the HBM bytes; 0.112 against 0.090 ms per 1080p frame), so it is built only when SDFHIP_SAMPLE_RECORDS=1 is in
// the environment at upload: an experiment that stays bit-identical (tests/test_gpu_parity.py::test_pre_decoded_cells...).
static void build_dense4(sdfhip_scene *s)
{
    const int F = (int)s->depth;
    if (!s->stack_ok || !s->d_top || F < 1 || F > 10 || s->top_level + s->fine_bits != F) return;
    const char *env = getenv("SDFHIP_SAMPLE_RECORDS");
    if (!env || atoi(env) != 1) return;
    const size_t ncell = (size_t)1 << (3 * F);
    if (ncell * 4 > s->total_mem / 32) return;
    const GridRef g{s->d_top, s->d_fine, s->top_level, s->fine_bits, 0};
    const uint32_t n_chunks = (uint32_t)((ncell + 255) / 256);
    uint32_t *d4 = nullptr, *d_chunks = nullptr;
    uint4 *recs = nullptr;
    do {
        if (hipMalloc((void **)&d4, ncell * 4) != hipSuccess) break;
        if (hipMalloc((void **)&d_chunks, ((size_t)n_chunks + 1) * 4) != hipSuccess) break;
        const uint32_t tb = (uint32_t)((ncell + 255) / 256 < 16384 ? (ncell + 255) / 256 : 16384), cb = n_chunks < 16384u ? n_chunks : 16384u;
        hipLaunchKernelGGL(k_d4_fill, dim3(tb), dim3(256), 0, s->stream, g, d4, F);
        hipLaunchKernelGGL(k_d4_count, dim3(cb), dim3(256), 0, s->stream, d4, ncell, n_chunks, d_chunks);
        hipLaunchKernelGGL(k_split_scan, dim3(1), dim3(1024), 0, s->stream, d_chunks, n_chunks);
        uint32_t n_rec = 0;
        if (hipMemcpyAsync(&n_rec, d_chunks + n_chunks, 4, hipMemcpyDeviceToHost, s->stream) != hipSuccess) break;
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) break;
        if ((uint64_t)n_rec * 4 >= 0x3F000000ull) break;                          // TAG + index must stay a finite float
        if ((uint64_t)n_rec * 64 > s->total_mem / 16) break;
        if (hipMalloc((void **)&recs, ((size_t)n_rec + 1) * 64) != hipSuccess) break;
        hipLaunchKernelGGL(k_d4_anchor, dim3(cb), dim3(256), 0, s->stream, g, d4, ncell, n_chunks, d_chunks, recs, F);
        hipLaunchKernelGGL(k_d4_share, dim3(tb), dim3(256), 0, s->stream, g, d4, F);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) break;
        s->d_d4 = d4; d4 = nullptr;
        s->d_recs = recs; recs = nullptr;
        s->d4_bytes = (uint64_t)ncell * 4 + ((uint64_t)n_rec + 1) * 64;
    } while (false);
    (void)hipGetLastError();
    if (d4) (void)hipFree(d4);
    if (recs) (void)hipFree(recs);
    if (d_chunks) (void)hipFree(d_chunks);
}

extern "C" int sdfhip_scene_upload(int device, const int32_t *structs, const uint8_t *values,
                                   uint32_t n, sdfhip_scene **out)
{
    return sdfhip::scene_from_arrays(device, structs, values, n, false, out);
}

// structs / values on the host (sdfhip_scene_upload), or already in `device`'s memory (sdfhip_sdfgen_scene: the tree the GPU
// builder has just made never leaves HBM)
int sdfhip::scene_from_arrays(int device, const int32_t *structs, const uint8_t *values, uint32_t n, bool resident, sdfhip_scene **out)
{
    if (!structs || !values || !out || n == 0)
        return fail(SDFHIP_ERR_ARG, "scene_upload: null argument or empty scene");
    *out = nullptr;
    uint32_t depth = 0;
    int consistent = 0;

    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail(SDFHIP_ERR_DEVICE, "scene_upload: device %d of %d does not exist", device, ndev);
    DeviceGuard g(device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "scene_upload: hipSetDevice(%d) failed", device);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SDFHIP_ERR_DEVICE, "scene_upload: device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);

    sdfhip_scene *s = new (std::nothrow) sdfhip_scene();
    if (!s) return fail(SDFHIP_ERR_NOMEM, "scene_upload: out of host memory");
    s->device = device; s->n = n; s->depth = 0; s->stack_ok = 0;       // (both set once the tree has been validated, below)
    s->alloc = nullptr; s->nodes = nullptr; s->stream = nullptr; s->d_verdict = nullptr; s->d_top = nullptr; s->top_level = 0; s->d_fine = nullptr; s->fine_bits = 0; s->fine_bytes = 0; s->d_d4 = nullptr; s->d_recs = nullptr; s->d4_bytes = 0;
    s->d_top2 = s->d_fine2 = nullptr; s->top2_level = s->fine2_bits = s->fine2_order = s->scatter_tried = 0; s->top2_bytes = 0; s->total_mem = prop.totalGlobalMem;
    s->n_scratch = 0; s->uses = 0; s->d_frame = nullptr; s->dbg_tile_perm = nullptr; s->dbg_tile_cost = nullptr; s->frame_cap = 0; s->ev0 = s->ev1 = nullptr;
    s->cu_count = prop.multiProcessorCount;
    for (int b = 0; b < sdfhip_scene::HOST_BANDS; b++) { s->band_stream[b] = nullptr; s->band_done[b] = nullptr; }

    void *d_s = nullptr, *d_v = nullptr;
    auto bail = [&](hipError_t e, const char *what) {
        if (d_s && !resident) (void)hipFree(d_s);
        if (d_v && !resident) (void)hipFree(d_v);
        sdfhip_scene_free(s);
        return fail(SDFHIP_ERR_DEVICE, "scene_upload: %s failed: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
    const size_t bytes = (size_t)n * 8;
    if ((e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    if ((e = hipEventCreate(&s->ev0)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreate(&s->ev1)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipMalloc(&s->alloc, (size_t)n * 16 + 128)) != hipSuccess) return bail(e, "hipMalloc(records)");
    s->nodes = reinterpret_cast<NodeRec *>(static_cast<char *>(s->alloc) + 112);
    if ((e = hipMalloc((void **)&s->d_verdict, 2 * sizeof(uint32_t))) != hipSuccess) return bail(e, "hipMalloc(verdict)");
    if (resident) {
        d_s = const_cast<int32_t *>(structs); d_v = const_cast<uint8_t *>(values);
    } else {
        if ((e = hipMalloc(&d_s, bytes)) != hipSuccess) return bail(e, "hipMalloc(structs)");
        if ((e = hipMalloc(&d_v, bytes)) != hipSuccess) return bail(e, "hipMalloc(values)");
        if ((e = hipMemcpyAsync(d_s, structs, bytes, hipMemcpyHostToDevice, s->stream)) != hipSuccess) return bail(e, "hipMemcpy(structs)");
        if ((e = hipMemcpyAsync(d_v, values, bytes, hipMemcpyHostToDevice, s->stream)) != hipSuccess) return bail(e, "hipMemcpy(values)");
    }
    {   // validation (sdfhip_octdata_validate's verdicts, on the device: see k_validate)
        uint32_t *d_verdict = s->d_verdict, verdict[2] = { 0u, 0u };
        if ((e = hipMemsetAsync(d_verdict, 0, sizeof verdict, s->stream)) != hipSuccess) return bail(e, "hipMemset(verdict)");
        hipLaunchKernelGGL(k_validate, dim3((n + 255u) / 256u), dim3(256), 0, s->stream, (const int2 *)d_s, n, d_verdict);
        if ((e = hipGetLastError()) != hipSuccess) return bail(e, "k_validate launch");
        if ((e = hipMemcpyAsync(verdict, d_verdict, sizeof verdict, hipMemcpyDeviceToHost, s->stream)) != hipSuccess) return bail(e, "hipMemcpy(verdict)");
        if ((e = hipStreamSynchronize(s->stream)) != hipSuccess) return bail(e, "k_validate");
        if (verdict[0] & 1u) {                          // a bad link: the host function finds it again and says which
            if (!resident) { (void)hipFree(d_s); (void)hipFree(d_v); }
            d_s = d_v = nullptr;
            sdfhip_scene_free(s);
            if (resident) return fail(SDFHIP_ERR_BAD_TREE, "scene_from_arrays: the tree has a link out of range or an endless parent chain");
            const int rcv = sdfhip_octdata_validate(structs, n, &depth, &consistent);
            return rcv != SDFHIP_OK ? rcv : fail(SDFHIP_ERR_BAD_TREE, "scene_upload: the tree has a link out of range or an endless parent chain");
        }
        consistent = (verdict[0] & 2u) ? 0 : 1;
        depth = consistent ? verdict[1] : 0xFFFFFFFFu;
        s->depth = depth;
        s->stack_ok = (consistent && depth <= (uint32_t)MAX_STACK) ? 1 : 0;
    }
    uint32_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_fuse, dim3(blocks), dim3(256), 0, s->stream, (const int2 *)d_s,
                       (const uint2 *)d_v, s->nodes, n);
    if ((e = hipGetLastError()) != hipSuccess) return bail(e, "k_fuse launch");
    // Top grid.  Dense and as deep as the tree -- every leaf in the grid, a find is one load -- for trees of depth
    // <= 8 (268 MB).  A deeper tree gets a split grid (below): the dense grid of a depth-9 tree is 2.1 GB, a frame
    // touches a quarter of it, and two 1080p frames in flight are HBM-bound on that traffic; the split grid moves a
    // third of the bytes for 14 % more instructions -- faster where the dense grid is HBM-bound (1080p pipelined
    // 0.092 vs 0.097 ms, 0.097 vs 0.106 with a moving camera), 2-3 % slower where it is not (4K, one frame alone),
    // in a third of the memory.  SDFHIP_TOP_GRID_LEVEL=<depth> asks for the dense grid (it must fit 1/64 of the
    // device's memory), smaller values for a partial top grid (cursor-stack kernels): at most MAX_TOP_LEVEL and by
    // default no larger than the tree's own records (16 bytes per cell and per node; at least 64 KB); 0 = none.
    int top_level = 0;
    const uint64_t dense_bytes = depth <= 10 ? ((uint64_t)sizeof(TopCell) << (3 * depth)) : ~0ull;
    const char *env_level = getenv("SDFHIP_TOP_GRID_LEVEL");
    const bool dense_asked = env_level && atoi(env_level) >= (int)depth && dense_bytes <= prop.totalGlobalMem / FULL_GRID_SHARE;
    if (depth >= 1 && depth <= 10 && (dense_bytes <= DENSE_GRID_MAX_BYTES || dense_asked)) {
        top_level = (int)depth;
    } else {
        const size_t budget = (size_t)n * 16 > ((size_t)1 << 16) ? (size_t)n * 16 : ((size_t)1 << 16);
        while (top_level < MAX_TOP_LEVEL && (uint32_t)top_level < depth &&
               (sizeof(TopCell) << (3 * (top_level + 1))) <= budget)
            top_level++;
    }
    if (const char *env = getenv("SDFHIP_TOP_GRID_LEVEL")) {
        const int v = atoi(env);
        if (v >= 0 && v <= 10) top_level = v < (int)depth ? v : (int)depth;
    }
    // Split grid for trees too deep for a dense grid of their depth (10-12 levels): a dense coarse level C whose
    // internal cells point at dense blocks of the remaining FB = depth - C levels.  Every leaf is one or two
    // loads away (CursorF kernels), the coarse level stays cache-resident, and the blocks exist only where the
    // tree is deep.  Taken when the blocks fit 1/16 of the device's memory; SDFHIP_TOP_GRID_SPLIT=C forces a
    // coarse level (0 = never).
    int split = 0;
    if ((uint32_t)top_level < depth && depth <= (uint32_t)LM) {
        // coarse level: as deep as 8, no larger than the tree's own records, leaving at most 4 levels to the blocks
        const size_t budget = (size_t)n * 16 > ((size_t)1 << 16) ? (size_t)n * 16 : ((size_t)1 << 16);
        int C = 0;
        while (C < MAX_TOP_LEVEL && C + 1 < (int)depth && (sizeof(TopCell) << (3 * (C + 1))) <= budget) C++;
        if (C >= 1 && (int)depth - C <= 4) split = C;
    }
    if (const char *env = getenv("SDFHIP_TOP_GRID_SPLIT")) {
        const int v = atoi(env);
        split = (v >= 1 && v < (int)depth && (int)depth - v <= 6 && v <= 8) ? v : 0;
    }
    if (getenv("SDFHIP_TOP_GRID_LEVEL")) split = getenv("SDFHIP_TOP_GRID_SPLIT") ? split : 0;   // an explicit level means a plain grid
    bool split_built = false;
    if (s->stack_ok && split > 0) {
        TopCell *coarse = nullptr, *fine = nullptr;
        uint64_t fbytes = 0;
        if (build_split_grid(s, split, (int)depth - split, 0, prop.totalGlobalMem / 16, &coarse, &fine, &fbytes)) {
            s->d_top = coarse; s->d_fine = fine;
            s->top_level = split; s->fine_bits = (int)depth - split; s->fine_bytes = fbytes;
            split_built = true;
        }
    }
    if (s->stack_ok && top_level > 0 && !split_built) {
        // the grid is an accelerator, not part of the scene: without memory for it, shrink it
        while (top_level > 0 && hipMalloc((void **)&s->d_top, sizeof(TopCell) << (3 * top_level)) != hipSuccess) {
            (void)hipGetLastError();
            s->d_top = nullptr;
            top_level--;
        }
        if (top_level > 0) {
            const size_t ncell = (size_t)1 << (3 * top_level);
            s->top_level = top_level;
            const uint32_t tb = (uint32_t)((ncell + 255) / 256 < 8192 ? (ncell + 255) / 256 : 8192);
            hipLaunchKernelGGL(k_top_grid, dim3(tb), dim3(256), 0, s->stream, s->nodes, s->d_top, top_level,
                               (uint32_t)top_level >= depth ? 1 : 0);
            if ((e = hipGetLastError()) != hipSuccess) return bail(e, "k_top_grid launch");
        }
    }
    if ((e = hipStreamSynchronize(s->stream)) != hipSuccess) return bail(e, "k_fuse");
    build_dense4(s);
    if (!resident) { (void)hipFree(d_s); (void)hipFree(d_v); }
    d_s = nullptr; d_v = nullptr;
    *out = s;
    return SDFHIP_OK;
}

extern "C" int sdfhip_scene_top_grid(const sdfhip_scene *s, int32_t *level, uint64_t *bytes)
{
    if (!s) return fail(SDFHIP_ERR_ARG, "scene_top_grid: null scene");
    if (level) *level = s->d_top ? s->top_level : 0;
    if (bytes) *bytes = s->d_top ? ((uint64_t)sizeof(TopCell) << (3 * s->top_level)) + s->fine_bytes + s->d4_bytes + s->top2_bytes : 0;
    return SDFHIP_OK;
}

bool sdfhip::scene_has_full_depth_grid(const sdfhip_scene *s)
{
    return s && s->stack_ok && s->d_top && (s->fine_bits || (uint32_t)s->top_level >= s->depth);     // render_impl's `two`
}

extern "C" int sdfhip_scene_info(const sdfhip_scene *s, uint32_t *n, uint32_t *depth,
                                 int *stack_kernel_ok, int *device)
{
    if (!s) return fail(SDFHIP_ERR_ARG, "scene_info: null scene");
    if (n) *n = s->n;
    if (depth) *depth = s->depth;
    if (stack_kernel_ok) *stack_kernel_ok = s->stack_ok;
    if (device) *device = s->device;
    return SDFHIP_OK;
}

// Beside the scene's own grid, the bounce levels of the path-traced pipeline read a split grid of the same cells with larger,
// sub-cube-ordered blocks -- unless the scene's grid already has that coarse level.  Built once: by
// sdfhip_scene_prepare_path (at load time: allocations and two stream synchronisations), or else in front of the first
// path-traced render, before its clock starts.  SDFHIP_SCATTER_GRID=0 turns it off, 1..4 sets the blocks' levels (default 4, 3 for shallow trees);
// SDFHIP_SCATTER_ORDER=0 stores the blocks in x-y-z order.  Without memory for it (1/32 of the device's) the bounce levels
// read the scene's own grid.
static void ensure_scatter_grid(sdfhip_scene *s)
{
    if (s->scatter_tried) return;
    s->scatter_tried = 1;
    if (!s->stack_ok || !s->d_top || !((s->fine_bits && s->d_fine) || (uint32_t)s->top_level >= s->depth)) return;   // the pipeline needs a full-depth grid
    const char *env = getenv("SDFHIP_SCATTER_GRID");
    // blocks of 8^FB fine cells; 0 = off.  Default 16^3-cell blocks (64 KB each) for trees of depth 6 and more: cfg-5 21.9 ms
    // against 22.3 with 8^3 (and 24.6 with 4^3) for 1.00 instead of 0.83 GB at depth 9, on a 288 GB device
    // (without the variable, blocks that do not fit the memory share fall back to the next smaller size)
    for (int FB = env ? atoi(env) : ((int)s->depth >= 6 ? 4 : 3); FB >= 1; FB = env ? 0 : FB - 1) {
        if (!(FB <= 4 && (int)s->depth - FB >= 1 && (int)s->depth - FB <= MAX_TOP_LEVEL)) continue;
        if (s->fine_bits && s->top_level == (int)s->depth - FB) return;            // the scene's own grid is that grid
        uint64_t fbytes = 0;
        s->fine2_order = (FB >= 2 && !(getenv("SDFHIP_SCATTER_ORDER") && atoi(getenv("SDFHIP_SCATTER_ORDER")) == 0)) ? 1 : 0;
        if (build_split_grid(s, (int)s->depth - FB, FB, s->fine2_order, s->total_mem / 32, &s->d_top2, &s->d_fine2, &fbytes)) {
            s->top2_level = (int)s->depth - FB; s->fine2_bits = FB;
            s->top2_bytes = ((uint64_t)sizeof(TopCell) << (3 * s->top2_level)) + fbytes;
            return;
        }
    }
}

extern "C" int sdfhip_scene_prepare_path(sdfhip_scene *s)
{
    if (!s) return fail(SDFHIP_ERR_ARG, "scene_prepare_path: null scene");
    std::lock_guard<std::mutex> lk(s->lock);
    DeviceGuard g(s->device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "scene_prepare_path: hipSetDevice(%d) failed", s->device);
    ensure_scatter_grid(s);
    return SDFHIP_OK;
}

namespace {

// The scratch of stream `st` on this scene, with room for `records` hit records (0: control words only).
// Created on a stream's first render; grown (after the stream has drained) when a larger frame comes.
int get_scratch(sdfhip_scene *s, hipStream_t st, size_t records, sdfhip_scene::Scratch **out)
{
    sdfhip_scene::Scratch *sc = nullptr;
    for (int i = 0; i < s->n_scratch; i++)
        if (s->scratch[i].stream == st) sc = &s->scratch[i];
    if (!sc && s->n_scratch == sdfhip_scene::MAX_SCRATCH) {
        // every slot is taken (a host that makes a stream per frame gets here after 16 frames): the least recently used scratch whose
        // stream has drained goes to the new stream -- its buffers stay, its state (launch parity, tile order, counters) starts afresh.
        // (A destroyed stream whose handle value HIP hands out again finds its old slot above: harmless for the buffers, and the tile
        // order is dropped by the geometry / camera check or is simply a valid order of the same tiles.)
        int best = -1;
        for (int i = 0; i < s->n_scratch; i++) {
            // (the slot's own event, not hipStreamQuery on its stream: a host that makes a stream per frame destroys them, and a
            // stale handle must not be handed back to HIP)
            if (hipEventQuery(s->scratch[i].idle) != hipSuccess) { (void)hipGetLastError(); continue; }
            if (best < 0 || s->scratch[i].last_use < s->scratch[best].last_use) best = i;
        }
        if (best < 0)
            return fail(SDFHIP_ERR_ARG, "render: %d streams have renders in flight on one scene handle (at most %d at a time)", sdfhip_scene::MAX_SCRATCH, sdfhip_scene::MAX_SCRATCH);
        sc = &s->scratch[best];
        sc->stream = st; sc->launches = 0; sc->ord_valid = false; memset(sc->ord_sig, 0, sizeof sc->ord_sig);
        const size_t ctl_bytes = (sdfhip_scene::CTL_HIT_WORDS + sdfhip_scene::CTL_QUEUE_WORDS + sdfhip_scene::CTL_PT_WORDS +
                                  sdfhip_scene::CTL_COUNTER_WORDS) * sizeof(uint32_t);
        HIP_TRY(hipMemsetAsync(sc->ctl, 0, ctl_bytes, st));
    }
    if (!sc) {
        sc = &s->scratch[s->n_scratch];
        sc->stream = st; sc->hit_buf = nullptr; sc->records = 0; sc->ctl = nullptr; sc->launches = 0; sc->pt_buf = nullptr; sc->pt_bytes = 0; sc->idle = nullptr;
        sc->ord_cost = nullptr; sc->ord_class = nullptr; sc->ord_perm = nullptr; sc->ord_tiles = sc->ord_blocks = 0; sc->ord_valid = false; memset(sc->ord_sig, 0, sizeof sc->ord_sig);
        const size_t ctl_bytes = (sdfhip_scene::CTL_HIT_WORDS + sdfhip_scene::CTL_QUEUE_WORDS + sdfhip_scene::CTL_PT_WORDS +
                                  sdfhip_scene::CTL_COUNTER_WORDS) * sizeof(uint32_t);
        HIP_TRY(hipEventCreateWithFlags(&sc->idle, hipEventDisableTiming));
        { const hipError_t em = hipMalloc((void **)&sc->ctl, ctl_bytes); if (em != hipSuccess) { (void)hipEventDestroy(sc->idle); return fail(SDFHIP_ERR_DEVICE, "hipMalloc(scratch) failed: %s", hipGetErrorString(em)); } }
        // zeroed IN the stream that will use it: a hipMemset on the null stream is not ordered against a non-blocking
        // stream (the first frame on a new scratch would, now and then, have met counters that were not zero yet)
        HIP_TRY(hipMemsetAsync(sc->ctl, 0, ctl_bytes, st));
        s->n_scratch++;
    }
    sc->last_use = ++s->uses;
    if (records > sc->records) {
        HIP_TRY(hipStreamSynchronize(st));
        if (sc->hit_buf) { (void)hipFree(sc->hit_buf); sc->hit_buf = nullptr; sc->records = 0; }
        HIP_TRY(hipMalloc((void **)&sc->hit_buf, records * 64));
        sc->records = records;
    }
    *out = sc;
    return SDFHIP_OK;
}

// the path-traced pipeline's buffers on a stream's scratch
int get_pt_scratch(sdfhip_scene *s, hipStream_t st, size_t bytes, sdfhip_scene::Scratch **out)
{
    int rc = get_scratch(s, st, 0, out);
    if (rc != SDFHIP_OK) return rc;
    sdfhip_scene::Scratch *sc = *out;
    if (bytes > sc->pt_bytes) {
        HIP_TRY(hipStreamSynchronize(st));
        if (sc->pt_buf) { (void)hipFree(sc->pt_buf); sc->pt_buf = nullptr; sc->pt_bytes = 0; }
        HIP_TRY(hipMalloc((void **)&sc->pt_buf, bytes));
        sc->pt_bytes = bytes;
    }
    return SDFHIP_OK;
}

// path-traced mode as a pipeline: camera segments, one kernel per bounce level, the ordered sum
template <int CUR, bool COUNT>
int launch_pt(sdfhip_scene *s, sdfhip_scene::Scratch *sc, dim3 grid, hipStream_t st, RenderParams &P)
{
    // (a multiple of 64: the bounce waves' pushes then spread evenly over the 64 sub-queues, which is what their capacity assumes)
    const uint32_t resident = ((uint32_t)s->cu_count * 32u + 63u) & ~63u;
    hipError_t e;
    if ((e = hipMemsetAsync(P.pt_ctl, 0, sdfhip_scene::CTL_PT_WORDS * sizeof(uint32_t), st)) != hipSuccess)
        return fail(SDFHIP_ERR_DEVICE, "render_path: hipMemsetAsync failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL((k_pt_primary<CUR, COUNT>), grid, dim3(64), 0, st, P);
    for (uint32_t b = 0; b <= P.pt_bounces; b++) {
        P.pt_level = b;
        // the queue this level fills was drained by the level before it
        if (b > 0 && (e = hipMemsetAsync(P.pt_ctl + (size_t)((b & 1u) ^ 1u) * HIT_QUEUES * 32, 0, HIT_QUEUES * 32 * sizeof(uint32_t), st)) != hipSuccess)
            return fail(SDFHIP_ERR_DEVICE, "render_path: hipMemsetAsync failed: %s", hipGetErrorString(e));
        if (s->d_top2) {
            // incoherent rays: the same cells through the split grid (the cursor does not depend on the grid it was filled from)
            RenderParams P2 = P;
            P2.top = s->d_top2; P2.top_level = s->top2_level; P2.fine = s->d_fine2; P2.fine_bits = s->fine2_bits; P2.fine_order = s->fine2_order;
            hipLaunchKernelGGL((k_pt_bounce<CUR_STACK_SPLIT, COUNT>), dim3(resident), dim3(64), 0, st, P2);
        } else {
            hipLaunchKernelGGL((k_pt_bounce<CUR, COUNT>), dim3(resident), dim3(64), 0, st, P);
        }
    }
    const size_t npx = (size_t)P.nrows_out * P.width;
    const uint32_t rb = (uint32_t)((npx + 255) / 256 < 4096 ? (npx + 255) / 256 : 4096);
    hipLaunchKernelGGL((k_pt_resolve<COUNT>), dim3(rb), dim3(256), 0, st, P);
    (void)sc;
    return SDFHIP_OK;
}

// k_march + k_shadow for one cursor kind and counting choice, by output mode
template <int CUR, bool COUNT>
void launch_two(uint32_t mode, dim3 grid, dim3 shade_grid, hipStream_t st, const RenderParams &P, bool queued)
{
    if (!queued) {                                      // the default: k_march alone, its waves march their own shadow rays
        if (mode == OUT_RGBA32F)     hipLaunchKernelGGL((k_march<CUR, COUNT, OUT_RGBA32F, false>), grid, dim3(64), 0, st, P);
        else if (mode == OUT_GAMMA8) hipLaunchKernelGGL((k_march<CUR, COUNT, OUT_GAMMA8, false>), grid, dim3(64), 0, st, P);
        else if (mode == OUT_HEAT8)  hipLaunchKernelGGL((k_march<CUR, COUNT, OUT_HEAT8, false>), grid, dim3(64), 0, st, P);
        else if (mode == OUT_SPARSE) hipLaunchKernelGGL((k_march<CUR, COUNT, OUT_SPARSE, false>), grid, dim3(64), 0, st, P);
        else                         hipLaunchKernelGGL((k_march<CUR, COUNT, OUT_WIRE, false>), grid, dim3(64), 0, st, P);
        return;
    }
    auto go = [&](auto march, auto shade) {
        hipLaunchKernelGGL(march, grid, dim3(64), 0, st, P);
        hipLaunchKernelGGL(shade, shade_grid, dim3(64), 0, st, P);
    };
    if (mode == OUT_RGBA32F)     go(k_march<CUR, COUNT, OUT_RGBA32F, true>, k_shadow<CUR, COUNT, OUT_RGBA32F>);
    else if (mode == OUT_GAMMA8) go(k_march<CUR, COUNT, OUT_GAMMA8, true>, k_shadow<CUR, COUNT, OUT_GAMMA8>);
    else if (mode == OUT_HEAT8)  go(k_march<CUR, COUNT, OUT_HEAT8, true>, k_shadow<CUR, COUNT, OUT_HEAT8>);
    else                         go(k_march<CUR, COUNT, OUT_WIRE, true>, k_shadow<CUR, COUNT, OUT_WIRE>);
}

// the default kernel through the grid's second form, 4-byte words + sample records (CursorFF): not counting, shadow rays marched in the wave
void launch_fast(uint32_t mode, dim3 grid, hipStream_t st, const RenderParams &P)
{
    if (mode == OUT_RGBA32F)     hipLaunchKernelGGL((k_march<CUR_DENSE4, false, OUT_RGBA32F, false>), grid, dim3(64), 0, st, P);
    else if (mode == OUT_GAMMA8) hipLaunchKernelGGL((k_march<CUR_DENSE4, false, OUT_GAMMA8, false>), grid, dim3(64), 0, st, P);
    else if (mode == OUT_HEAT8)  hipLaunchKernelGGL((k_march<CUR_DENSE4, false, OUT_HEAT8, false>), grid, dim3(64), 0, st, P);
    else if (mode == OUT_SPARSE) hipLaunchKernelGGL((k_march<CUR_DENSE4, false, OUT_SPARSE, false>), grid, dim3(64), 0, st, P);
    else                         hipLaunchKernelGGL((k_march<CUR_DENSE4, false, OUT_WIRE, false>), grid, dim3(64), 0, st, P);
}

template <int CUR, bool COUNT>
void launch_pair(bool compact, int bt, dim3 grid, hipStream_t st, const RenderParams &P)
{
    if (compact)        hipLaunchKernelGGL((k_compact<CUR, COUNT>), grid, dim3(64), 0, st, P);
    else if (bt == 64)  hipLaunchKernelGGL((k_plain<CUR, COUNT, 64>), grid, dim3(64), 0, st, P);
    else if (bt == 128) hipLaunchKernelGGL((k_plain<CUR, COUNT, 128>), grid, dim3(128), 0, st, P);
    else                hipLaunchKernelGGL((k_plain<CUR, COUNT, 256>), grid, dim3(256), 0, st, P);
}

int render_impl(sdfhip_scene *s, const sdfhip_info *info, uint32_t width, uint32_t height,
                uint32_t band_rows, uint32_t band_first, uint32_t band_stride, uint32_t nrows_out,
                uint32_t flags, float *d_out, hipStream_t st, sdfhip_stats *stats,
                const sdfhip_pathtrace *pt = nullptr, uint32_t n_frames = 1,
                const uint16_t *bands = nullptr, uint32_t n_bands = 0, uint32_t sparse_cap = 0, bool sparse = false, uint32_t sparse_base = 0)
{
    // `info` points at n_frames consecutive Info blocks (batched launch: plain kernel only)
    if (n_frames == 0 || n_frames > (uint32_t)MAX_BATCH)
        return fail(SDFHIP_ERR_ARG, "render: n_frames %u outside 1..%d", n_frames, MAX_BATCH);
    if (width == 0 || height == 0 || nrows_out == 0 || band_rows == 0 || band_stride == 0)
        return fail(SDFHIP_ERR_ARG, "render: zero-sized frame or band");
    if ((uint64_t)width * nrows_out > 0x7FFFFFFFull)
        return fail(SDFHIP_ERR_ARG, "render: %u x %u pixels exceed the 31-bit pixel index", width, nrows_out);
    uint32_t kind = flags & SDFHIP_KERNEL_MASK;
    if (kind > SDFHIP_KERNEL_STACK) return fail(SDFHIP_ERR_ARG, "render: unknown kernel selector %u", kind);
    if (kind == SDFHIP_KERNEL_STACK && !s->stack_ok)
        return fail(SDFHIP_ERR_ARG, "render: the cursor-stack kernel needs a parent/child-consistent tree of depth <= %d (this scene: depth %u)", MAX_STACK, s->depth);
    const bool use_stack = kind == SDFHIP_KERNEL_STACK || (kind == SDFHIP_KERNEL_AUTO && s->stack_ok);
    const bool compact = (flags & SDFHIP_FLAG_COMPACT) != 0;
    const bool count = (flags & SDFHIP_FLAG_COUNT) != 0;
    const bool wire = (flags & SDFHIP_FLAG_WIRE) != 0;
    const uint32_t out_mode = sparse ? 4u : wire ? 3u : (flags & SDFHIP_FLAG_DISPLAY_DEBUG) ? 2u : ((flags & SDFHIP_FLAG_DISPLAY) ? 1u : 0u);
    if (sparse && (wire || compact || pt || (flags & (SDFHIP_FLAG_DISPLAY | SDFHIP_FLAG_DISPLAY_DEBUG | SDFHIP_TUNE_SHADOW_QUEUE | SDFHIP_TUNE_ONE_KERNEL))))
        return fail(SDFHIP_ERR_ARG, "render_sparse: sparse shares come from the default kernel only (no display pass, path tracing, compaction or A/B forms)");
    if (wire && (compact || pt || (flags & (SDFHIP_FLAG_DISPLAY | SDFHIP_FLAG_DISPLAY_DEBUG))))
        return fail(SDFHIP_ERR_ARG, "render: wire pixels come from the plain kernel only, without the display pass");
    if (wire && ((uint64_t)nrows_out * width) % 4 != 0)
        return fail(SDFHIP_ERR_ARG, "render: wire buffers need nrows_out * width to be a multiple of 4 (the byte plane follows the float plane)");
    if (pt) {
        if (pt->spp == 0 || pt->spp > 4096 || pt->max_bounces > 64)
            return fail(SDFHIP_ERR_ARG, "render_path: spp %u (1..4096) or max_bounces %u (0..64) out of range", pt->spp, pt->max_bounces);
        if (out_mode != 0 || compact)
            return fail(SDFHIP_ERR_ARG, "render_path: the display pass and compaction are not available in path-traced mode");
    }

    RenderParams P;
    P.nodes = s->nodes; P.n_nodes = s->n; P.top = s->d_top; P.top_level = s->top_level; P.fine = s->d_fine; P.fine_bits = s->fine_bits; P.fine_order = 0; P.d4 = s->d_d4;
    // (a word of d4 is TAG + the record's index in 16-byte units: the kernel adds the whole word to this pointer)
    P.recs = reinterpret_cast<const uint4 *>(reinterpret_cast<uintptr_t>(s->d_recs) - ((uintptr_t)D4_TAG << 4));
    P.out = reinterpret_cast<float4 *>(d_out);
    P.width = width; P.height = height;
    P.band_rows = band_rows; P.band_first = band_first; P.band_stride = band_stride;
    P.nrows_out = nrows_out;
    P.n_band_list = 0;
    memset(P.band_list, 0, sizeof P.band_list);
    if (bands) {                                      // an explicit band list replaces first/stride
        if (n_bands == 0 || n_bands > (uint32_t)MAX_BAND_LIST)
            return fail(SDFHIP_ERR_ARG, "render_bands: %u bands outside 1..%d", n_bands, MAX_BAND_LIST);
        if ((uint64_t)n_bands * band_rows > nrows_out)
            return fail(SDFHIP_ERR_ARG, "render_bands: %u bands of %u rows do not fit nrows_out = %u", n_bands, band_rows, nrows_out);
        const uint32_t frame_bands = (height + band_rows - 1) / band_rows;
        for (uint32_t i = 0; i < n_bands; i++) {
            if (bands[i] >= frame_bands)
                return fail(SDFHIP_ERR_ARG, "render_bands: band %u of a frame with %u bands", (unsigned)bands[i], frame_bands);
            P.band_list[i] = bands[i];
        }
        P.n_band_list = n_bands;
    }
    P.tile_order = (flags >> 8) & 0xF;
    const uint32_t btsel = (flags >> 12) & 0xF;                       // tuning knob: 0 = default
    const int bt = btsel == 3 ? 256 : (btsel == 2 ? 128 : 64);
    const uint32_t tile_w = (compact || pt) ? 8u : (bt >= 128 ? 16u : 8u);
    const uint32_t tile_h = (compact || pt) ? 8u : (uint32_t)bt / 8u / (tile_w / 8u);
    P.tiles_x = (width + tile_w - 1) / tile_w;
    P.tiles_y = (nrows_out + tile_h - 1) / tile_h;
    P.n_tiles = P.tiles_x * P.tiles_y;
    auto unpack = [](const sdfhip_info *in, FrameInfo &I) {
        I.h0x = in->heading[0][0]; I.h0y = in->heading[0][1]; I.h0z = in->heading[0][2];
        I.h1x = in->heading[1][0]; I.h1y = in->heading[1][1]; I.h1z = in->heading[1][2];
        I.h2x = in->heading[2][0]; I.h2y = in->heading[2][1]; I.h2z = in->heading[2][2];
        I.posx = in->position[0]; I.posy = in->position[1]; I.posz = in->position[2];
        I.margin = in->margin;
        I.margin2 = in->margin * 2.0f;
        I.screen_w = in->screen_size[0]; I.screen_h = in->screen_size[1];
        I.limit = in->limit;
        I.lightx = in->light[0]; I.lighty = in->light[1]; I.lightz = in->light[2];
        I.fov = in->fov;
        I.k_strength = exp2f(in->strength) - 1.0f;   // Compute.hlsl:216, once per frame
    };
    P.n_frames = n_frames;
    memset(P.frames, 0, sizeof P.frames);
    for (uint32_t f = 0; f < n_frames; f++) unpack(info + f, P.frames[f]);
    if (n_frames > 1 && (compact || pt || count))
        return fail(SDFHIP_ERR_ARG, "render_batch: only the default kernels, without counting, render several frames per launch");
    if (compact && (width > 65535u || nrows_out > 65535u))
        return fail(SDFHIP_ERR_ARG, "render: the compact kernel packs a pixel's x and row into 16 bits each (frame %u x %u)", width, nrows_out);
    P.out_mode = out_mode;
    P.sparse_cap = sparse_cap;
    P.sparse_base = sparse_base;
    {   // the sky constant of Compute.hlsl:196 through DisplayFrag.hlsl:24, alpha excluded
        auto q = [](float c) { float v = powf(c, 1.0f / 2.2f); v = v > 1.0f ? 1.0f : (v > 0.0f ? v : 0.0f); return (uint32_t)(v * 255.0f + 0.5f); };
        P.sky8 = q(0.005f) | (q(0.01f) << 8) | (q(0.2f) << 16);
    }
    P.pt_spp = pt ? pt->spp : 0; P.pt_bounces = pt ? pt->max_bounces : 0; P.pt_seed = pt ? pt->seed : 0;
    P.pt_albedo = pt ? pt->albedo : 0.0f;
    P.counters = nullptr;
    // cursor kind: generic, cursor stack, or cursor stack with a top grid as deep as the tree
    const int cur = !use_stack ? CUR_GENERIC : (s->d_top && s->fine_bits) ? CUR_STACK_SPLIT :
                    (s->d_top && (uint32_t)s->top_level >= s->depth) ? CUR_STACK_FULL : CUR_STACK;
    // the two-kernel pipeline (k_march -> k_shadow) is the default wherever a find is a grid lookup
    const bool two = (cur == CUR_STACK_FULL || cur == CUR_STACK_SPLIT) && !compact && !pt && bt == 64 &&
                     P.tile_order == 0 && !(flags & SDFHIP_TUNE_ONE_KERNEL);

    dim3 grid, shade_grid;
    if (compact) {
        uint32_t blocks = (uint32_t)s->cu_count * 32u;   // one wave per workgroup, 32 waves per CU
        grid = dim3(blocks < P.n_tiles ? blocks : (P.n_tiles ? P.n_tiles : 1));
    } else {
        grid = dim3(P.tile_order == 0 ? 8u * ((P.tiles_y + 7u) / 8u) * P.tiles_x : P.n_tiles, n_frames);
    }
    sdfhip_scene::Scratch *sc = nullptr;
    P.queue = nullptr; P.hit_a = nullptr; P.hit_b = nullptr; P.hit_c = nullptr; P.hit_d = nullptr; P.hit_ctl = nullptr; P.hit_cap = 0; P.hit_set = 0;
    P.tile_perm = n_frames == 1 ? s->dbg_tile_perm : nullptr;      // (the experiment hook is for single frames: its arrays hold one frame's tiles)
    P.tile_cost = n_frames == 1 ? s->dbg_tile_cost : nullptr;
    P.pt_q[0] = P.pt_q[1] = nullptr; P.pt_ctl = nullptr; P.pt_cap = 0; P.pt_level = 0; P.pt_e = nullptr; P.pt_t = nullptr; P.pt_n = nullptr;
    const bool queued = two && (flags & SDFHIP_TUNE_SHADOW_QUEUE) != 0;
    // SDFHIP_FLAG_TILE_ORDER: this frame's tiles in the order made from the last frame of the same geometry on this stream
    // (frames of more than 65 536 tiles -- 4K -- run 16 rounds of workgroups: their tail is short and the order costs locality)
    const bool ordered = two && (flags & SDFHIP_FLAG_TILE_ORDER) != 0 && n_frames == 1 && P.n_tiles <= 65536u &&
                         grid.x <= 8u * 1024u * (uint32_t)ORDER_SPAN && !s->dbg_tile_perm && !s->dbg_tile_cost;
    if (ordered) {
        int rcs = get_scratch(s, st, 0, &sc);
        if (rcs != SDFHIP_OK) return rcs;
        uint32_t sig[8] = { width, height, nrows_out, band_rows, band_first, band_stride, P.n_band_list, 0u };
        for (uint32_t i = 0; i < P.n_band_list; i++) sig[7] = sig[7] * 31u + P.band_list[i] + 1u;
        if (P.n_tiles > sc->ord_tiles || grid.x > sc->ord_blocks) {
            HIP_TRY(hipStreamSynchronize(st));
            if (sc->ord_cost) (void)hipFree(sc->ord_cost);
            if (sc->ord_class) (void)hipFree(sc->ord_class);
            if (sc->ord_perm) (void)hipFree(sc->ord_perm);
            sc->ord_cost = nullptr; sc->ord_class = nullptr; sc->ord_perm = nullptr; sc->ord_tiles = sc->ord_blocks = 0; sc->ord_valid = false;
            HIP_TRY(hipMalloc((void **)&sc->ord_cost, (size_t)P.n_tiles * sizeof(uint16_t)));
            HIP_TRY(hipMalloc((void **)&sc->ord_class, (size_t)P.n_tiles));
            HIP_TRY(hipMalloc((void **)&sc->ord_perm, (size_t)grid.x * sizeof(uint32_t)));
            sc->ord_tiles = P.n_tiles; sc->ord_blocks = grid.x;
        }
        if (memcmp(sig, sc->ord_sig, sizeof sig) != 0) { sc->ord_valid = false; memcpy(sc->ord_sig, sig, sizeof sig); }
        P.tile_perm = sc->ord_valid ? sc->ord_perm : nullptr;
        P.tile_cost = sc->ord_cost;
    }
    if (queued) {
        // a queue takes the hits of every 64th workgroup: room for all their pixels
        P.hit_cap = ((grid.x + HIT_QUEUES - 1u) / HIT_QUEUES) * 64u;
        const size_t records = (size_t)n_frames * HIT_QUEUES * P.hit_cap;
        int rcs = get_scratch(s, st, records, &sc);
        if (rcs != SDFHIP_OK) return rcs;
        P.hit_a = reinterpret_cast<float4 *>(sc->hit_buf);
        P.hit_b = reinterpret_cast<int4 *>(sc->hit_buf + sc->records * 16);
        P.hit_c = reinterpret_cast<uint4 *>(sc->hit_buf + sc->records * 32);
        P.hit_d = reinterpret_cast<float4 *>(sc->hit_buf + sc->records * 48);
        P.hit_ctl = sc->ctl;
        P.hit_set = sc->launches++ & 1u;
        // every queued hit is shaded by a resident wave: at most one chunk of 64 per k_march workgroup
        const uint32_t resident = (uint32_t)s->cu_count * 32u;
        shade_grid = dim3(grid.x < resident ? grid.x : resident, n_frames);
    } else if (compact) {
        int rcs = get_scratch(s, st, 0, &sc);
        if (rcs != SDFHIP_OK) return rcs;
        P.queue = sc->ctl + sdfhip_scene::CTL_HIT_WORDS;
        HIP_TRY(hipMemsetAsync(P.queue, 0, sdfhip_scene::CTL_QUEUE_WORDS * sizeof(uint32_t), st));
    }
    if (count) {                                          // this stream's own counters (its scratch)
        if (!sc) { int rcs = get_scratch(s, st, 0, &sc); if (rcs != SDFHIP_OK) return rcs; }
        P.counters = reinterpret_cast<unsigned long long *>(sc->ctl + sdfhip_scene::CTL_HIT_WORDS + sdfhip_scene::CTL_QUEUE_WORDS + sdfhip_scene::CTL_PT_WORDS);
        HIP_TRY(hipMemsetAsync(P.counters, 0, sdfhip_scene::CTL_COUNTER_WORDS * sizeof(uint32_t), st));
    }
    if (sparse) {
        if (!two) return fail(SDFHIP_ERR_ARG, "render_sparse: this scene has no full-depth grid (trees deeper than 12 levels or with inconsistent links render dense shares)");
    }
    if (pt && (cur == CUR_STACK_FULL || cur == CUR_STACK_SPLIT) && !(flags & SDFHIP_TUNE_ONE_KERNEL)) ensure_scatter_grid(s);   // (before the clock)
    if (stats) HIP_TRY(hipEventRecord(s->ev0, st));
    if (two) {
        if (!count && !queued && s->d_d4 && !(flags & SDFHIP_TUNE_BYTE_CELLS)) launch_fast(out_mode, grid, st, P);
        else if (cur == CUR_STACK_SPLIT) { if (count) launch_two<CUR_STACK_SPLIT, true>(out_mode, grid, shade_grid, st, P, queued); else launch_two<CUR_STACK_SPLIT, false>(out_mode, grid, shade_grid, st, P, queued); }
        else                        { if (count) launch_two<CUR_STACK_FULL, true>(out_mode, grid, shade_grid, st, P, queued); else launch_two<CUR_STACK_FULL, false>(out_mode, grid, shade_grid, st, P, queued); }
        // the next frame's launch order, behind this frame in its stream -- unless the order in use was made from a frame with this
        // very camera block: the same camera gives the same costs and the same order (a viewer at rest pays for the order once)
        if (ordered && !(sc->ord_valid && memcmp(info, &sc->ord_info, sizeof(sdfhip_info)) == 0)) {
            sc->ord_info = *info;
            hipLaunchKernelGGL(k_tile_class, dim3((P.n_tiles + 255u) / 256u), dim3(256), 0, st, sc->ord_cost, sc->ord_class, P.tiles_x, P.tiles_y);
            hipLaunchKernelGGL(k_tile_order, dim3(8), dim3(1024), 0, st, sc->ord_class, sc->ord_perm, P.tiles_x, P.tiles_y);
            sc->ord_valid = true;
        }
    }
    else if (pt && (cur == CUR_STACK_FULL || cur == CUR_STACK_SPLIT) && !(flags & SDFHIP_TUNE_ONE_KERNEL)) {
        // the pipeline of kernels (k_pt_primary -> k_pt_bounce per level -> k_pt_resolve)
        grid = dim3(8u * ((P.tiles_y + 7u) / 8u) * P.tiles_x);
        const size_t npx = (size_t)nrows_out * width, npaths = npx * pt->spp;
        if (npaths >= ((size_t)1 << 32))
            return fail(SDFHIP_ERR_ARG, "render_path: %zu paths (pixels x spp) exceed the 32-bit path index", npaths);
        P.pt_cap = (uint32_t)((((size_t)grid.x + HIT_QUEUES - 1) / HIT_QUEUES) * 64 * pt->spp + 8192);
        const size_t qbytes = (size_t)PT_RECORDS * 16 * HIT_QUEUES * P.pt_cap;        // one hit queue
        const size_t ebytes = (size_t)(pt->max_bounces + 1) * npaths * 4, tbytes = npaths * 4;
        int rcs = get_pt_scratch(s, st, 2 * qbytes + ebytes + 2 * tbytes, &sc);
        if (rcs != SDFHIP_OK) return rcs;
        P.pt_q[0] = reinterpret_cast<float4 *>(sc->pt_buf);
        P.pt_q[1] = reinterpret_cast<float4 *>(sc->pt_buf + qbytes);
        P.pt_e = reinterpret_cast<float *>(sc->pt_buf + 2 * qbytes);
        P.pt_t = reinterpret_cast<float *>(sc->pt_buf + 2 * qbytes + ebytes);
        P.pt_n = reinterpret_cast<uint32_t *>(sc->pt_buf + 2 * qbytes + ebytes + tbytes);
        P.pt_ctl = sc->ctl + sdfhip_scene::CTL_HIT_WORDS + sdfhip_scene::CTL_QUEUE_WORDS;
        int rcl;
        if (cur == CUR_STACK_SPLIT) rcl = count ? launch_pt<CUR_STACK_SPLIT, true>(s, sc, grid, st, P) : launch_pt<CUR_STACK_SPLIT, false>(s, sc, grid, st, P);
        else                        rcl = count ? launch_pt<CUR_STACK_FULL, true>(s, sc, grid, st, P) : launch_pt<CUR_STACK_FULL, false>(s, sc, grid, st, P);
        if (rcl != SDFHIP_OK) return rcl;
    }
    else if (pt) {
        grid = dim3(8u * ((P.tiles_y + 7u) / 8u) * P.tiles_x);
        auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, dim3(64), 0, st, P); };
        if (cur == CUR_STACK_SPLIT) { if (count) go(k_path<CUR_STACK_SPLIT, true>); else go(k_path<CUR_STACK_SPLIT, false>); }
        else if (cur == CUR_STACK_FULL) { if (count) go(k_path<CUR_STACK_FULL, true>); else go(k_path<CUR_STACK_FULL, false>); }
        else if (cur == CUR_STACK) { if (count) go(k_path<CUR_STACK, true>); else go(k_path<CUR_STACK, false>); }
        else                       { if (count) go(k_path<CUR_GENERIC, true>); else go(k_path<CUR_GENERIC, false>); }
    }
    else if ((flags & SDFHIP_TUNE_LDS_TOP) && cur == CUR_STACK && s->d_top && s->top_level <= 3 && !compact && !count) {
        // measurement variant: the top grid staged in LDS per workgroup (64- or 256-thread workgroups)
        if (bt == 256) hipLaunchKernelGGL((k_plain<CUR_STACK, false, 256, true>), grid, dim3(256), 0, st, P);
        else           hipLaunchKernelGGL((k_plain<CUR_STACK, false, 64, true>), grid, dim3(64), 0, st, P);
    }
    else if (cur == CUR_STACK_SPLIT) { if (count) launch_pair<CUR_STACK_SPLIT, true>(compact, bt, grid, st, P); else launch_pair<CUR_STACK_SPLIT, false>(compact, bt, grid, st, P); }
    else if (cur == CUR_STACK_FULL) { if (count) launch_pair<CUR_STACK_FULL, true>(compact, bt, grid, st, P); else launch_pair<CUR_STACK_FULL, false>(compact, bt, grid, st, P); }
    else if (cur == CUR_STACK)      { if (count) launch_pair<CUR_STACK, true>(compact, bt, grid, st, P); else launch_pair<CUR_STACK, false>(compact, bt, grid, st, P); }
    else                            { if (count) launch_pair<CUR_GENERIC, true>(compact, bt, grid, st, P); else launch_pair<CUR_GENERIC, false>(compact, bt, grid, st, P); }
    HIP_TRY(hipGetLastError());
    if (sc) HIP_TRY(hipEventRecord(sc->idle, st));        // this stream's scratch is busy until here
    if (stats) {
        HIP_TRY(hipEventRecord(s->ev1, st));
        HIP_TRY(hipEventSynchronize(s->ev1));
        memset(stats, 0, sizeof *stats);
        HIP_TRY(hipEventElapsedTime(&stats->kernel_ms, s->ev0, s->ev1));
        stats->kernel_used = (use_stack ? SDFHIP_KERNEL_STACK : SDFHIP_KERNEL_GENERIC) |
                             (compact ? SDFHIP_FLAG_COMPACT : 0u);
        if (count) {
            unsigned long long h[6];
            HIP_TRY(hipMemcpyAsync(h, P.counters, sizeof h, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            stats->n_nodes = h[0]; stats->n_samples = h[1]; stats->n_steps = h[2]; stats->n_shadow_rays = h[3];
            stats->n_loads = h[4]; stats->n_hits = h[5];
        }
    }
    return SDFHIP_OK;
}

}  // namespace

extern "C" int sdfhip_render_device(sdfhip_scene *s, const sdfhip_info *info, uint32_t width,
                                    uint32_t height, uint32_t band_rows, uint32_t band_first,
                                    uint32_t band_stride, uint32_t nrows_out, uint32_t flags,
                                    float *d_rgba_out, void *stream, sdfhip_stats *stats)
{
    if (!s || !info || !d_rgba_out) return fail(SDFHIP_ERR_ARG, "render_device: null argument");
    std::lock_guard<std::mutex> lk(s->lock);
    DeviceGuard g(s->device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "render_device: hipSetDevice(%d) failed", s->device);
    hipStream_t st = (hipStream_t)stream;   // NULL = the HIP default stream, as everywhere in HIP
    return render_impl(s, info, width, height, band_rows, band_first, band_stride, nrows_out, flags,
                       d_rgba_out, st, stats);
}

extern "C" int sdfhip_render_batch_device(sdfhip_scene *s, const sdfhip_info *infos, uint32_t n_frames,
                                          uint32_t width, uint32_t height, uint32_t band_rows,
                                          uint32_t band_first, uint32_t band_stride, uint32_t nrows_out,
                                          uint32_t flags, float *d_rgba_out, void *stream, sdfhip_stats *stats)
{
    if (!s || !infos || !d_rgba_out) return fail(SDFHIP_ERR_ARG, "render_batch_device: null argument");
    std::lock_guard<std::mutex> lk(s->lock);
    DeviceGuard g(s->device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "render_batch_device: hipSetDevice(%d) failed", s->device);
    return render_impl(s, infos, width, height, band_rows, band_first, band_stride, nrows_out, flags,
                       d_rgba_out, (hipStream_t)stream, stats, nullptr, n_frames);
}

// ---- host frames the copy engine can write by itself ---------------------------------------------------------------------
// A copy into pageable memory is staged by the runtime inside the copy call: the host sits in it, and the copies of a frame's
// bands go one after another behind the host.  Into page-locked memory the call returns at once and the band's copy starts when
// its march ends.  A host that keeps ONE frame array for its lifetime (the viewer: Program.cs:94-99 reads every frame back into
// the same texture-sized array) either lets the library allocate it (sdfhip_host_alloc) or page-locks its own once
// (sdfhip_host_register: C#, a GCHandleType.Pinned handle held as long as the registration); sdfhip_render looks the
// destination up here.
namespace {
struct HostRange { uintptr_t p; size_t bytes; uintptr_t dev; bool ours; };
std::mutex g_host_lock;
std::vector<HostRange> g_host_ranges;
// the address the devices know [p, p + bytes) by, or null when the range is not page-locked here; ours: it is the library's own
void *host_range_device_pointer(const void *p, size_t bytes, bool *ours)
{
    std::lock_guard<std::mutex> lk(g_host_lock);
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    for (const HostRange &r : g_host_ranges)
        if (a >= r.p && a + bytes <= r.p + r.bytes) { *ours = r.ours; return reinterpret_cast<void *>(r.dev + (a - r.p)); }
    return nullptr;
}
int host_range_add(void *p, size_t bytes, bool ours, const char *what)
{
    void *dev = nullptr;
    hipError_t e = hipHostGetDevicePointer(&dev, p, 0);
    if (e != hipSuccess || !dev) {
        (void)hipGetLastError();
        if (ours) (void)hipHostFree(p); else (void)hipHostUnregister(p);
        return fail(SDFHIP_ERR_DEVICE, "%s: hipHostGetDevicePointer failed: %s", what, hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> lk(g_host_lock);
    g_host_ranges.push_back(HostRange{reinterpret_cast<uintptr_t>(p), bytes, reinterpret_cast<uintptr_t>(dev), ours});
    return SDFHIP_OK;
}
}
extern "C" int sdfhip_host_alloc(uint64_t bytes, void **out)
{
    if (!out || bytes == 0) return fail(SDFHIP_ERR_ARG, "host_alloc: null or zero argument");
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, (size_t)bytes, hipHostMallocPortable | hipHostMallocMapped);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(SDFHIP_ERR_NOMEM, "host_alloc: hipHostMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e)); }
    int rc = host_range_add(p, (size_t)bytes, true, "host_alloc");
    if (rc != SDFHIP_OK) return rc;
    *out = p;
    return SDFHIP_OK;
}
extern "C" int sdfhip_host_register(void *p, uint64_t bytes)
{
    if (!p || bytes == 0) return fail(SDFHIP_ERR_ARG, "host_register: null or zero argument");
    hipError_t e = hipHostRegister(p, (size_t)bytes, hipHostRegisterPortable | hipHostRegisterMapped);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(SDFHIP_ERR_DEVICE, "host_register: hipHostRegister(%llu bytes) failed: %s", (unsigned long long)bytes, hipGetErrorString(e)); }
    return host_range_add(p, (size_t)bytes, false, "host_register");
}
extern "C" int sdfhip_host_release(void *p)
{
    if (!p) return SDFHIP_OK;
    HostRange r{0, 0, 0, false};
    {
        std::lock_guard<std::mutex> lk(g_host_lock);
        for (size_t i = 0; i < g_host_ranges.size(); i++)
            if (g_host_ranges[i].p == reinterpret_cast<uintptr_t>(p)) { r = g_host_ranges[i]; g_host_ranges.erase(g_host_ranges.begin() + (long)i); break; }
    }
    if (!r.p) return fail(SDFHIP_ERR_ARG, "host_release: %p was neither allocated nor registered here", p);
    hipError_t e = r.ours ? hipHostFree(p) : hipHostUnregister(p);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(SDFHIP_ERR_DEVICE, "host_release: %s", hipGetErrorString(e)); }
    return SDFHIP_OK;
}

extern "C" int sdfhip_render(sdfhip_scene *s, const sdfhip_info *info, uint32_t width,
                             uint32_t height, uint32_t flags, float *rgba_out, sdfhip_stats *stats)
{
    if (!s || !info || !rgba_out) return fail(SDFHIP_ERR_ARG, "render: null argument");
    if (width == 0 || height == 0) return fail(SDFHIP_ERR_ARG, "render: zero-sized frame");
    if (flags & SDFHIP_FLAG_WIRE) return fail(SDFHIP_ERR_ARG, "render: SDFHIP_FLAG_WIRE is for the device-resident entry points");
    std::lock_guard<std::mutex> lk(s->lock);
    DeviceGuard g(s->device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "render: hipSetDevice(%d) failed", s->device);
    auto t0 = std::chrono::steady_clock::now();
    const size_t px_bytes = (flags & (SDFHIP_FLAG_DISPLAY | SDFHIP_FLAG_DISPLAY_DEBUG)) ? 4 : sizeof(float4);
    // The viewer's call: one frame in flight by construction, and the copy to the host (33 MB of RGBA32F at 1080p: 0.6 ms at
    // PCIe's 55 GB/s) is most of it.  So the frame goes in row bands -- band b on its own stream behind band b - 1, its copy
    // behind it on that stream: the copy engine starts after the first band's march (a quarter of the frame) and runs beside the
    // others -- and every band launches its tiles in the order of their cost in the last frame (SDFHIP_FLAG_TILE_ORDER, kept per
    // stream; a band alone ends when its longest wave does).  Not with statistics or counters asked (one launch, one clock),
    // not for the A/B kernel forms, not for small frames.
    // Bands: 4 for frames of 4 M pixels and more, else one (each band costs a launch, a copy call and an event on the host, and a
    // 1080p march is a fifth of its copy: measured 0.735 / 0.744 / 0.757 ms with 1 / 2 / 4 bands against 0.770 in the plain form;
    // 4K: 2.77 / 2.62 / 2.51 against 2.75; 4K RGBA8: 0.96 / 0.84 / 0.76 against 0.96 -- scripts/host_frame.py);
    // SDFHIP_HOST_BANDS=n sets the number, 0 = the plain form (no tile order either)
    const char *hb = getenv("SDFHIP_HOST_BANDS");
    const bool viewer = !stats && !(flags & (SDFHIP_FLAG_COUNT | SDFHIP_FLAG_COMPACT | SDFHIP_TUNE_ONE_KERNEL | SDFHIP_TUNE_SHADOW_QUEUE | SDFHIP_TUNE_LDS_TOP)) &&
                        ((flags >> 8) & 0xFFu) == 0 && !(hb && atoi(hb) == 0);
    uint32_t nb = 1;
    if (viewer) {
        // A page-locked destination (sdfhip_host_alloc / _register): the march can store its pixels into the host's array
        // itself -- no device frame, no copy: the stores cross PCIe while the other waves march.  Measured (scripts/host_frame.py
        // --locked, profiles/r03_host_frame.txt): into the library's own allocation 1080p 0.676 ms against 0.711-0.731 with band
        // copies into the same memory (RGBA8: 0.260 against 0.292); at 4K the copy engine's 55 GB/s beat the stores' 51 (2.47
        // against 2.58 ms), so frames of 4 M pixels and more go in bands.  Into the caller's own registered array (4 KB pages
        // wherever they happened to lie) the stores are slower -- 1080p RGBA32F 0.758 against 0.714 -- and are used for frames of
        // less than 16 MB only (1080p RGBA8: 0.270 against 0.293).
        bool ours = false;
        const size_t frame_bytes = (size_t)width * height * px_bytes;
        void *const known = host_range_device_pointer(rgba_out, frame_bytes, &ours), *const direct = hb ? nullptr : known;
        const bool locked = known != nullptr;
        if (direct && (ours ? (size_t)width * height < ((size_t)4 << 20) : frame_bytes < ((size_t)16 << 20))) {
            int rc = render_impl(s, info, width, height, height, 0, 1, height, flags | SDFHIP_FLAG_TILE_ORDER,
                                 reinterpret_cast<float *>(direct), s->stream, nullptr);
            if (rc != SDFHIP_OK) return rc;
            HIP_TRY(hipStreamSynchronize(s->stream));
            return SDFHIP_OK;
        }
        nb = ((size_t)width * height >= ((size_t)4 << 20) || (locked && frame_bytes >= ((size_t)16 << 20))) ? 4u : 1u;
        if (hb && atoi(hb) > 0) nb = (uint32_t)atoi(hb);
        nb = nb < 1u ? 1u : (nb > (uint32_t)sdfhip_scene::HOST_BANDS ? (uint32_t)sdfhip_scene::HOST_BANDS : nb);
        if (height < 64u * nb) nb = 1;
    }
    const bool banded = viewer;
    const uint32_t rows = banded ? (((height + nb - 1) / nb + 7u) & ~7u) : height;           // whole 8x8 tiles per band
    size_t need = (size_t)width * rows * nb;
    if (need > s->frame_cap) {
        if (s->d_frame) { (void)hipFree(s->d_frame); s->d_frame = nullptr; s->frame_cap = 0; }
        HIP_TRY(hipMalloc((void **)&s->d_frame, need * sizeof(float4)));
        s->frame_cap = need;
    }
    if (!banded) {
        int rc = render_impl(s, info, width, height, height, 0, 1, height, flags,
                             reinterpret_cast<float *>(s->d_frame), s->stream, stats);
        if (rc != SDFHIP_OK) return rc;
        HIP_TRY(hipMemcpyAsync(rgba_out, s->d_frame, (size_t)width * height * px_bytes, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        if (stats)
            stats->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return SDFHIP_OK;
    }
    for (uint32_t b = 0; b < nb; b++) {
        if (!s->band_stream[b]) {
            HIP_TRY(hipStreamCreateWithFlags(&s->band_stream[b], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&s->band_done[b], hipEventDisableTiming));
        }
    }
    char *const d_base = reinterpret_cast<char *>(s->d_frame);
    for (uint32_t b = 0; b < nb && b * rows < height; b++) {                    // every band's march first ...
        if (b) HIP_TRY(hipStreamWaitEvent(s->band_stream[b], s->band_done[b - 1], 0));
        int rc = render_impl(s, info, width, height, rows, b, nb, rows, flags | SDFHIP_FLAG_TILE_ORDER,
                             reinterpret_cast<float *>(d_base + (size_t)b * rows * width * px_bytes), s->band_stream[b], nullptr);
        if (rc != SDFHIP_OK) return rc;
        HIP_TRY(hipEventRecord(s->band_done[b], s->band_stream[b]));
    }
    for (uint32_t b = 0; b < nb && b * rows < height; b++) {                    // ... then the copies, each behind its band
        const uint32_t r0 = b * rows, nr = (r0 + rows <= height) ? rows : height - r0;
        HIP_TRY(hipMemcpyAsync(reinterpret_cast<char *>(rgba_out) + (size_t)r0 * width * px_bytes, d_base + (size_t)r0 * width * px_bytes,
                               (size_t)nr * width * px_bytes, hipMemcpyDeviceToHost, s->band_stream[b]));
    }
    for (uint32_t b = 0; b < nb && b * rows < height; b++) HIP_TRY(hipStreamSynchronize(s->band_stream[b]));
    return SDFHIP_OK;
}

extern "C" int sdfhip_render_path_device(sdfhip_scene *s, const sdfhip_info *info, const sdfhip_pathtrace *pt,
                                         uint32_t width, uint32_t height, uint32_t band_rows,
                                         uint32_t band_first, uint32_t band_stride, uint32_t nrows_out,
                                         uint32_t flags, float *d_rgba_out, void *stream, sdfhip_stats *stats)
{
    if (!s || !info || !pt || !d_rgba_out) return fail(SDFHIP_ERR_ARG, "render_path_device: null argument");
    std::lock_guard<std::mutex> lk(s->lock);
    DeviceGuard g(s->device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "render_path_device: hipSetDevice(%d) failed", s->device);
    return render_impl(s, info, width, height, band_rows, band_first, band_stride, nrows_out, flags,
                       d_rgba_out, (hipStream_t)stream, stats, pt);
}

extern "C" int sdfhip_render_bands_device(sdfhip_scene *s, const sdfhip_info *infos, uint32_t n_frames,
                                          const sdfhip_pathtrace *pt, uint32_t width, uint32_t height,
                                          uint32_t band_rows, const uint16_t *bands, uint32_t n_bands,
                                          uint32_t nrows_out, uint32_t flags, float *d_rgba_out, void *stream,
                                          sdfhip_stats *stats)
{
    if (!s || !infos || !bands || !d_rgba_out) return fail(SDFHIP_ERR_ARG, "render_bands_device: null argument");
    if (pt && n_frames != 1) return fail(SDFHIP_ERR_ARG, "render_bands_device: the path-traced mode renders one frame per launch");
    std::lock_guard<std::mutex> lk(s->lock);
    DeviceGuard g(s->device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "render_bands_device: hipSetDevice(%d) failed", s->device);
    return render_impl(s, infos, width, height, band_rows, 0, 1, nrows_out, flags, d_rgba_out,
                       (hipStream_t)stream, stats, pt, n_frames, bands, n_bands);
}

extern "C" int sdfhip_render_path(sdfhip_scene *s, const sdfhip_info *info, const sdfhip_pathtrace *pt,
                                  uint32_t width, uint32_t height, uint32_t flags, float *rgba_out,
                                  sdfhip_stats *stats)
{
    if (!s || !info || !pt || !rgba_out) return fail(SDFHIP_ERR_ARG, "render_path: null argument");
    if (width == 0 || height == 0) return fail(SDFHIP_ERR_ARG, "render_path: zero-sized frame");
    std::lock_guard<std::mutex> lk(s->lock);
    DeviceGuard g(s->device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "render_path: hipSetDevice(%d) failed", s->device);
    auto t0 = std::chrono::steady_clock::now();
    size_t need = (size_t)width * height;
    if (need > s->frame_cap) {
        if (s->d_frame) { (void)hipFree(s->d_frame); s->d_frame = nullptr; s->frame_cap = 0; }
        HIP_TRY(hipMalloc((void **)&s->d_frame, need * sizeof(float4)));
        s->frame_cap = need;
    }
    int rc = render_impl(s, info, width, height, height, 0, 1, height, flags,
                         reinterpret_cast<float *>(s->d_frame), s->stream, stats, pt);
    if (rc != SDFHIP_OK) return rc;
    HIP_TRY(hipMemcpyAsync(rgba_out, s->d_frame, need * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
    uint32_t overflow = 0;                              // a hit that found no room in its queue (see pt_push): never silently
    for (int i = 0; i < s->n_scratch; i++)
        if (s->scratch[i].stream == s->stream && s->scratch[i].pt_buf)
            HIP_TRY(hipMemcpyAsync(&overflow, s->scratch[i].ctl + sdfhip_scene::CTL_HIT_WORDS + sdfhip_scene::CTL_QUEUE_WORDS + (size_t)2 * HIT_QUEUES * 32,
                                   sizeof overflow, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (overflow) return fail(SDFHIP_ERR_NOMEM, "render_path: a hit queue of the path-traced pipeline overflowed; the frame is incomplete");
    if (stats)
        stats->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return SDFHIP_OK;
}

extern "C" int sdfhip_render_display(sdfhip_scene *s, const sdfhip_info *info, uint32_t width,
                                     uint32_t height, uint32_t flags, int debug, uint8_t *rgba8_out,
                                     sdfhip_stats *stats)
{
    flags = (flags & ~(uint32_t)(SDFHIP_FLAG_DISPLAY | SDFHIP_FLAG_DISPLAY_DEBUG)) |
            (debug ? SDFHIP_FLAG_DISPLAY_DEBUG : SDFHIP_FLAG_DISPLAY);
    return sdfhip_render(s, info, width, height, flags, reinterpret_cast<float *>(rgba8_out), stats);
}

static int deinterleave_impl(int device, const void *d_gathered, void *d_frame, uint32_t width, uint32_t height,
                            uint32_t band_rows, uint32_t world, uint32_t rows_per_rank, const uint8_t *owner,
                            uint32_t pixel_bytes, uint32_t frames, void *stream, uint32_t only_rank = 0xFFFFFFFFu)
{
    if (only_rank != 0xFFFFFFFFu && only_rank >= world)
        return fail(SDFHIP_ERR_ARG, "deinterleave_share: rank %u of %u", only_rank, world);
    if (frames == 0) return fail(SDFHIP_ERR_ARG, "deinterleave: frames must be >= 1");
    if (!d_gathered || !d_frame || width == 0 || height == 0 || band_rows == 0 || world == 0)
        return fail(SDFHIP_ERR_ARG, "deinterleave: null or zero argument");
    if (pixel_bytes != 16 && pixel_bytes != 4 && pixel_bytes != 5)
        return fail(SDFHIP_ERR_ARG, "deinterleave: pixel_bytes must be 16 (RGBA32F), 5 (wire) or 4 (RGBA8), got %u", pixel_bytes);
    if (pixel_bytes == 5 && ((size_t)rows_per_rank * width) % 4 != 0)
        return fail(SDFHIP_ERR_ARG, "deinterleave: wire buffers need rows_per_rank * width to be a multiple of 4");
    const uint32_t nbands = (height + band_rows - 1) / band_rows;
    BandMap M;
    M.n = 0;
    memset(M.src, 0, sizeof M.src);
    uint32_t need_rows = ((nbands + world - 1) / world) * band_rows;
    if (owner) {                                      // local band = how many earlier bands the same rank owns
        if (nbands > (uint32_t)MAX_BAND_LIST || world > 64)
            return fail(SDFHIP_ERR_ARG, "deinterleave_bands: %u bands (max %d) over %u ranks (max 64)", nbands, MAX_BAND_LIST, world);
        uint32_t have[64] = { 0 };
        for (uint32_t b = 0; b < nbands; b++) {
            if (owner[b] >= world) return fail(SDFHIP_ERR_ARG, "deinterleave_bands: band %u belongs to rank %u of %u", b, (unsigned)owner[b], world);
            M.src[b] = (uint16_t)((uint32_t)owner[b] << 10 | have[owner[b]]++);
        }
        M.n = nbands;
        need_rows = 0;
        for (uint32_t r = 0; r < world; r++) need_rows = have[r] * band_rows > need_rows ? have[r] * band_rows : need_rows;
    }
    if (rows_per_rank < need_rows)
        return fail(SDFHIP_ERR_ARG, "deinterleave: rows_per_rank %u < %u needed for %u bands over %u ranks", rows_per_rank, need_rows, nbands, world);
    DeviceGuard g(device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "deinterleave: hipSetDevice(%d) failed", device);
    size_t total = (size_t)width * height * frames;
    uint32_t blocks = (uint32_t)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (pixel_bytes == 16)
        hipLaunchKernelGGL((k_deinterleave<float4, float4>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const float4 *)d_gathered, (float4 *)d_frame, width, height, band_rows,
                           world, rows_per_rank, frames, M, only_rank);
    else if (pixel_bytes == 5)
        hipLaunchKernelGGL((k_deinterleave<WirePlanes, float4>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const WirePlanes *)d_gathered, (float4 *)d_frame, width, height, band_rows,
                           world, rows_per_rank, frames, M, only_rank);
    else
        hipLaunchKernelGGL((k_deinterleave<uint32_t, uint32_t>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const uint32_t *)d_gathered, (uint32_t *)d_frame, width, height, band_rows,
                           world, rows_per_rank, frames, M, only_rank);
    HIP_TRY(hipGetLastError());
    return SDFHIP_OK;
}

extern "C" int sdfhip_deinterleave_device(int device, const void *d_gathered, void *d_frame,
                                          uint32_t width, uint32_t height, uint32_t band_rows,
                                          uint32_t world, uint32_t rows_per_rank, uint32_t pixel_bytes,
                                          uint32_t frames, void *stream)
{
    return deinterleave_impl(device, d_gathered, d_frame, width, height, band_rows, world, rows_per_rank, nullptr,
                             pixel_bytes, frames, stream);
}

extern "C" int sdfhip_deinterleave_bands_device(int device, const void *d_gathered, void *d_frame,
                                                uint32_t width, uint32_t height, uint32_t band_rows,
                                                uint32_t world, uint32_t rows_per_rank, const uint8_t *owner,
                                                uint32_t pixel_bytes, uint32_t frames, void *stream)
{
    if (!owner) return fail(SDFHIP_ERR_ARG, "deinterleave_bands: null owner table");
    return deinterleave_impl(device, d_gathered, d_frame, width, height, band_rows, world, rows_per_rank, owner,
                             pixel_bytes, frames, stream);
}

extern "C" int sdfhip_deinterleave_share_device(int device, const void *d_share, void *d_frame,
                                                uint32_t width, uint32_t height, uint32_t band_rows,
                                                uint32_t world, uint32_t rows_per_rank, const uint8_t *owner,
                                                uint32_t rank, uint32_t pixel_bytes, uint32_t frames, void *stream)
{
    return deinterleave_impl(device, d_share, d_frame, width, height, band_rows, world, rows_per_rank, owner,
                             pixel_bytes, frames, stream, rank);
}

extern "C" uint64_t sdfhip_wire_sparse_bytes(uint32_t width, uint32_t rows, uint32_t capacity)
{
    return (uint64_t)sparse_layout(width, rows, capacity).bytes;
}

extern "C" uint64_t sdfhip_wire_sparse_head_offset(uint32_t width, uint32_t rows, uint32_t capacity)
{
    return (uint64_t)sparse_layout(width, rows, capacity).off_head;
}

extern "C" int sdfhip_wire_compact_device(int device, const void *d_wire, void *d_sparse, uint32_t width, uint32_t rows,
                                          uint32_t frames, uint32_t capacity, void *stream)
{
    if (!d_wire || !d_sparse || width == 0 || rows == 0 || frames == 0)
        return fail(SDFHIP_ERR_ARG, "wire_compact: null or zero argument");
    if (((size_t)rows * width) % 4 != 0) return fail(SDFHIP_ERR_ARG, "wire_compact: rows * width must be a multiple of 4");
    DeviceGuard g(device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "wire_compact: hipSetDevice(%d) failed", device);
    const SparseLayout L = sparse_layout(width, rows, capacity);
    const dim3 grid((L.tiles + 3) / 4, frames);
    hipLaunchKernelGGL(k_sparse_masks, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t *)d_wire, (uint8_t *)d_sparse, L);
    hipLaunchKernelGGL(k_sparse_scan, dim3(frames), dim3(1024), 0, (hipStream_t)stream, (uint8_t *)d_sparse, L);
    hipLaunchKernelGGL(k_sparse_scatter, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t *)d_wire, (uint8_t *)d_sparse, L);
    HIP_TRY(hipGetLastError());
    return SDFHIP_OK;
}

extern "C" int sdfhip_deinterleave_sparse_device(int device, const void *d_gathered, void *d_frame, uint32_t width,
                                                 uint32_t height, uint32_t band_rows, uint32_t world,
                                                 uint32_t rows_per_rank, const uint8_t *owner, uint32_t capacity,
                                                 uint32_t frames, uint32_t *d_overflow, void *stream)
{
    if (frames == 0 || !d_gathered || !d_frame || width == 0 || height == 0 || band_rows == 0 || world == 0)
        return fail(SDFHIP_ERR_ARG, "deinterleave_sparse: null or zero argument");
    if (band_rows % 8 != 0 || rows_per_rank % 8 != 0)
        return fail(SDFHIP_ERR_ARG, "deinterleave_sparse: bands must be whole 8x8 tiles (band_rows %u, rows_per_rank %u)", band_rows, rows_per_rank);
    const uint32_t nbands = (height + band_rows - 1) / band_rows;
    BandMap M;
    M.n = 0;
    memset(M.src, 0, sizeof M.src);
    uint32_t need_rows = ((nbands + world - 1) / world) * band_rows;
    if (owner) {
        if (nbands > (uint32_t)MAX_BAND_LIST || world > 64)
            return fail(SDFHIP_ERR_ARG, "deinterleave_sparse: %u bands (max %d) over %u ranks (max 64)", nbands, MAX_BAND_LIST, world);
        uint32_t have[64] = { 0 };
        for (uint32_t b = 0; b < nbands; b++) {
            if (owner[b] >= world) return fail(SDFHIP_ERR_ARG, "deinterleave_sparse: band %u belongs to rank %u of %u", b, (unsigned)owner[b], world);
            M.src[b] = (uint16_t)((uint32_t)owner[b] << 10 | have[owner[b]]++);
        }
        M.n = nbands;
        need_rows = 0;
        for (uint32_t r = 0; r < world; r++) need_rows = have[r] * band_rows > need_rows ? have[r] * band_rows : need_rows;
    }
    if (rows_per_rank < need_rows)
        return fail(SDFHIP_ERR_ARG, "deinterleave_sparse: rows_per_rank %u < %u needed", rows_per_rank, need_rows);
    DeviceGuard g(device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "deinterleave_sparse: hipSetDevice(%d) failed", device);
    const SparseLayout L = sparse_layout(width, rows_per_rank, capacity);
    size_t total = (size_t)width * height * frames;
    uint32_t blocks = (uint32_t)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(k_deinterleave_sparse, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)d_gathered,
                       (float4 *)d_frame, width, height, band_rows, world, frames, L, M, d_overflow);
    HIP_TRY(hipGetLastError());
    return SDFHIP_OK;
}

// ---- sparse shares written by the march kernel itself (OUT_SPARSE) ---------------------------------------------
extern "C" uint64_t sdfhip_sparse2_bytes(uint32_t width, uint32_t rows, uint32_t frames, uint32_t capacity)
{
    return (uint64_t)sparse2_layout(width, rows, frames, capacity).bytes;
}

extern "C" uint64_t sdfhip_sparse2_floats_offset(uint32_t width, uint32_t rows, uint32_t frames)
{
    return (uint64_t)sparse2_layout(width, rows, frames, 0).off_floats;
}

extern "C" int sdfhip_render_sparse_device(sdfhip_scene *s, const sdfhip_info *infos, uint32_t n_frames, uint32_t width,
                                           uint32_t height, uint32_t band_rows, const uint16_t *bands, uint32_t n_bands,
                                           uint32_t nrows_out, uint32_t capacity, uint32_t count_base, uint32_t flags, void *d_share, void *stream)
{
    if (!s || !infos || !bands || !d_share) return fail(SDFHIP_ERR_ARG, "render_sparse_device: null argument");
    if (capacity == 0) return fail(SDFHIP_ERR_ARG, "render_sparse_device: capacity 0");
    std::lock_guard<std::mutex> lk(s->lock);
    DeviceGuard g(s->device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "render_sparse_device: hipSetDevice(%d) failed", s->device);
    return render_impl(s, infos, width, height, band_rows, 0, 1, nrows_out, flags, reinterpret_cast<float *>(d_share),
                       (hipStream_t)stream, nullptr, nullptr, n_frames, bands, n_bands, capacity, true, count_base);
}

extern "C" int sdfhip_deinterleave_sparse2_device(int device, const void *const *d_shares, void *d_frame, uint32_t width,
                                                  uint32_t height, uint32_t band_rows, uint32_t world, uint32_t rows_per_rank,
                                                  const uint8_t *owner, uint32_t capacity, uint32_t frames, uint32_t flags,
                                                  int only_rank, uint32_t *counts_out, void *stream)
{
    if (frames == 0 || frames > (uint32_t)MAX_BATCH || !d_shares || !d_frame || width == 0 || height == 0 || band_rows == 0 || world == 0)
        return fail(SDFHIP_ERR_ARG, "deinterleave_sparse2: null or zero argument");
    if (world > MULTI_MAX_RANKS) return fail(SDFHIP_ERR_ARG, "deinterleave_sparse2: %u ranks (at most %u)", world, MULTI_MAX_RANKS);
    if (only_rank >= (int)world) return fail(SDFHIP_ERR_ARG, "deinterleave_sparse2: rank %d of %u", only_rank, world);
    const uint32_t nbands = (height + band_rows - 1) / band_rows;
    BandMap M;
    M.n = 0;
    memset(M.src, 0, sizeof M.src);
    uint32_t need_rows = ((nbands + world - 1) / world) * band_rows;
    if (owner) {
        if (nbands > (uint32_t)MAX_BAND_LIST)
            return fail(SDFHIP_ERR_ARG, "deinterleave_sparse2: %u bands (max %d)", nbands, MAX_BAND_LIST);
        uint32_t have[MULTI_MAX_RANKS] = { 0 };
        for (uint32_t b = 0; b < nbands; b++) {
            if (owner[b] >= world) return fail(SDFHIP_ERR_ARG, "deinterleave_sparse2: band %u belongs to rank %u of %u", b, (unsigned)owner[b], world);
            M.src[b] = (uint16_t)((uint32_t)owner[b] << 10 | have[owner[b]]++);
        }
        M.n = nbands;
        need_rows = 0;
        for (uint32_t r = 0; r < world; r++) need_rows = have[r] * band_rows > need_rows ? have[r] * band_rows : need_rows;
    }
    if (rows_per_rank < need_rows)
        return fail(SDFHIP_ERR_ARG, "deinterleave_sparse2: rows_per_rank %u < %u needed", rows_per_rank, need_rows);
    ShareTable S;
    memset(&S, 0, sizeof S);
    for (uint32_t r = 0; r < world; r++) {
        if (!d_shares[r] && (only_rank < 0 || (uint32_t)only_rank == r)) return fail(SDFHIP_ERR_ARG, "deinterleave_sparse2: rank %u has no share", r);
        S.p[r] = static_cast<const uint8_t *>(d_shares[r]);
    }
    DeviceGuard g(device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "deinterleave_sparse2: hipSetDevice(%d) failed", device);
    const Sparse2Layout L = sparse2_layout(width, rows_per_rank, frames, capacity);
    const size_t total = (size_t)width * height * frames;
    const uint32_t blocks = (uint32_t)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    const uint32_t only = only_rank < 0 ? 0xFFFFFFFFu : (uint32_t)only_rank;
    auto q = [](float c) { float v = powf(c, 1.0f / 2.2f); v = v > 1.0f ? 1.0f : (v > 0.0f ? v : 0.0f); return (uint32_t)(v * 255.0f + 0.5f); };
    const uint32_t sky8 = q(0.005f) | (q(0.01f) << 8) | (q(0.2f) << 16);
    if (flags & SDFHIP_FLAG_DISPLAY_DEBUG)
        hipLaunchKernelGGL((k_deinterleave_sparse2<OUT_HEAT8>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, d_frame, width, height, band_rows, world, L, M, only, sky8, counts_out);
    else if (flags & SDFHIP_FLAG_DISPLAY)
        hipLaunchKernelGGL((k_deinterleave_sparse2<OUT_GAMMA8>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, d_frame, width, height, band_rows, world, L, M, only, sky8, counts_out);
    else
        hipLaunchKernelGGL((k_deinterleave_sparse2<OUT_RGBA32F>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, d_frame, width, height, band_rows, world, L, M, only, sky8, counts_out);
    HIP_TRY(hipGetLastError());
    return SDFHIP_OK;
}

extern "C" int sdfhip_debug_tile_order(sdfhip_scene *s, const uint32_t *d_perm, uint16_t *d_cost)
{
    if (!s) return fail(SDFHIP_ERR_ARG, "debug_tile_order: null scene");
    std::lock_guard<std::mutex> lk(s->lock);
    s->dbg_tile_perm = d_perm; s->dbg_tile_cost = d_cost;
    return SDFHIP_OK;
}

extern "C" int sdfhip_debug_step_classes(sdfhip_scene *s, void *stream, uint64_t *out6)
{
    if (!s || !out6) return fail(SDFHIP_ERR_ARG, "debug_step_classes: null argument");
    std::lock_guard<std::mutex> lk(s->lock);
    DeviceGuard g(s->device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "debug_step_classes: hipSetDevice(%d) failed", s->device);
    for (int i = 0; i < s->n_scratch; i++)
        if (s->scratch[i].stream == (hipStream_t)stream) {
            HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
            const uint32_t *c = s->scratch[i].ctl + sdfhip_scene::CTL_HIT_WORDS + sdfhip_scene::CTL_QUEUE_WORDS + sdfhip_scene::CTL_PT_WORDS;
            HIP_TRY(hipMemcpy(out6, reinterpret_cast<const unsigned long long *>(c) + 6, 6 * sizeof(uint64_t), hipMemcpyDeviceToHost));
            return SDFHIP_OK;
        }
    return fail(SDFHIP_ERR_ARG, "debug_step_classes: no counting render has run on that stream of this scene");
}

extern "C" int sdfhip_debug_unorm_table(int device, float *out256)
{
    if (!out256) return fail(SDFHIP_ERR_ARG, "debug_unorm_table: null argument");
    DeviceGuard g(device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "debug_unorm_table: hipSetDevice(%d) failed", device);
    float *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 256 * sizeof(float)));
    hipLaunchKernelGGL(k_unorm_table, dim3(1), dim3(256), 0, 0, d);
    hipError_t e = hipMemcpy(out256, d, 256 * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(SDFHIP_ERR_DEVICE, "debug_unorm_table: %s", hipGetErrorString(e));
    return SDFHIP_OK;
}
