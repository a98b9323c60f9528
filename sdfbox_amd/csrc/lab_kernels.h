// LABORATORY (built only with -DSDFHIP_EXPERIMENTS into libsdfhip_lab.so; include/sdfhip_experimental.h): kernels of the
// measured-and-dropped forms that stay as bit-identical A/B knobs -- the grid's second form (4-byte words + sample records),
// round 2's sparse wire format made by three compaction kernels, the shadow-ray queue's second kernel, a test hook.
// DESIGN.md sections 4.3, 4.7 and 5 hold the measurements.  Nothing here is compiled into libsdfhip.so.
#pragma once
#ifndef SDFHIP_EXPERIMENTS
#error "lab_kernels.h belongs to the experiments build (-DSDFHIP_EXPERIMENTS)"
#endif
#include "gather_kernels.h"

namespace sdfhip {

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
// (the upload's k_split_scan again: a non-template kernel lives in ONE translation unit)
// exclusive scan of n values in place by ONE workgroup of 1024 threads; the total goes to v[n]
__global__ __launch_bounds__(1024) void k_d4_scan(uint32_t *__restrict__ v, uint32_t n)
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

// sdfhip_debug_touch_*: the distinct lines of a pair of bitmaps ([8 XCDs][words] each) -> out[0..1] chip-wide (the OR over the
// XCDs), out[2..3] summed over the XCDs; the bitmaps are cleared for the next phase
__global__ __launch_bounds__(256) void k_touch_count(uint32_t *a, uint32_t words_a, uint32_t *b, uint32_t words_b, unsigned long long *out)
{
    unsigned long long any[2] = { 0, 0 }, sum[2] = { 0, 0 };
    for (int k = 0; k < 2; k++) {
        uint32_t *bits = k ? b : a;
        const uint32_t words = k ? words_b : words_a;
        if (!bits) continue;
        for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (size_t)gridDim.x * blockDim.x) {
            uint32_t u = 0;
            for (uint32_t x = 0; x < 8; x++) {
                const uint32_t v = bits[(size_t)x * words + w];
                u |= v; sum[k] += (unsigned long long)__popc(v);
                bits[(size_t)x * words + w] = 0u;
            }
            any[k] += (unsigned long long)__popc(u);
        }
    }
    for (int k = 0; k < 2; k++) {
        for (int o = 32; o > 0; o >>= 1) { any[k] += __shfl_down(any[k], o); sum[k] += __shfl_down(sum[k], o); }
        if ((threadIdx.x & 63u) == 0u) { if (any[k]) atomicAdd(&out[k], any[k]); if (sum[k]) atomicAdd(&out[2 + k], sum[k]); }
    }
}

__global__ void k_unorm_table(float *out)
{
    out[threadIdx.x] = unorm8((float)threadIdx.x);
}

}  // namespace sdfhip
