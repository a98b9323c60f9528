// Rank 0's side of the tile gather (gfx950): gathered band buffers -> frame rows, and the expansion of the sparse shares the
// march kernel writes (OUT_SPARSE, raymarch_kernels.h).  Templates and plain structs only (gather.hip instantiates them).
#pragma once
#include "raymarch_kernels.h"

#include <type_traits>

namespace sdfhip {

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

// The same with the frame's geometry in the launch grid (see k_expand_sparse2 below: no division per pixel): workgroup
// (blockIdx.x, band * parts + part, frame) copies up to rows_per_block rows of 256 neighbouring columns.  Needs the band map.
template <class T>
__global__ __launch_bounds__(256) void k_deinterleave_rows(const T *__restrict__ gathered, T *__restrict__ frame, uint32_t width, uint32_t height,
                                                           uint32_t band_rows, uint32_t rows_per_block, uint32_t parts, uint32_t rows_per_rank,
                                                           uint32_t frames, const BandMap M, uint32_t only_rank)
{
    const uint32_t band = blockIdx.y / parts, part = blockIdx.y - band * parts, f = blockIdx.z, x = blockIdx.x * 256u + threadIdx.x;
    const uint32_t e = M.src[band];
    uint32_t rank = e >> 10;
    const uint32_t lband = e & 1023u;
    if (only_rank != 0xFFFFFFFFu) { if (rank != only_rank) return; rank = 0; }
    const uint32_t first = part * rows_per_block, y0 = band * band_rows + first;
    if (x >= width || y0 >= height) return;
    const uint32_t rows = min(rows_per_block, height - y0);
    const T *__restrict__ src = gathered + (((size_t)rank * frames + f) * rows_per_rank + (size_t)lband * band_rows + first) * width + x;
    T *__restrict__ dst = frame + ((size_t)f * height + y0) * width + x;
    for (uint32_t j = 0; j < rows; j++) dst[(size_t)j * width] = src[(size_t)j * width];
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
// The same expansion with the frame's geometry in the launch grid instead of in divisions: workgroup (blockIdx.x, band, frame) takes
// 256 neighbouring columns of one band of one frame and walks the band's rows, so which rank rendered the band, where its rows lie in
// that rank's share and the tile row are per-workgroup scalars (the per-pixel form above spends five integer divisions, two of them
// 64-bit, on every pixel -- about as long as the frame's bytes take to write); a tile's mask and slot base are loaded once per eight
// rows.  Needs the band map (frames of at most MAX_BAND_LIST bands: the host fills it for round-robin deals too).
template <int MODE>
__global__ __launch_bounds__(256) void k_expand_sparse2(const ShareTable S, void *__restrict__ frame, uint32_t width, uint32_t height,
                                                        uint32_t band_rows, uint32_t rows_per_block, uint32_t parts, uint32_t world,
                                                        Sparse2Layout L, const BandMap M, uint32_t only_rank, uint32_t sky8,
                                                        uint32_t *__restrict__ counts)
{
    if (counts && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < world && S.p[threadIdx.x])
        counts[threadIdx.x] = *reinterpret_cast<const uint32_t *>(S.p[threadIdx.x]);
    // blockIdx.y = band * parts + part: a band of 16 rows goes to two workgroups of one tile row each (rows_per_block = 8 when the
    // bands are whole tile rows, else the band)
    const uint32_t band = blockIdx.y / parts, part = blockIdx.y - band * parts, f = blockIdx.z, x = blockIdx.x * 256u + threadIdx.x;
    const uint32_t e = M.src[band], rank = e >> 10, lband = e & 1023u;
    if (only_rank != 0xFFFFFFFFu && rank != only_rank) return;
    const uint32_t first = part * rows_per_block, y0 = band * band_rows + first;
    if (x >= width || y0 >= height) return;
    const uint8_t *__restrict__ src = S.p[rank];
    const uint32_t rows = min(rows_per_block, height - y0), yl0 = lband * band_rows + first;
    const unsigned long long *__restrict__ masks = reinterpret_cast<const unsigned long long *>(src + L.off_masks);
    const uint32_t *__restrict__ bases = reinterpret_cast<const uint32_t *>(src + L.off_bases);
    const float *__restrict__ floats = reinterpret_cast<const float *>(src + L.off_floats);
    const size_t ft0 = (size_t)f * L.tiles + (x >> 3);
    size_t i = ((size_t)f * height + y0) * width + x;
    unsigned long long m = 0;
    uint32_t base = 0;
    for (uint32_t j = 0; j < rows; j++, i += width) {
        const uint32_t yl = yl0 + j;
        const size_t ft = ft0 + (size_t)(yl >> 3) * L.tiles_x;
        if (j == 0 || (yl & 7u) == 0) { m = masks[ft]; base = bases[ft]; }
        const uint32_t bit = (yl & 7u) * 8u + (x & 7u);
        const uint32_t code = src[L.off_codes + ft * 64 + bit];
        float a = 0.0f;
        if ((m >> bit) & 1ull) {
            const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << bit) - 1ull));
            if (slot < L.capacity) a = floats[slot];
        }
        const float4 v = wire_expand(a, code);
        if (MODE == OUT_RGBA32F) {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4 *>(frame) + i);
        } else {
            uint32_t q;
            if (MODE == OUT_HEAT8) q = heat8(v.w);
            else if (code > 140u) q = sky8 | alpha8(v.w);
            else { const uint32_t g = gamma8(a); q = g | (g << 8) | (g << 16) | alpha8(v.w); }
            __builtin_nontemporal_store(q, reinterpret_cast<uint32_t *>(frame) + i);
        }
    }
}
}  // namespace sdfhip
