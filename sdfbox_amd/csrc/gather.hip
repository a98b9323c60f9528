// libsdfhip.so, device half, host side: rank 0's side of the tile gather -- gathered band buffers back into row
// order, and the expansion of the sparse shares the ranks' march kernels write (sdfhip_render_sparse_device, render.hip).
#include "gather_kernels.h"
#include "scene.h"
#include "abi_guard.h"

#include <cmath>
#include <cstring>

using namespace sdfhip;

int sdfhip::deinterleave_impl(int device, const void *d_gathered, void *d_frame, uint32_t width, uint32_t height,
                              uint32_t band_rows, uint32_t world, uint32_t rows_per_rank, const uint8_t *owner,
                              uint32_t pixel_bytes, uint32_t frames, void *stream, uint32_t only_rank)
{
    if (only_rank != 0xFFFFFFFFu && only_rank >= world)
        return fail(SDFHIP_ERR_ARG, "deinterleave_share: rank %u of %u", only_rank, world);
    if (frames == 0) return fail(SDFHIP_ERR_ARG, "deinterleave: frames must be >= 1");
    if (!d_gathered || !d_frame || width == 0 || height == 0 || band_rows == 0 || world == 0)
        return fail(SDFHIP_ERR_ARG, "deinterleave: null or zero argument");
#ifdef SDFHIP_EXPERIMENTS
    if (pixel_bytes != 16 && pixel_bytes != 4 && pixel_bytes != 5)
        return fail(SDFHIP_ERR_ARG, "deinterleave: pixel_bytes must be 16 (RGBA32F), 5 (wire) or 4 (RGBA8), got %u", pixel_bytes);
#else
    if (pixel_bytes != 16 && pixel_bytes != 4)
        return fail(SDFHIP_ERR_ARG, "deinterleave: pixel_bytes must be 16 (RGBA32F) or 4 (RGBA8), got %u", pixel_bytes);
#endif
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
    if (!g.ok) return (void)hipGetLastError(), fail(SDFHIP_ERR_DEVICE, "deinterleave: hipSetDevice(%d) failed", device);
    if (nbands <= (uint32_t)MAX_BAND_LIST && world <= 64 && (pixel_bytes == 16 || pixel_bytes == 4)) {
        // the frame's geometry in the launch grid (k_deinterleave_rows): a round-robin deal gets its band map here
        if (!M.n) {
            for (uint32_t b = 0; b < nbands; b++) M.src[b] = (uint16_t)((b % world) << 10 | (b / world));
            M.n = nbands;
        }
        const uint32_t rows_per_block = band_rows % 8u == 0 ? 8u : band_rows, parts = band_rows / rows_per_block;
        const dim3 grid((width + 255u) / 256u, nbands * parts, frames);
        if (pixel_bytes == 16)
            hipLaunchKernelGGL((k_deinterleave_rows<float4>), grid, dim3(256), 0, (hipStream_t)stream, (const float4 *)d_gathered, (float4 *)d_frame,
                               width, height, band_rows, rows_per_block, parts, rows_per_rank, frames, M, only_rank);
        else
            hipLaunchKernelGGL((k_deinterleave_rows<uint32_t>), grid, dim3(256), 0, (hipStream_t)stream, (const uint32_t *)d_gathered, (uint32_t *)d_frame,
                               width, height, band_rows, rows_per_block, parts, rows_per_rank, frames, M, only_rank);
        HIP_TRY(hipGetLastError());
        return SDFHIP_OK;
    }
    size_t total = (size_t)width * height * frames;
    uint32_t blocks = (uint32_t)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (pixel_bytes == 16)
        hipLaunchKernelGGL((k_deinterleave<float4, float4>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const float4 *)d_gathered, (float4 *)d_frame, width, height, band_rows,
                           world, rows_per_rank, frames, M, only_rank);
#ifdef SDFHIP_EXPERIMENTS
    else if (pixel_bytes == 5)
        hipLaunchKernelGGL((k_deinterleave<WirePlanes, float4>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const WirePlanes *)d_gathered, (float4 *)d_frame, width, height, band_rows,
                           world, rows_per_rank, frames, M, only_rank);
#endif
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
try {
    return deinterleave_impl(device, d_gathered, d_frame, width, height, band_rows, world, rows_per_rank, nullptr,
                             pixel_bytes, frames, stream);
}
SDFHIP_ABI_CATCH(sdfhip_deinterleave_device)

extern "C" int sdfhip_deinterleave_bands_device(int device, const void *d_gathered, void *d_frame,
                                                uint32_t width, uint32_t height, uint32_t band_rows,
                                                uint32_t world, uint32_t rows_per_rank, const uint8_t *owner,
                                                uint32_t pixel_bytes, uint32_t frames, void *stream)
try {
    if (!owner) return fail(SDFHIP_ERR_ARG, "deinterleave_bands: null owner table");
    return deinterleave_impl(device, d_gathered, d_frame, width, height, band_rows, world, rows_per_rank, owner,
                             pixel_bytes, frames, stream);
}
SDFHIP_ABI_CATCH(sdfhip_deinterleave_bands_device)

// ---- sparse shares written by the march kernel itself (OUT_SPARSE) ---------------------------------------------
extern "C" uint64_t sdfhip_sparse2_bytes(uint32_t width, uint32_t rows, uint32_t frames, uint32_t capacity)
try {
    return (uint64_t)sparse2_layout(width, rows, frames, capacity).bytes;
}
SDFHIP_ABI_CATCH_AS(sdfhip_sparse2_bytes, 0)

extern "C" uint64_t sdfhip_sparse2_floats_offset(uint32_t width, uint32_t rows, uint32_t frames)
try {
    return (uint64_t)sparse2_layout(width, rows, frames, 0).off_floats;
}
SDFHIP_ABI_CATCH_AS(sdfhip_sparse2_floats_offset, 0)

extern "C" int sdfhip_deinterleave_sparse2_device(int device, const void *const *d_shares, void *d_frame, uint32_t width,
                                                  uint32_t height, uint32_t band_rows, uint32_t world, uint32_t rows_per_rank,
                                                  const uint8_t *owner, uint32_t capacity, uint32_t frames, uint32_t flags,
                                                  int only_rank, uint32_t *counts_out, void *stream)
try {
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
    if (!g.ok) return (void)hipGetLastError(), fail(SDFHIP_ERR_DEVICE, "deinterleave_sparse2: hipSetDevice(%d) failed", device);
    const Sparse2Layout L = sparse2_layout(width, rows_per_rank, frames, capacity);
    const size_t total = (size_t)width * height * frames;
    const uint32_t blocks = (uint32_t)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    const uint32_t only = only_rank < 0 ? 0xFFFFFFFFu : (uint32_t)only_rank;
    auto q = [](float c) { float v = powf(c, 1.0f / 2.2f); v = v > 1.0f ? 1.0f : (v > 0.0f ? v : 0.0f); return (uint32_t)(v * 255.0f + 0.5f); };
    const uint32_t sky8 = q(0.005f) | (q(0.01f) << 8) | (q(0.2f) << 16);
    if (nbands <= (uint32_t)MAX_BAND_LIST) {           // (512 bands x at most band_rows / 8 parts: far inside the grid's 65 535)
        // the frame's geometry in the launch grid (k_expand_sparse2): a round-robin deal gets its band map here
        if (!M.n) {
            for (uint32_t b = 0; b < nbands; b++) M.src[b] = (uint16_t)((b % world) << 10 | (b / world));
            M.n = nbands;
        }
        const uint32_t rows_per_block = band_rows % 8u == 0 ? 8u : band_rows, parts = band_rows / rows_per_block;
        const dim3 grid((width + 255u) / 256u, nbands * parts, frames);
        if (flags & SDFHIP_FLAG_DISPLAY_DEBUG)
            hipLaunchKernelGGL((k_expand_sparse2<OUT_HEAT8>), grid, dim3(256), 0, (hipStream_t)stream, S, d_frame, width, height, band_rows, rows_per_block, parts, world, L, M, only, sky8, counts_out);
        else if (flags & SDFHIP_FLAG_DISPLAY)
            hipLaunchKernelGGL((k_expand_sparse2<OUT_GAMMA8>), grid, dim3(256), 0, (hipStream_t)stream, S, d_frame, width, height, band_rows, rows_per_block, parts, world, L, M, only, sky8, counts_out);
        else
            hipLaunchKernelGGL((k_expand_sparse2<OUT_RGBA32F>), grid, dim3(256), 0, (hipStream_t)stream, S, d_frame, width, height, band_rows, rows_per_block, parts, world, L, M, only, sky8, counts_out);
        HIP_TRY(hipGetLastError());
        return SDFHIP_OK;
    }
    if (flags & SDFHIP_FLAG_DISPLAY_DEBUG)
        hipLaunchKernelGGL((k_deinterleave_sparse2<OUT_HEAT8>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, d_frame, width, height, band_rows, world, L, M, only, sky8, counts_out);
    else if (flags & SDFHIP_FLAG_DISPLAY)
        hipLaunchKernelGGL((k_deinterleave_sparse2<OUT_GAMMA8>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, d_frame, width, height, band_rows, world, L, M, only, sky8, counts_out);
    else
        hipLaunchKernelGGL((k_deinterleave_sparse2<OUT_RGBA32F>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, d_frame, width, height, band_rows, world, L, M, only, sky8, counts_out);
    HIP_TRY(hipGetLastError());
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_deinterleave_sparse2_device)
