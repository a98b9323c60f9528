// LABORATORY: the host side of the experiments build (libsdfhip_lab.so, -DSDFHIP_EXPERIMENTS; include/sdfhip_experimental.h).
// Everything here was built, proved bit-identical to the oracle and measured slower than (or equal to) the product's path;
// it stays as A/B knobs for measurements and as regression tests of the alternatives (DESIGN.md sections 4.2-4.7, 5).
// Nothing in this file is compiled into libsdfhip.so.
#ifndef SDFHIP_EXPERIMENTS
#error "lab.hip belongs to the experiments build (-DSDFHIP_EXPERIMENTS)"
#endif
#include "lab_kernels.h"
#include "scene.h"
#include "abi_guard.h"
#include "../../include/sdfhip_experimental.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <new>

using namespace sdfhip;

// The grid's second form (CursorFF, raymarch_device.h), made from the 16-byte cells once they exist -- a dense grid as deep as the
// tree, or a split one: a word per cell of the deepest level and a 64-byte sample record per non-flat leaf.  An accelerator of an
// accelerator: trees deeper than 10 levels (4 GB of words at depth 10), a grid that is not as deep as the tree, or too little
// memory do without it, and the default kernel reads the 16-byte cells.
// MEASURED SLOWER than the 16-byte cells on the bench frames (30 % fewer VALU instructions, 1.7 x the L1 tag
// lookups and 2.7 x the HBM bytes; 0.112 against 0.090 ms per 1080p frame), so it is built only when SDFHIP_SAMPLE_RECORDS=1 is in
// the environment at upload: an experiment that stays bit-identical (tests/test_gpu_parity.py::test_pre_decoded_cells...).
void sdfhip::build_dense4(sdfhip_scene *s)
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
        hipLaunchKernelGGL(k_d4_scan, dim3(1), dim3(1024), 0, s->stream, d_chunks, n_chunks);
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

namespace {

// k_march + k_shadow for one cursor kind and counting choice, by output mode (SDFHIP_TUNE_SHADOW_QUEUE)
template <int CUR, bool COUNT>
void launch_queued(uint32_t mode, dim3 grid, dim3 shade_grid, hipStream_t st, const RenderParams &P)
{
    auto go = [&](auto march, auto shade) {
        hipLaunchKernelGGL(march, march_grid(P, grid), dim3(64), 0, st, P);
        hipLaunchKernelGGL(shade, shade_grid, dim3(64), 0, st, P);
    };
    if (mode == OUT_RGBA32F)     go(k_march<CUR, COUNT, OUT_RGBA32F, true>, k_shadow<CUR, COUNT, OUT_RGBA32F>);
    else if (mode == OUT_GAMMA8) go(k_march<CUR, COUNT, OUT_GAMMA8, true>, k_shadow<CUR, COUNT, OUT_GAMMA8>);
    else if (mode == OUT_HEAT8)  go(k_march<CUR, COUNT, OUT_HEAT8, true>, k_shadow<CUR, COUNT, OUT_HEAT8>);
    else                         go(k_march<CUR, COUNT, OUT_WIRE, true>, k_shadow<CUR, COUNT, OUT_WIRE>);
}

// the default kernel with wire pixels as its output (round 1's gather format)
template <int CUR, bool COUNT>
void launch_march_wire(dim3 grid, hipStream_t st, const RenderParams &P)
{
    hipLaunchKernelGGL((k_march<CUR, COUNT, OUT_WIRE, false>), march_grid(P, grid), dim3(64), 0, st, P);
}

// the default kernel through the grid's second form, 4-byte words + sample records (CursorFF): not counting, shadow rays marched in the wave
void launch_fast(uint32_t mode, dim3 grid, hipStream_t st, const RenderParams &P)
{
    if (mode == OUT_RGBA32F)     hipLaunchKernelGGL((k_march<CUR_DENSE4, false, OUT_RGBA32F, false>), march_grid(P, grid), dim3(64), 0, st, P);
    else if (mode == OUT_GAMMA8) hipLaunchKernelGGL((k_march<CUR_DENSE4, false, OUT_GAMMA8, false>), march_grid(P, grid), dim3(64), 0, st, P);
    else if (mode == OUT_HEAT8)  hipLaunchKernelGGL((k_march<CUR_DENSE4, false, OUT_HEAT8, false>), march_grid(P, grid), dim3(64), 0, st, P);
    else if (mode == OUT_SPARSE) hipLaunchKernelGGL((k_march<CUR_DENSE4, false, OUT_SPARSE, false>), march_grid(P, grid), dim3(64), 0, st, P);
    else                         hipLaunchKernelGGL((k_march<CUR_DENSE4, false, OUT_WIRE, false>), march_grid(P, grid), dim3(64), 0, st, P);
}

// round 1's one-kernel form for any cursor and workgroup size
template <int CUR, bool COUNT>
void launch_plain(int bt, dim3 grid, hipStream_t st, const RenderParams &P)
{
    if (bt == 64)       hipLaunchKernelGGL((k_plain<CUR, COUNT, 64>), grid, dim3(64), 0, st, P);
    else if (bt == 128) hipLaunchKernelGGL((k_plain<CUR, COUNT, 128>), grid, dim3(128), 0, st, P);
    else                hipLaunchKernelGGL((k_plain<CUR, COUNT, 256>), grid, dim3(256), 0, st, P);
}

}  // namespace

int sdfhip::launch_experiment(sdfhip_scene *s, const RenderCall &c, RenderParams &P, int cur, bool count, dim3 grid, bool *launched,
                              sdfhip_scene::Scratch **scp)
{
    *launched = false;
    const uint32_t flags = c.flags;
    hipStream_t st = c.st;
    const bool compact = (flags & SDFHIP_FLAG_COMPACT) != 0, wire = (flags & SDFHIP_FLAG_WIRE) != 0;
    const bool display = (flags & (SDFHIP_FLAG_DISPLAY | SDFHIP_FLAG_DISPLAY_DEBUG)) != 0;
    if (c.sparse && (wire || (flags & (SDFHIP_TUNE_SHADOW_QUEUE | SDFHIP_TUNE_ONE_KERNEL))))
        return fail(SDFHIP_ERR_ARG, "render_sparse: sparse shares come from the default kernel only (no A/B forms)");
    if (wire && (compact || c.pt || display))
        return fail(SDFHIP_ERR_ARG, "render: wire pixels come from the plain kernel only, without the display pass");
    if (wire && ((uint64_t)c.nrows_out * c.width) % 4 != 0)
        return fail(SDFHIP_ERR_ARG, "render: wire buffers need nrows_out * width to be a multiple of 4 (the byte plane follows the float plane)");
    if (wire) P.out_mode = OUT_WIRE;
    const uint32_t mode = P.out_mode;
    P.d4 = s->d_d4;
    // (a word of d4 is TAG + the record's index in 16-byte units: the kernel adds the whole word to this pointer)
    P.recs = reinterpret_cast<const uint4 *>(reinterpret_cast<uintptr_t>(s->d_recs) - ((uintptr_t)D4_TAG << 4));
    P.tile_order = (flags >> SDFHIP_TUNE_ORDER_SHIFT) & 0xF;
    const uint32_t btsel = (flags >> SDFHIP_TUNE_BLOCK_SHIFT) & 0xF;
    const int bt = btsel == 3 ? 256 : (btsel == 2 ? 128 : 64);
    P.tile_perm = c.n_frames == 1 ? s->dbg_tile_perm : nullptr;      // (the experiment hook is for single frames: its arrays hold one frame's tiles)
    //  the hook's entries pack tile_row << 16 | tile_col and its launch has grid.z = grid.x / 8: refuse what neither can hold
    if (P.tile_perm && (P.tiles_x >= 65536u || P.tiles_y >= 65536u || grid.x / 8u > 65535u))
        return fail(SDFHIP_ERR_ARG, "render: the tile-order hook (sdfhip_debug_tile_order) takes frames of at most 65 535 tile rows / columns and "
                                    "8 x 65 535 workgroups (this one: %u x %u tiles, %u workgroups)", P.tiles_x, P.tiles_y, grid.x);
    P.perm_per_label = grid.x / 8u;                                  // (the hook's array: [XCD label][slot], as the library's own)
    P.tile_cost = c.n_frames == 1 ? s->dbg_tile_cost : nullptr;
    const bool grid_lookup = cur == CUR_STACK_FULL || cur == CUR_STACK_SPLIT;
    // where the product launches k_march
    const bool two = grid_lookup && !compact && !c.pt && bt == 64 && P.tile_order == 0 && !(flags & SDFHIP_TUNE_ONE_KERNEL);
    if (two) {
        if (flags & SDFHIP_TUNE_SHADOW_QUEUE) {
            // SDFHIP_SHADOW_MIN_LANES=T: waves with at least T shadow rays march them in place (default here 65: every ray is queued, round 2's form;
            // the product's SDFHIP_FLAG_COMPACT is this pair of kernels with T = COMPACT_MIN_LANES)
            uint32_t hit_min = 65u;
            if (const char *env = getenv("SDFHIP_SHADOW_MIN_LANES")) { const int v = atoi(env); if (v >= 1 && v <= 65) hit_min = (uint32_t)v; }
            dim3 shade_grid;
            int rcs = prepare_shadow_queue(s, c, P, grid, hit_min, scp, &shade_grid);
            if (rcs != SDFHIP_OK) return rcs;
            if (cur == CUR_STACK_SPLIT) { if (count) launch_queued<CUR_STACK_SPLIT, true>(mode, grid, shade_grid, st, P); else launch_queued<CUR_STACK_SPLIT, false>(mode, grid, shade_grid, st, P); }
            else                        { if (count) launch_queued<CUR_STACK_FULL, true>(mode, grid, shade_grid, st, P); else launch_queued<CUR_STACK_FULL, false>(mode, grid, shade_grid, st, P); }
            *launched = true;
        } else if (!count && s->d_d4 && !(flags & SDFHIP_TUNE_BYTE_CELLS)) {
            launch_fast(mode, grid, st, P);
            *launched = true;
        } else if (wire) {
            if (cur == CUR_STACK_SPLIT) { if (count) launch_march_wire<CUR_STACK_SPLIT, true>(grid, st, P); else launch_march_wire<CUR_STACK_SPLIT, false>(grid, st, P); }
            else                        { if (count) launch_march_wire<CUR_STACK_FULL, true>(grid, st, P); else launch_march_wire<CUR_STACK_FULL, false>(grid, st, P); }
            *launched = true;
        }
        return SDFHIP_OK;                             // (else: the product's k_march, with the tile hooks above)
    }
    if (c.pt) {
        if (grid_lookup && (flags & SDFHIP_TUNE_ONE_KERNEL)) {      // round 1's one-kernel path tracer on a scene the pipeline would take
            const dim3 g1(8u * ((P.tiles_y + 7u) / 8u) * P.tiles_x);
            auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, g1, dim3(64), 0, st, P); };
            if (cur == CUR_STACK_SPLIT) { if (count) go(k_path<CUR_STACK_SPLIT, true>); else go(k_path<CUR_STACK_SPLIT, false>); }
            else                        { if (count) go(k_path<CUR_STACK_FULL, true>); else go(k_path<CUR_STACK_FULL, false>); }
            *launched = true;
        }
        return SDFHIP_OK;
    }
    if (compact) return SDFHIP_OK;                    // the product's: k_march -> k_shadow, or k_compact (launch_fallback; SDFHIP_TUNE_PERSISTENT_WAVES)
    // the one-kernel form: other workgroup sizes cover 16-pixel-wide tiles
    const uint32_t tile_w = bt >= 128 ? 16u : 8u, tile_h = (uint32_t)bt / 8u / (tile_w / 8u);
    P.tiles_x = (c.width + tile_w - 1) / tile_w;
    P.tiles_y = (c.nrows_out + tile_h - 1) / tile_h;
    P.n_tiles = P.tiles_x * P.tiles_y;
    const dim3 gp(P.tile_order == 0 ? 8u * ((P.tiles_y + 7u) / 8u) * P.tiles_x : P.n_tiles, c.n_frames);
    if ((flags & SDFHIP_TUNE_LDS_TOP) && cur == CUR_STACK && s->d_top && s->top_level <= 3 && !count) {
        // measurement variant: the top grid staged in LDS per workgroup (64- or 256-thread workgroups)
        if (bt == 256) hipLaunchKernelGGL((k_plain<CUR_STACK, false, 256, true>), gp, dim3(256), 0, st, P);
        else           hipLaunchKernelGGL((k_plain<CUR_STACK, false, 64, true>), gp, dim3(64), 0, st, P);
    }
    else if (cur == CUR_STACK_SPLIT) { if (count) launch_plain<CUR_STACK_SPLIT, true>(bt, gp, st, P); else launch_plain<CUR_STACK_SPLIT, false>(bt, gp, st, P); }
    else if (cur == CUR_STACK_FULL)  { if (count) launch_plain<CUR_STACK_FULL, true>(bt, gp, st, P); else launch_plain<CUR_STACK_FULL, false>(bt, gp, st, P); }
    else if (cur == CUR_STACK)       { if (count) launch_plain<CUR_STACK, true>(bt, gp, st, P); else launch_plain<CUR_STACK, false>(bt, gp, st, P); }
    else                             { if (count) launch_plain<CUR_GENERIC, true>(bt, gp, st, P); else launch_plain<CUR_GENERIC, false>(bt, gp, st, P); }
    *launched = true;
    return SDFHIP_OK;
}

extern "C" int sdfhip_deinterleave_share_device(int device, const void *d_share, void *d_frame,
                                                uint32_t width, uint32_t height, uint32_t band_rows,
                                                uint32_t world, uint32_t rows_per_rank, const uint8_t *owner,
                                                uint32_t rank, uint32_t pixel_bytes, uint32_t frames, void *stream)
try {
    return sdfhip::deinterleave_impl(device, d_share, d_frame, width, height, band_rows, world, rows_per_rank, owner,
                             pixel_bytes, frames, stream, rank);
}
SDFHIP_ABI_CATCH(sdfhip_deinterleave_share_device)

extern "C" uint64_t sdfhip_wire_sparse_bytes(uint32_t width, uint32_t rows, uint32_t capacity)
try {
    return (uint64_t)sparse_layout(width, rows, capacity).bytes;
}
SDFHIP_ABI_CATCH_AS(sdfhip_wire_sparse_bytes, 0)

extern "C" uint64_t sdfhip_wire_sparse_head_offset(uint32_t width, uint32_t rows, uint32_t capacity)
try {
    return (uint64_t)sparse_layout(width, rows, capacity).off_head;
}
SDFHIP_ABI_CATCH_AS(sdfhip_wire_sparse_head_offset, 0)

extern "C" int sdfhip_wire_compact_device(int device, const void *d_wire, void *d_sparse, uint32_t width, uint32_t rows,
                                          uint32_t frames, uint32_t capacity, void *stream)
try {
    if (!d_wire || !d_sparse || width == 0 || rows == 0 || frames == 0)
        return fail(SDFHIP_ERR_ARG, "wire_compact: null or zero argument");
    if (((size_t)rows * width) % 4 != 0) return fail(SDFHIP_ERR_ARG, "wire_compact: rows * width must be a multiple of 4");
    DeviceGuard g(device);
    if (!g.ok) return (void)hipGetLastError(), fail(SDFHIP_ERR_DEVICE, "wire_compact: hipSetDevice(%d) failed", device);
    const SparseLayout L = sparse_layout(width, rows, capacity);
    const dim3 grid((L.tiles + 3) / 4, frames);
    hipLaunchKernelGGL(k_sparse_masks, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t *)d_wire, (uint8_t *)d_sparse, L);
    hipLaunchKernelGGL(k_sparse_scan, dim3(frames), dim3(1024), 0, (hipStream_t)stream, (uint8_t *)d_sparse, L);
    hipLaunchKernelGGL(k_sparse_scatter, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t *)d_wire, (uint8_t *)d_sparse, L);
    HIP_TRY(hipGetLastError());
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_wire_compact_device)

extern "C" int sdfhip_deinterleave_sparse_device(int device, const void *d_gathered, void *d_frame, uint32_t width,
                                                 uint32_t height, uint32_t band_rows, uint32_t world,
                                                 uint32_t rows_per_rank, const uint8_t *owner, uint32_t capacity,
                                                 uint32_t frames, uint32_t *d_overflow, void *stream)
try {
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
    if (!g.ok) return (void)hipGetLastError(), fail(SDFHIP_ERR_DEVICE, "deinterleave_sparse: hipSetDevice(%d) failed", device);
    const SparseLayout L = sparse_layout(width, rows_per_rank, capacity);
    size_t total = (size_t)width * height * frames;
    uint32_t blocks = (uint32_t)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(k_deinterleave_sparse, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)d_gathered,
                       (float4 *)d_frame, width, height, band_rows, world, frames, L, M, d_overflow);
    HIP_TRY(hipGetLastError());
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_deinterleave_sparse_device)

extern "C" int sdfhip_debug_tile_order(sdfhip_scene *s, const uint32_t *d_perm, uint16_t *d_cost)
try {
    if (!s) return fail(SDFHIP_ERR_ARG, "debug_tile_order: null scene");
    std::lock_guard<std::mutex> lk(s->lock);
    s->dbg_tile_perm = d_perm; s->dbg_tile_cost = d_cost;
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_debug_tile_order)

// ---- sdfhip_debug_touch_*: the compulsory bytes of this design ------------------------------------------
// What a frame MUST move is the distinct 128-byte lines of the grid its lookups touch (+ the frame it stores); the counters say
// what it DID move.  The counting builds of the kernels mark every lookup's line in a bitmap per array and XCD.
static void touch_release(sdfhip_scene *s)
{
    for (int a = 0; a < 4; a++) { if (s->touch.bits[a]) (void)hipFree(s->touch.bits[a]); s->touch.bits[a] = nullptr; s->touch.words[a] = 0; }
    if (s->touch.result) (void)hipFree(s->touch.result);
    s->touch.result = nullptr; s->touch.on = false; s->touch.phase = 0;
}

void sdfhip::touch_params(const sdfhip_scene *s, RenderParams &P, int pair)
{
    P.touch_top = s->touch.bits[pair]; P.touch_top_words = s->touch.words[pair];
    P.touch_fine = s->touch.bits[pair + 1]; P.touch_fine_words = s->touch.words[pair + 1];
}

void sdfhip::touch_phase(sdfhip_scene *s, hipStream_t st, int pair)
{
    if (!s->touch.on || s->touch.phase >= sdfhip_scene::Touch::MAX_PHASES) return;
    unsigned long long *out = s->touch.result + (size_t)s->touch.phase * 8;
    const uint32_t most = s->touch.words[pair] > s->touch.words[pair + 1] ? s->touch.words[pair] : s->touch.words[pair + 1];
    const uint32_t blocks = (most + 255u) / 256u < 4096u ? (most + 255u) / 256u : 4096u;
    hipLaunchKernelGGL(k_touch_count, dim3(blocks ? blocks : 1u), dim3(256), 0, st, s->touch.bits[pair], s->touch.words[pair],
                       s->touch.bits[pair + 1], s->touch.words[pair + 1], out);
    const unsigned long long which = (unsigned long long)pair;
    (void)hipMemcpyAsync(out + 4, &which, sizeof which, hipMemcpyHostToDevice, st);     // (pageable source: copied before the call returns)
    s->touch.phase++;
}

extern "C" int sdfhip_debug_touch_begin(sdfhip_scene *s)
try {
    if (!s) return fail(SDFHIP_ERR_ARG, "debug_touch_begin: null scene");
    std::lock_guard<std::mutex> lk(s->lock);
    DeviceGuard g(s->device);
    if (!g.ok) return (void)hipGetLastError(), fail(SDFHIP_ERR_DEVICE, "debug_touch_begin: hipSetDevice(%d) failed", s->device);
    if (!s->d_top || s->top_level + s->fine_bits != (int)s->depth)
        return fail(SDFHIP_ERR_ARG, "debug_touch_begin: the scene has no grid as deep as its tree (the lookups that are counted go through one)");
    HIP_TRY(hipDeviceSynchronize());
    touch_release(s);
    const TopCell *arrays[4] = { s->d_top, s->d_fine, s->d_top2, s->d_fine2 };
    const uint64_t bytes[4] = { (uint64_t)sizeof(TopCell) << (3 * s->top_level), s->fine_bytes,
                                s->d_top2 ? (uint64_t)sizeof(TopCell) << (3 * s->top2_level) : 0,
                                s->d_top2 ? s->top2_bytes - ((uint64_t)sizeof(TopCell) << (3 * s->top2_level)) : 0 };
    for (int a = 0; a < 4; a++) {
        if (!arrays[a] || !bytes[a]) continue;
        if (reinterpret_cast<uintptr_t>(arrays[a]) % 128u) { touch_release(s); return fail(SDFHIP_ERR_DEVICE, "debug_touch_begin: grid array %d is not line-aligned", a); }
        const uint64_t lines = (bytes[a] + 127) / 128, words = (lines + 31) / 32;
        if (words > 0x7FFFFFFFull) { touch_release(s); return fail(SDFHIP_ERR_ARG, "debug_touch_begin: grid array %d is too large", a); }
        s->touch.words[a] = (uint32_t)words;
        hipError_t e = device_alloc((void **)&s->touch.bits[a], (size_t)words * 8 * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemset(s->touch.bits[a], 0, (size_t)words * 8 * sizeof(uint32_t));
        if (e != hipSuccess) { touch_release(s); (void)hipGetLastError(); return fail(SDFHIP_ERR_NOMEM, "debug_touch_begin: %s", hipGetErrorString(e)); }
    }
    const size_t rbytes = (size_t)sdfhip_scene::Touch::MAX_PHASES * 8 * sizeof(unsigned long long);
    hipError_t e = device_alloc((void **)&s->touch.result, rbytes);
    if (e == hipSuccess) e = hipMemset(s->touch.result, 0, rbytes);
    if (e != hipSuccess) { touch_release(s); (void)hipGetLastError(); return fail(SDFHIP_ERR_NOMEM, "debug_touch_begin: %s", hipGetErrorString(e)); }
    s->touch.on = true; s->touch.phase = 0;
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_debug_touch_begin)

extern "C" int sdfhip_debug_touch_end(sdfhip_scene *s, uint64_t *out, uint32_t max_phases, uint32_t *n_phases, uint64_t *array_bytes4)
try {
    if (!s || !out || !n_phases) return fail(SDFHIP_ERR_ARG, "debug_touch_end: null argument");
    std::lock_guard<std::mutex> lk(s->lock);
    DeviceGuard g(s->device);
    if (!g.ok) return (void)hipGetLastError(), fail(SDFHIP_ERR_DEVICE, "debug_touch_end: hipSetDevice(%d) failed", s->device);
    if (!s->touch.on) return fail(SDFHIP_ERR_ARG, "debug_touch_end: sdfhip_debug_touch_begin has not been called on this scene");
    HIP_TRY(hipDeviceSynchronize());
    const uint32_t n = s->touch.phase < max_phases ? s->touch.phase : max_phases;
    if (n) HIP_TRY(hipMemcpy(out, s->touch.result, (size_t)n * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    *n_phases = n;
    if (array_bytes4) {
        array_bytes4[0] = (uint64_t)sizeof(TopCell) << (3 * s->top_level); array_bytes4[1] = s->fine_bytes;
        array_bytes4[2] = s->d_top2 ? (uint64_t)sizeof(TopCell) << (3 * s->top2_level) : 0;
        array_bytes4[3] = s->d_top2 ? s->top2_bytes - array_bytes4[2] : 0;
    }
    touch_release(s);
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_debug_touch_end)

// ---- sdfhip_debug_fail_host_allocations: the exception firewall of the DEVICE half, under injected failures ---------------------
// tests/host_fault_injection.cpp fails operator new under the host half of the library on the CPU.  The entry points that need a
// GPU (upload, render, the multi-device handle, the point-cloud builder) allocate too -- vectors, threads' state, the scene handle
// -- and can only be exercised on the box: the LABORATORY library carries its own operator new (hidden visibility: it replaces the
// allocations made from this library's own code, inlined container code included, and nobody else's), which throws std::bad_alloc
// at the k-th allocation from now when the hook below has armed it.  Every entry point must come back with a status code.
static std::atomic<long long> g_alloc_countdown{-1};      // < 0: off
static std::atomic<unsigned long long> g_alloc_thrown{0};
static void *lab_alloc(size_t n)
{
    long long c = g_alloc_countdown.load(std::memory_order_relaxed);
    while (c >= 0) {
        if (g_alloc_countdown.compare_exchange_weak(c, c - 1)) {
            if (c == 0) { g_alloc_thrown.fetch_add(1); throw std::bad_alloc(); }
            break;
        }
    }
    void *p = malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
// (kept inside this library by its link map, lab.map: <new> declares them with default visibility, and an exported definition would
// neither bind this library's own calls -- the process's first operator new, libstdc++'s, would -- nor stay out of the host's)
void *operator new(size_t n) { return lab_alloc(n); }
void *operator new[](size_t n) { return lab_alloc(n); }
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, size_t) noexcept { free(p); }
void operator delete[](void *p, size_t) noexcept { free(p); }

extern "C" int sdfhip_debug_fail_host_allocations(int64_t countdown, uint64_t *thrown_so_far)
try {
    if (thrown_so_far) *thrown_so_far = g_alloc_thrown.load();
    g_alloc_countdown.store(countdown < 0 ? -1 : (long long)countdown);
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_debug_fail_host_allocations)

extern "C" int sdfhip_debug_step_classes(sdfhip_scene *s, void *stream, uint64_t *out6)
try {
    if (!s || !out6) return fail(SDFHIP_ERR_ARG, "debug_step_classes: null argument");
    std::lock_guard<std::mutex> lk(s->lock);
    DeviceGuard g(s->device);
    if (!g.ok) return (void)hipGetLastError(), fail(SDFHIP_ERR_DEVICE, "debug_step_classes: hipSetDevice(%d) failed", s->device);
    for (int i = 0; i < s->n_scratch; i++)
        if (s->scratch[i].stream == (hipStream_t)stream) {
            HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
            const uint32_t *c = s->scratch[i].ctl + sdfhip_scene::CTL_HIT_WORDS + sdfhip_scene::CTL_QUEUE_WORDS + sdfhip_scene::CTL_PT_WORDS;
            HIP_TRY(hipMemcpy(out6, reinterpret_cast<const unsigned long long *>(c) + 6, 6 * sizeof(uint64_t), hipMemcpyDeviceToHost));
            return SDFHIP_OK;
        }
    return fail(SDFHIP_ERR_ARG, "debug_step_classes: no counting render has run on that stream of this scene");
}
SDFHIP_ABI_CATCH(sdfhip_debug_step_classes)

extern "C" int sdfhip_debug_unorm_table(int device, float *out256)
try {
    if (!out256) return fail(SDFHIP_ERR_ARG, "debug_unorm_table: null argument");
    DeviceGuard g(device);
    if (!g.ok) return (void)hipGetLastError(), fail(SDFHIP_ERR_DEVICE, "debug_unorm_table: hipSetDevice(%d) failed", device);
    float *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 256 * sizeof(float)));
    hipLaunchKernelGGL(k_unorm_table, dim3(1), dim3(256), 0, 0, d);
    hipError_t e = hipMemcpy(out256, d, 256 * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(SDFHIP_ERR_DEVICE, "debug_unorm_table: %s", hipGetErrorString(e));
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_debug_unorm_table)
