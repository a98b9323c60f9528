// libsdfhip.so, device half, host side: the render entry points and the launches behind them.  The ray-march kernels:
// raymarch_kernels.h.
//
// Replaces the reference's dispatch:
//   SdfBox/Program.cs:81,94               UpdateBuffer(info) + DispatchSized  -> sdfhip_render*
//   SdfBox/Program.cs:96-99               display pass draw                   -> sdfhip_render_display
//
// Every entry point builds a RenderCall and ends in render_impl, which validates it, fills the kernel parameters
// (fill_params) and hands the frame to ONE of
//   launch_default    k_march: trees behind a grid as deep as the tree (every validated tree of depth <= 12) -- the product's path
//   launch_path       the path-traced mode: the pipeline of kernels over hit queues (or k_path for trees without such a grid)
//   launch_fallback   k_plain<64> for trees the shader's own traversal must walk (inconsistent links, deeper than 12 levels: CursorG;
//                     no memory for a full grid: CursorS), and k_compact (SDFHIP_FLAG_COMPACT: BASELINE cfg-3's wavefront ray compaction)
//   launch_experiment (experiments build only, lab.hip) the measured-and-dropped A/B forms
#include "raymarch_kernels.h"
#include "tile_order_kernels.h"
#include "scene.h"
#include "abi_guard.h"
#ifdef SDFHIP_EXPERIMENTS
#include "../../include/sdfhip_experimental.h"
#endif

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace sdfhip;

namespace {

// ---- kernel parameters of a call ------------------------------------------------------------------------------------------
struct Plan {
    bool use_stack = false, compact = false, count = false, grid_lookup = false;
    bool persistent = false;          // (experiments build) SDFHIP_FLAG_COMPACT as the persistent-wave kernel on a grid cursor too
    int cur = CUR_GENERIC;
    uint32_t out_mode = 0;
    dim3 grid;
    sdfhip_scene::Scratch *sc = nullptr;
};

void unpack_info(const sdfhip_info *in, FrameInfo &I)
{
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
    I.half_aspect = in->screen_size[0] / in->screen_size[1] * 0.5f;   // Compute.hlsl:165, the shader's own two operations
}

// the sky constant of Compute.hlsl:196 through DisplayFrag.hlsl:24, alpha excluded
uint32_t sky_through_the_display_pass()
{
    auto q = [](float c) { float v = powf(c, 1.0f / 2.2f); v = v > 1.0f ? 1.0f : (v > 0.0f ? v : 0.0f); return (uint32_t)(v * 255.0f + 0.5f); };
    return q(0.005f) | (q(0.01f) << 8) | (q(0.2f) << 16);
}

// Validates the call and fills P and the plan: which cursor, which output, the frame's geometry, cameras and band list.
int fill_params(sdfhip_scene *s, const RenderCall &c, RenderParams &P, Plan &plan)
{
    // `info` points at n_frames consecutive Info blocks (batched launch: the default kernels only)
    if (c.n_frames == 0 || c.n_frames > (uint32_t)MAX_BATCH)
        return fail(SDFHIP_ERR_ARG, "render: n_frames %u outside 1..%d", c.n_frames, MAX_BATCH);
    if (c.width == 0 || c.height == 0 || c.nrows_out == 0 || c.band_rows == 0 || c.band_stride == 0)
        return fail(SDFHIP_ERR_ARG, "render: zero-sized frame or band");
    if ((uint64_t)c.width * c.nrows_out > 0x7FFFFFFFull)
        return fail(SDFHIP_ERR_ARG, "render: %u x %u pixels exceed the 31-bit pixel index", c.width, c.nrows_out);
    const uint32_t flags = c.flags, kind = flags & SDFHIP_KERNEL_MASK;
    if (kind > SDFHIP_KERNEL_STACK) return fail(SDFHIP_ERR_ARG, "render: unknown kernel selector %u", kind);
    if (kind == SDFHIP_KERNEL_STACK && !s->stack_ok)
        return fail(SDFHIP_ERR_ARG, "render: the cursor-stack kernel needs a parent/child-consistent tree of depth <= %d (this scene: depth %u)", MAX_STACK, s->depth);
#ifndef SDFHIP_EXPERIMENTS
    if (flags & ~(uint32_t)(SDFHIP_KERNEL_MASK | SDFHIP_FLAG_COMPACT | SDFHIP_FLAG_COUNT | SDFHIP_FLAG_DISPLAY | SDFHIP_FLAG_DISPLAY_DEBUG | SDFHIP_FLAG_TILE_ORDER))
        return fail(SDFHIP_ERR_ARG, "render: flags %#x are not known to this library (the A/B knobs of include/sdfhip_experimental.h exist in libsdfhip_lab.so only)", flags);
#endif
    plan.use_stack = kind == SDFHIP_KERNEL_STACK || (kind == SDFHIP_KERNEL_AUTO && s->stack_ok);
    plan.compact = (flags & SDFHIP_FLAG_COMPACT) != 0;
#ifdef SDFHIP_EXPERIMENTS
    plan.persistent = plan.compact && (flags & SDFHIP_TUNE_PERSISTENT_WAVES) != 0;
#endif
    plan.count = (flags & SDFHIP_FLAG_COUNT) != 0;
    plan.out_mode = c.sparse ? (uint32_t)OUT_SPARSE : (flags & SDFHIP_FLAG_DISPLAY_DEBUG) ? (uint32_t)OUT_HEAT8 : (flags & SDFHIP_FLAG_DISPLAY) ? (uint32_t)OUT_GAMMA8 : (uint32_t)OUT_RGBA32F;
    if (c.sparse && (plan.compact || c.pt || (flags & (SDFHIP_FLAG_DISPLAY | SDFHIP_FLAG_DISPLAY_DEBUG))))
        return fail(SDFHIP_ERR_ARG, "render_sparse: sparse shares come from the default kernel only (no display pass, path tracing or compaction)");
    if (c.pt) {
        if (c.pt->spp == 0 || c.pt->spp > 4096 || c.pt->max_bounces > 64)
            return fail(SDFHIP_ERR_ARG, "render_path: spp %u (1..4096) or max_bounces %u (0..64) out of range", c.pt->spp, c.pt->max_bounces);
        if (plan.out_mode != 0 || plan.compact)
            return fail(SDFHIP_ERR_ARG, "render_path: the display pass and compaction are not available in path-traced mode");
    }
    if (c.n_frames > 1 && (plan.compact || c.pt || plan.count))
        return fail(SDFHIP_ERR_ARG, "render_batch: only the default kernels, without counting, render several frames per launch");

    memset(&P, 0, sizeof P);
    P.nodes = s->nodes; P.n_nodes = s->n; P.top = s->d_top; P.top_level = s->top_level; P.fine = s->d_fine; P.fine_bits = s->fine_bits; P.fine_order = 0;
    P.out = reinterpret_cast<float4 *>(c.d_out);
    P.out_host = c.out_host ? 1u : 0u;
    P.width = c.width; P.height = c.height;
    P.band_rows = c.band_rows; P.band_first = c.band_first; P.band_stride = c.band_stride;
    P.band_shift = 32u;
    if ((c.band_rows & (c.band_rows - 1u)) == 0u) { P.band_shift = 0u; while ((1u << P.band_shift) < c.band_rows) P.band_shift++; }
    P.nrows_out = c.nrows_out;
    if (c.bands) {                                      // an explicit band list replaces first/stride
        if (c.n_bands == 0 || c.n_bands > (uint32_t)MAX_BAND_LIST)
            return fail(SDFHIP_ERR_ARG, "render_bands: %u bands outside 1..%d", c.n_bands, MAX_BAND_LIST);
        if ((uint64_t)c.n_bands * c.band_rows > c.nrows_out)
            return fail(SDFHIP_ERR_ARG, "render_bands: %u bands of %u rows do not fit nrows_out = %u", c.n_bands, c.band_rows, c.nrows_out);
        const uint32_t frame_bands = (c.height + c.band_rows - 1) / c.band_rows;
        for (uint32_t i = 0; i < c.n_bands; i++)
            if (c.bands[i] >= frame_bands)
                return fail(SDFHIP_ERR_ARG, "render_bands: band %u of a frame with %u bands", (unsigned)c.bands[i], frame_bands);
        P.n_band_list = c.n_bands;
        if (c.n_bands <= (uint32_t)INLINE_BAND_LIST) {
            for (uint32_t i = 0; i < c.n_bands; i++) P.band_list[i] = c.bands[i];
        } else {
            // a list longer than the kernel arguments hold lives in the stream's scratch, copied there when it changes (a gather's deal
            // is the same frame after frame: compared with the host copy kept beside it)
            sdfhip_scene::Scratch *sc = nullptr;
            int rcs = get_scratch(s, c.st, 0, &sc);
            if (rcs != SDFHIP_OK) return rcs;
            plan.sc = sc;
            if (!sc->band_list) HIP_TRY(device_alloc((void **)&sc->band_list, (size_t)MAX_BAND_LIST * sizeof(uint16_t)));
            if (sc->band_n != c.n_bands || memcmp(sc->band_host, c.bands, (size_t)c.n_bands * sizeof(uint16_t)) != 0) {
                memcpy(sc->band_host, c.bands, (size_t)c.n_bands * sizeof(uint16_t));
                HIP_TRY(hipMemcpyAsync(sc->band_list, sc->band_host, (size_t)c.n_bands * sizeof(uint16_t), hipMemcpyHostToDevice, c.st));
                sc->band_n = c.n_bands;
            }
            P.band_ptr = sc->band_list;
        }
    }
    P.tiles_x = (c.width + 7u) / 8u;
    P.tiles_y = (c.nrows_out + 7u) / 8u;
    P.n_tiles = P.tiles_x * P.tiles_y;
    P.n_frames = c.n_frames;
    for (uint32_t f = 0; f < c.n_frames; f++) unpack_info(c.info + f, P.frames[f]);
    P.out_mode = plan.out_mode;
    P.sparse_cap = c.sparse_cap;
    P.sparse_base = c.sparse_base;
    P.sky8 = sky_through_the_display_pass();
    P.pt_spp = c.pt ? c.pt->spp : 0; P.pt_bounces = c.pt ? c.pt->max_bounces : 0; P.pt_seed = c.pt ? c.pt->seed : 0;
    P.pt_albedo = c.pt ? c.pt->albedo : 0.0f;
    // cursor kind: generic, cursor stack, or a grid as deep as the tree (dense or split)
    plan.cur = !plan.use_stack ? CUR_GENERIC : (s->d_top && s->fine_bits) ? CUR_STACK_SPLIT :
               (s->d_top && (uint32_t)s->top_level >= s->depth) ? CUR_STACK_FULL : CUR_STACK;
    plan.grid_lookup = plan.cur == CUR_STACK_FULL || plan.cur == CUR_STACK_SPLIT;
    // one 8x8 tile per 64-lane workgroup, XCD k renders tile rows k, k + 8, ... (tile_of_block): the grid is padded to 8 * ceil(tiles_y / 8) rows
    plan.grid = dim3(8u * ((P.tiles_y + 7u) / 8u) * P.tiles_x, c.n_frames);
    if (plan.compact && (!plan.grid_lookup || plan.persistent) && (c.width > 65535u || c.nrows_out > 65535u))
        return fail(SDFHIP_ERR_ARG, "render: the persistent-wave compact kernel packs a pixel's x and row into 16 bits each (frame %u x %u)", c.width, c.nrows_out);
    if (c.sparse && !plan.grid_lookup)
        return fail(SDFHIP_ERR_ARG, "render_sparse: this scene has no full-depth grid (trees deeper than 12 levels or with inconsistent links render dense shares)");
    return SDFHIP_OK;
}

int need_scratch(sdfhip_scene *s, const RenderCall &c, Plan &plan)
{
    if (plan.sc) return SDFHIP_OK;
    return get_scratch(s, c.st, 0, &plan.sc);
}

// ---- the product's path: k_march --------------------------------------------------------------------------------------------
template <int CUR, bool COUNT>
void launch_march(uint32_t mode, dim3 flat, hipStream_t st, const RenderParams &P)
{
    const dim3 grid = march_grid(P, flat);
    // the frame goes straight into page-locked host memory (sdfhip_render's direct path: never a counting render): the instantiations
    // that store it with plain stores (frame_store in raymarch_kernels.h)
    if (!COUNT && P.out_host && mode <= (uint32_t)OUT_HEAT8) {
        if (mode == OUT_RGBA32F)     hipLaunchKernelGGL((k_march<CUR, false, OUT_RGBA32F | OUT_HOST, false>), grid, dim3(64), 0, st, P);
        else if (mode == OUT_GAMMA8) hipLaunchKernelGGL((k_march<CUR, false, OUT_GAMMA8 | OUT_HOST, false>), grid, dim3(64), 0, st, P);
        else                         hipLaunchKernelGGL((k_march<CUR, false, OUT_HEAT8 | OUT_HOST, false>), grid, dim3(64), 0, st, P);
        return;
    }
    if (mode == OUT_RGBA32F)     hipLaunchKernelGGL((k_march<CUR, COUNT, OUT_RGBA32F, false>), grid, dim3(64), 0, st, P);
    else if (mode == OUT_GAMMA8) hipLaunchKernelGGL((k_march<CUR, COUNT, OUT_GAMMA8, false>), grid, dim3(64), 0, st, P);
    else if (mode == OUT_HEAT8)  hipLaunchKernelGGL((k_march<CUR, COUNT, OUT_HEAT8, false>), grid, dim3(64), 0, st, P);
    else                         hipLaunchKernelGGL((k_march<CUR, COUNT, OUT_SPARSE, false>), grid, dim3(64), 0, st, P);
}

int launch_default(sdfhip_scene *s, const RenderCall &c, RenderParams &P, Plan &plan)
{
    hipStream_t st = c.st;
    // SDFHIP_FLAG_TILE_ORDER: this frame's tiles in the order made from the last frame of the same geometry on this stream
    // (frames of more than 65 536 tiles -- 4K -- run 16 rounds of workgroups: their tail is short and the order costs locality)
    // A batch (n_frames > 1: the frames of a gather group) is ordered by the costs of its LAST frame in the previous launch.
    const bool ordered = (c.flags & SDFHIP_FLAG_TILE_ORDER) != 0 && P.n_tiles <= 65536u && plan.grid.x / 8u <= 65535u &&
                         plan.grid.x <= 8u * 1024u * (uint32_t)ORDER_SPAN && !P.tile_perm && !P.tile_cost;
    sdfhip_scene::Scratch *sc = nullptr;
    if (ordered) {
        int rcs = need_scratch(s, c, plan);
        if (rcs != SDFHIP_OK) return rcs;
        sc = plan.sc;
        uint32_t sig[8] = { c.width, c.height, c.nrows_out, c.band_rows, c.band_first, c.band_stride, P.n_band_list, 0u };
        for (uint32_t i = 0; i < P.n_band_list; i++) sig[7] = sig[7] * 31u + c.bands[i] + 1u;
        if (P.n_tiles > sc->ord_tiles || plan.grid.x > sc->ord_blocks) {
            HIP_TRY(hipStreamSynchronize(st));
            if (sc->ord_cost) (void)hipFree(sc->ord_cost);
            if (sc->ord_class) (void)hipFree(sc->ord_class);
            if (sc->ord_perm) (void)hipFree(sc->ord_perm);
            sc->ord_cost = nullptr; sc->ord_class = nullptr; sc->ord_perm = nullptr; sc->ord_tiles = sc->ord_blocks = 0; sc->ord_valid = false;
            HIP_TRY(device_alloc((void **)&sc->ord_cost, (size_t)MAX_BATCH * P.n_tiles * sizeof(uint16_t)));     // (every frame of a batch writes its costs)
            HIP_TRY(device_alloc((void **)&sc->ord_class, (size_t)P.n_tiles));
            HIP_TRY(device_alloc((void **)&sc->ord_perm, (size_t)plan.grid.x * sizeof(uint32_t)));
            sc->ord_tiles = P.n_tiles; sc->ord_blocks = plan.grid.x;
        }
        if (memcmp(sig, sc->ord_sig, sizeof sig) != 0) { sc->ord_valid = false; memcpy(sc->ord_sig, sig, sizeof sig); }
        P.tile_perm = sc->ord_valid ? sc->ord_perm : nullptr;
        P.perm_per_label = plan.grid.x / 8u;
        // (a camera at rest: the order in use was made from this very camera block -- the same costs would come out, and the order
        // is not made again below: the waves need not report them.  Two wave-wide reductions and two stores per tile less.)
        const sdfhip_info *last_info = c.info + (c.n_frames - 1u);
        P.tile_cost = (sc->ord_valid && memcmp(last_info, &sc->ord_info, sizeof(sdfhip_info)) == 0) ? nullptr : sc->ord_cost;
    }
    if (plan.cur == CUR_STACK_SPLIT) { if (plan.count) launch_march<CUR_STACK_SPLIT, true>(plan.out_mode, plan.grid, st, P); else launch_march<CUR_STACK_SPLIT, false>(plan.out_mode, plan.grid, st, P); }
    else                             { if (plan.count) launch_march<CUR_STACK_FULL, true>(plan.out_mode, plan.grid, st, P); else launch_march<CUR_STACK_FULL, false>(plan.out_mode, plan.grid, st, P); }
    // the next frame's launch order, behind this frame in its stream -- unless the order in use was made from a frame with this
    // very camera block: the same camera gives the same costs and the same order (a viewer at rest pays for the order once)
    const sdfhip_info *last = c.info + (c.n_frames - 1u);
    if (ordered && !(sc->ord_valid && memcmp(last, &sc->ord_info, sizeof(sdfhip_info)) == 0)) {
        sc->ord_info = *last;
        hipLaunchKernelGGL(k_tile_class, dim3((P.n_tiles + 255u) / 256u), dim3(256), 0, st, sc->ord_cost + (size_t)(c.n_frames - 1u) * P.n_tiles,
                           sc->ord_class, P.tiles_x, P.tiles_y);
        hipLaunchKernelGGL(k_tile_order, dim3(8), dim3(1024), 0, st, sc->ord_class, sc->ord_perm, P.tiles_x, P.tiles_y);
        sc->ord_valid = true;
    }
    return SDFHIP_OK;
}

// ---- the path-traced mode ----------------------------------------------------------------------------------------------------
// as a pipeline: camera segments, one kernel per bounce level, the ordered sum
template <int CUR, bool COUNT>
int launch_pt_pipeline(sdfhip_scene *s, dim3 grid, hipStream_t st, RenderParams &P)
{
    // (a multiple of 64: the bounce waves' pushes then spread evenly over the 64 sub-queues, which is what their capacity assumes)
    const uint32_t resident = ((uint32_t)s->cu_count * 32u + 63u) & ~63u;
    hipError_t e;
    if ((e = hipMemsetAsync(P.pt_ctl, 0, sdfhip_scene::CTL_PT_WORDS * sizeof(uint32_t), st)) != hipSuccess)
        return fail(SDFHIP_ERR_DEVICE, "render_path: hipMemsetAsync failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL((k_pt_primary<CUR, COUNT>), grid, dim3(64), 0, st, P);
#ifdef SDFHIP_EXPERIMENTS
    if (COUNT && s->touch.on) touch_phase(s, st, 0);   // sdfhip_debug_touch_*: the camera segments' lines, counted on their own
    uint32_t pt_sort_from = 0;                          // SDFHIP_PT_SORT_FROM=b: only the levels from b on are ordered
    if (const char *env = getenv("SDFHIP_PT_SORT_FROM")) pt_sort_from = (uint32_t)atoi(env);
#endif
    for (uint32_t b = 0; b <= P.pt_bounces; b++) {
        P.pt_level = b;
        // the queue this level fills was drained by the level before it
        if (b > 0 && (e = hipMemsetAsync(P.pt_ctl + (size_t)((b & 1u) ^ 1u) * HIT_QUEUES * 32, 0, HIT_QUEUES * 32 * sizeof(uint32_t), st)) != hipSuccess)
            return fail(SDFHIP_ERR_DEVICE, "render_path: hipMemsetAsync failed: %s", hipGetErrorString(e));
        RenderParams Pb = P;                            // the level's parameters
        if (s->d_top2) {
            // incoherent rays: the same cells through the split grid (the cursor does not depend on the grid it was filled from)
            Pb.top = s->d_top2; Pb.top_level = s->top2_level; Pb.fine = s->d_fine2; Pb.fine_bits = s->fine2_bits; Pb.fine_order = s->fine2_order;
        }
#ifdef SDFHIP_EXPERIMENTS
        // A/B: the level's entries in the order of (region of the hit, octant of the outgoing direction): see k_pt_key
        if (b < pt_sort_from) Pb.pt_sort_bits = 0;
        if (Pb.pt_sort_bits) {
            const uint32_t nkeys = 8u << (3 * Pb.pt_sort_bits);
            const dim3 sort_grid((uint32_t)s->cu_count * 4u);
            if ((e = hipMemsetAsync(Pb.pt_hist, 0, ((size_t)nkeys + 1) * sizeof(uint32_t), st)) != hipSuccess)
                return fail(SDFHIP_ERR_DEVICE, "render_path: hipMemsetAsync failed: %s", hipGetErrorString(e));
            if (s->d_top2) hipLaunchKernelGGL((k_pt_key<CUR_STACK_SPLIT>), sort_grid, dim3(PT_SORT_THREADS), 0, st, Pb);
            else           hipLaunchKernelGGL((k_pt_key<CUR>), sort_grid, dim3(PT_SORT_THREADS), 0, st, Pb);
            hipLaunchKernelGGL((k_pt_scan<0>), dim3(1), dim3(1024), 0, st, Pb.pt_hist, nkeys);
            hipLaunchKernelGGL((k_pt_scatter<0>), sort_grid, dim3(PT_SORT_THREADS), 0, st, Pb);
        }
#endif
#ifdef SDFHIP_EXPERIMENTS
        if (COUNT && s->touch.on) touch_params(s, Pb, s->d_top2 ? 2 : 0);
#endif
#ifdef SDFHIP_EXPERIMENTS
        // A/B (round 6): the level with lane refill (k_pt_bounce_refill: persistent waves, a lane that has finished its entry takes the next)
        const bool refill = getenv("SDFHIP_PT_REFILL") && atoi(getenv("SDFHIP_PT_REFILL")) != 0;
        if (refill && !Pb.pt_sort_bits) {
            if (s->d_top2) hipLaunchKernelGGL((k_pt_bounce_refill<CUR_STACK_SPLIT, COUNT>), dim3(resident), dim3(64), 0, st, Pb);
            else           hipLaunchKernelGGL((k_pt_bounce_refill<CUR, COUNT>), dim3(resident), dim3(64), 0, st, Pb);
        } else
#endif
        if (s->d_top2) hipLaunchKernelGGL((k_pt_bounce<CUR_STACK_SPLIT, COUNT>), dim3(resident), dim3(64), 0, st, Pb);
        else           hipLaunchKernelGGL((k_pt_bounce<CUR, COUNT>), dim3(resident), dim3(64), 0, st, Pb);
#ifdef SDFHIP_EXPERIMENTS
        if (COUNT && s->touch.on) touch_phase(s, st, s->d_top2 ? 2 : 0);     // ... and every bounce level's
#endif
    }
    const size_t npx = (size_t)P.nrows_out * P.width;
    const uint32_t rb = (uint32_t)((npx + 255) / 256 < 4096 ? (npx + 255) / 256 : 4096);
    hipLaunchKernelGGL((k_pt_resolve<COUNT>), dim3(rb), dim3(256), 0, st, P);
    return SDFHIP_OK;
}

int launch_path(sdfhip_scene *s, const RenderCall &c, RenderParams &P, Plan &plan)
{
    hipStream_t st = c.st;
    const dim3 grid(8u * ((P.tiles_y + 7u) / 8u) * P.tiles_x);
    if (!plan.grid_lookup) {
        // trees the shader's own traversal must walk: a lane is a pixel and walks its samples and bounces (k_path)
        auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, dim3(64), 0, st, P); };
        if (plan.cur == CUR_STACK) { if (plan.count) go(k_path<CUR_STACK, true>); else go(k_path<CUR_STACK, false>); }
        else                       { if (plan.count) go(k_path<CUR_GENERIC, true>); else go(k_path<CUR_GENERIC, false>); }
        return SDFHIP_OK;
    }
    // the pipeline of kernels (k_pt_primary -> k_pt_bounce per level -> k_pt_resolve)
    const size_t npx = (size_t)c.nrows_out * c.width, npaths = npx * c.pt->spp;
    if (npaths >= ((size_t)1 << 32))
        return fail(SDFHIP_ERR_ARG, "render_path: %zu paths (pixels x spp) exceed the 32-bit path index", npaths);
    P.pt_cap = (uint32_t)((((size_t)grid.x + HIT_QUEUES - 1) / HIT_QUEUES) * 64 * c.pt->spp + 8192);
    const size_t qbytes = (size_t)PT_RECORDS * 16 * HIT_QUEUES * P.pt_cap;        // one hit queue
    const size_t ebytes = (size_t)(c.pt->max_bounces + 1) * npaths * 4, tbytes = npaths * 4;
    size_t sort_bytes = 0;
#ifdef SDFHIP_EXPERIMENTS
    // SDFHIP_PT_SORT=R (1..3): the bounce levels take their entries in the order of a key (k_pt_key): an A/B, measured without gain
    uint32_t sort_bits = 0;
    if (const char *env = getenv("SDFHIP_PT_SORT")) { const int v = atoi(env); if (v >= 1 && v <= PT_SORT_MAX_BITS) sort_bits = (uint32_t)v; }
    const size_t n_entries = (size_t)HIT_QUEUES * P.pt_cap;
    sort_bytes = sort_bits ? n_entries * 6 + (((size_t)8 << (3 * PT_SORT_MAX_BITS)) + 64) * 4 : 0;
#endif
    sdfhip_scene::Scratch *sc = nullptr;
    int rcs = get_pt_scratch(s, st, 2 * qbytes + ebytes + 2 * tbytes + sort_bytes, &sc);
    if (rcs != SDFHIP_OK) return rcs;
    plan.sc = sc;
#ifdef SDFHIP_EXPERIMENTS
    if (sort_bits) {
        char *base = sc->pt_buf + 2 * qbytes + ebytes + 2 * tbytes;
        P.pt_perm = reinterpret_cast<uint32_t *>(base);
        P.pt_hist = reinterpret_cast<uint32_t *>(base + n_entries * 4);
        P.pt_key = reinterpret_cast<uint16_t *>(base + n_entries * 4 + (((size_t)8 << (3 * PT_SORT_MAX_BITS)) + 64) * 4);
        P.pt_sort_bits = sort_bits;
        if (const char *env = getenv("SDFHIP_PT_SORT_XCD")) P.pt_sort_xcd = atoi(env) != 0 ? 1u : 0u;
    }
#endif
    P.pt_q[0] = reinterpret_cast<float4 *>(sc->pt_buf);
    P.pt_q[1] = reinterpret_cast<float4 *>(sc->pt_buf + qbytes);
    P.pt_e = reinterpret_cast<float *>(sc->pt_buf + 2 * qbytes);
    P.pt_t = reinterpret_cast<float *>(sc->pt_buf + 2 * qbytes + ebytes);
    P.pt_n = reinterpret_cast<uint32_t *>(sc->pt_buf + 2 * qbytes + ebytes + tbytes);
    P.pt_ctl = sc->ctl + sdfhip_scene::CTL_HIT_WORDS + sdfhip_scene::CTL_QUEUE_WORDS;
    if (plan.cur == CUR_STACK_SPLIT) return plan.count ? launch_pt_pipeline<CUR_STACK_SPLIT, true>(s, grid, st, P) : launch_pt_pipeline<CUR_STACK_SPLIT, false>(s, grid, st, P);
    return plan.count ? launch_pt_pipeline<CUR_STACK_FULL, true>(s, grid, st, P) : launch_pt_pipeline<CUR_STACK_FULL, false>(s, grid, st, P);
}

// ---- wavefront ray compaction (SDFHIP_FLAG_COMPACT) on the default kernel: k_march<..., QUEUE> -> k_shadow -----------------------
template <int CUR, bool COUNT>
void launch_queued(uint32_t mode, dim3 grid, dim3 shade_grid, hipStream_t st, const RenderParams &P)
{
    auto go = [&](auto march, auto shade) {
        hipLaunchKernelGGL(march, march_grid(P, grid), dim3(64), 0, st, P);
        hipLaunchKernelGGL(shade, shade_grid, dim3(64), 0, st, P);
    };
    if (mode == OUT_RGBA32F)     go(k_march<CUR, COUNT, OUT_RGBA32F, true>, k_shadow<CUR, COUNT, OUT_RGBA32F>);
    else if (mode == OUT_GAMMA8) go(k_march<CUR, COUNT, OUT_GAMMA8, true>, k_shadow<CUR, COUNT, OUT_GAMMA8>);
    else                         go(k_march<CUR, COUNT, OUT_HEAT8, true>, k_shadow<CUR, COUNT, OUT_HEAT8>);
}

int launch_compact(sdfhip_scene *s, const RenderCall &c, RenderParams &P, Plan &plan)
{
    dim3 shade_grid;
    int rc = prepare_shadow_queue(s, c, P, plan.grid, COMPACT_MIN_LANES, &plan.sc, &shade_grid);
    if (rc != SDFHIP_OK) return rc;
    const uint32_t mode = plan.out_mode;
    if (plan.cur == CUR_STACK_SPLIT) { if (plan.count) launch_queued<CUR_STACK_SPLIT, true>(mode, plan.grid, shade_grid, c.st, P); else launch_queued<CUR_STACK_SPLIT, false>(mode, plan.grid, shade_grid, c.st, P); }
    else                             { if (plan.count) launch_queued<CUR_STACK_FULL, true>(mode, plan.grid, shade_grid, c.st, P); else launch_queued<CUR_STACK_FULL, false>(mode, plan.grid, shade_grid, c.st, P); }
    return SDFHIP_OK;
}

// ---- trees without a full-depth grid (and their form of the compaction: persistent waves with lane refill) ---------------------
template <int CUR, bool COUNT>
void launch_plain_or_compact(bool compact, dim3 grid, hipStream_t st, const RenderParams &P)
{
    if (compact) hipLaunchKernelGGL((k_compact<CUR, COUNT>), grid, dim3(64), 0, st, P);
    else         hipLaunchKernelGGL((k_plain<CUR, COUNT, 64>), grid, dim3(64), 0, st, P);
}

int launch_fallback(sdfhip_scene *s, const RenderCall &c, RenderParams &P, Plan &plan)
{
    hipStream_t st = c.st;
    dim3 grid = plan.grid;
    if (plan.compact) {
        // persistent waves: one wave per workgroup, 32 waves per CU, tiles from 8 atomic queues (this stream's scratch)
        const uint32_t blocks = (uint32_t)s->cu_count * 32u;
        grid = dim3(blocks < P.n_tiles ? blocks : (P.n_tiles ? P.n_tiles : 1));
        int rcs = need_scratch(s, c, plan);
        if (rcs != SDFHIP_OK) return rcs;
        P.queue = plan.sc->ctl + sdfhip_scene::CTL_HIT_WORDS;
        HIP_TRY(hipMemsetAsync(P.queue, 0, sdfhip_scene::CTL_QUEUE_WORDS * sizeof(uint32_t), st));
    }
    const bool k = plan.count, cp = plan.compact;
#ifdef SDFHIP_EXPERIMENTS
    // (A/B, SDFHIP_TUNE_PERSISTENT_WAVES: the persistent-wave form on a grid cursor, which carried SDFHIP_FLAG_COMPACT until round 4)
    if (plan.cur == CUR_STACK_SPLIT)     { if (k) launch_plain_or_compact<CUR_STACK_SPLIT, true>(cp, grid, st, P); else launch_plain_or_compact<CUR_STACK_SPLIT, false>(cp, grid, st, P); }
    else if (plan.cur == CUR_STACK_FULL) { if (k) launch_plain_or_compact<CUR_STACK_FULL, true>(cp, grid, st, P); else launch_plain_or_compact<CUR_STACK_FULL, false>(cp, grid, st, P); }
    else
#else
    if (plan.grid_lookup) return fail(SDFHIP_ERR_ARG, "render: no fallback kernel for a scene behind a full-depth grid");
#endif
    if (plan.cur == CUR_STACK)           { if (k) launch_plain_or_compact<CUR_STACK, true>(cp, grid, st, P); else launch_plain_or_compact<CUR_STACK, false>(cp, grid, st, P); }
    else                                 { if (k) launch_plain_or_compact<CUR_GENERIC, true>(cp, grid, st, P); else launch_plain_or_compact<CUR_GENERIC, false>(cp, grid, st, P); }
    return SDFHIP_OK;
}

}  // namespace

// the shadow-ray queue of a k_march<..., QUEUE> / k_shadow launch pair on the stream's scratch
int sdfhip::prepare_shadow_queue(sdfhip_scene *s, const RenderCall &c, RenderParams &P, dim3 grid, uint32_t hit_min, sdfhip_scene::Scratch **scp,
                                 dim3 *shade_grid)
{
    // a queue takes the rays of every HIT_QUEUES-th workgroup, and a workgroup queues fewer than hit_min of its 64 pixels
    const uint32_t per_wave = hit_min > 64u ? 64u : hit_min - 1u;
    P.hit_min = hit_min;
    P.hit_cap = ((grid.x + HIT_QUEUES - 1u) / HIT_QUEUES) * (per_wave ? per_wave : 1u);
    const size_t records = (size_t)c.n_frames * HIT_QUEUES * P.hit_cap;
    sdfhip_scene::Scratch *sc = nullptr;
    int rcs = get_scratch(s, c.st, records, &sc);
    if (rcs != SDFHIP_OK) return rcs;
    *scp = sc;
    P.hit_a = reinterpret_cast<float4 *>(sc->hit_buf);
    P.hit_b = reinterpret_cast<int4 *>(sc->hit_buf + sc->records * 16);
    P.hit_c = reinterpret_cast<uint4 *>(sc->hit_buf + sc->records * 32);
    P.hit_d = reinterpret_cast<float4 *>(sc->hit_buf + sc->records * 48);
    P.hit_ctl = sc->ctl;
    P.hit_set = sc->launches++ & 1u;
    // every queued ray is marched by a resident wave: at most one chunk of 64 per k_march workgroup
    const uint32_t resident = (uint32_t)s->cu_count * 32u;
    *shade_grid = dim3(grid.x < resident ? grid.x : resident, c.n_frames);
    return SDFHIP_OK;
}

// ---- statistics of a call, collected outside the handle's lock --------------------------------------------------------------
int sdfhip::take_ticket(sdfhip_scene *s, sdfhip_scene::StatsTicket **out)
{
    for (int i = 0; i < sdfhip_scene::MAX_TICKETS; i++) {
        sdfhip_scene::StatsTicket &t = s->tickets[i];
        if (t.in_use.load(std::memory_order_acquire)) continue;
        // (each piece by itself: a creation that failed half-way is completed by the next call, not taken for done)
        if (!t.ev0) HIP_TRY(hipEventCreate(&t.ev0));
        if (!t.ev1) HIP_TRY(hipEventCreate(&t.ev1));
        if (!t.h_counters) HIP_TRY(hipHostMalloc((void **)&t.h_counters, 6 * sizeof(unsigned long long), hipHostMallocDefault));
        t.counted = false; t.kernel_used = 0;
        t.in_use.store(true, std::memory_order_release);
        *out = &t;
        return SDFHIP_OK;
    }
    return fail(SDFHIP_ERR_ARG, "render: %d calls with statistics are in flight on one scene handle (at most %d at a time)", sdfhip_scene::MAX_TICKETS, sdfhip_scene::MAX_TICKETS);
}

int sdfhip::finish_stats(sdfhip_scene *s, sdfhip_scene::StatsTicket *t, sdfhip_stats *stats)
{
    DeviceGuard g(s->device);
    hipError_t e = hipEventSynchronize(t->ev1);
    memset(stats, 0, sizeof *stats);
    if (e == hipSuccess) e = hipEventElapsedTime(&stats->kernel_ms, t->ev0, t->ev1);
    stats->kernel_used = t->kernel_used;
    if (e == hipSuccess && t->counted) {
        const unsigned long long *h = t->h_counters;
        stats->n_nodes = h[0]; stats->n_samples = h[1]; stats->n_steps = h[2]; stats->n_shadow_rays = h[3];
        stats->n_loads = h[4]; stats->n_hits = h[5];
    }
    t->in_use.store(false, std::memory_order_release);
    if (e != hipSuccess) return fail(SDFHIP_ERR_DEVICE, "render: waiting for the statistics failed: %s", hipGetErrorString(e));
    return SDFHIP_OK;
}

int sdfhip::render_impl(sdfhip_scene *s, const RenderCall &c, sdfhip_scene::StatsTicket *ticket)
{
    RenderParams P;
    Plan plan;
    int rc = fill_params(s, c, P, plan);
    if (rc != SDFHIP_OK) return rc;
    hipStream_t st = c.st;
    if (plan.count) {                                          // this stream's own counters (its scratch)
        rc = need_scratch(s, c, plan);
        if (rc != SDFHIP_OK) return rc;
        P.counters = reinterpret_cast<unsigned long long *>(plan.sc->ctl + sdfhip_scene::CTL_HIT_WORDS + sdfhip_scene::CTL_QUEUE_WORDS + sdfhip_scene::CTL_PT_WORDS);
        HIP_TRY(hipMemsetAsync(P.counters, 0, sdfhip_scene::CTL_COUNTER_WORDS * sizeof(uint32_t), st));
    }
    if (c.pt && plan.grid_lookup) ensure_scatter_grid(s);     // (before the clock)
#ifdef SDFHIP_EXPERIMENTS
    if (s->touch.on && plan.count) touch_params(s, P, 0);      // sdfhip_debug_touch_*: this render's lookups mark the lines they touch
#endif
    if (ticket) HIP_TRY(hipEventRecord(ticket->ev0, st));
    bool launched = false;
#ifdef SDFHIP_EXPERIMENTS
    rc = launch_experiment(s, c, P, plan.cur, plan.count, plan.grid, &launched, &plan.sc);
    if (rc != SDFHIP_OK) return rc;
#endif
    if (!launched) {
        if (c.pt) rc = launch_path(s, c, P, plan);
        else if (plan.grid_lookup && !plan.compact) rc = launch_default(s, c, P, plan);
        else if (plan.grid_lookup && !plan.persistent) rc = launch_compact(s, c, P, plan);
        else rc = launch_fallback(s, c, P, plan);
        if (rc != SDFHIP_OK) return rc;
    }
    HIP_TRY(hipGetLastError());
#ifdef SDFHIP_EXPERIMENTS
    if (s->touch.on && plan.count && !(c.pt && plan.grid_lookup)) touch_phase(s, st, 0);    // (the path-traced pipeline closes a phase per kernel)
#endif
    if (ticket) {
        ticket->kernel_used = (plan.use_stack ? SDFHIP_KERNEL_STACK : SDFHIP_KERNEL_GENERIC) | (plan.compact ? SDFHIP_FLAG_COMPACT : 0u);
        if (plan.count) {
            HIP_TRY(hipMemcpyAsync(ticket->h_counters, P.counters, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            ticket->counted = true;
        }
        HIP_TRY(hipEventRecord(ticket->ev1, st));
    }
    if (plan.sc) HIP_TRY(hipEventRecord(plan.sc->idle, st));   // this stream's scratch is busy until here
    return SDFHIP_OK;
}

// ---- entry points: frames that stay in HBM -----------------------------------------------------------------------------------
namespace {
// A statistics ticket taken for a call goes back on EVERY way out that does not reach finish_stats (which frees it itself):
// a failed copy or synchronisation between the launch and the wait must not use one of the handle's tickets up for good.
struct TicketGuard {
    sdfhip_scene::StatsTicket *t = nullptr;
    ~TicketGuard() { if (t) t->in_use.store(false, std::memory_order_release); }
    sdfhip_scene::StatsTicket *release() { sdfhip_scene::StatsTicket *r = t; t = nullptr; return r; }
};
// One device-resident call: the launch under the handle's lock, the wait for its statistics (if asked) outside it.
int render_resident(sdfhip_scene *s, const RenderCall &c, sdfhip_stats *stats, const char *what)
{
    TicketGuard tg;
    {
        std::lock_guard<std::mutex> lk(s->lock);
        DeviceGuard g(s->device);
        if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "%s: hipSetDevice(%d) failed", what, s->device);
        if (stats) { int rt = take_ticket(s, &tg.t); if (rt != SDFHIP_OK) return rt; }
        const int rc = render_impl(s, c, tg.t);
        if (rc != SDFHIP_OK) return rc;
    }
    return stats ? finish_stats(s, tg.release(), stats) : SDFHIP_OK;
}
}  // namespace

extern "C" int sdfhip_render_device(sdfhip_scene *s, const sdfhip_info *info, uint32_t width,
                                    uint32_t height, uint32_t band_rows, uint32_t band_first,
                                    uint32_t band_stride, uint32_t nrows_out, uint32_t flags,
                                    float *d_rgba_out, void *stream, sdfhip_stats *stats)
try {
    if (!s || !info || !d_rgba_out) return fail(SDFHIP_ERR_ARG, "render_device: null argument");
    RenderCall c;
    c.info = info; c.width = width; c.height = height; c.band_rows = band_rows; c.band_first = band_first; c.band_stride = band_stride;
    c.nrows_out = nrows_out; c.flags = flags; c.d_out = d_rgba_out;
    c.st = (hipStream_t)stream;             // NULL = the HIP default stream, as everywhere in HIP
    return render_resident(s, c, stats, "render_device");
}
SDFHIP_ABI_CATCH(sdfhip_render_device)

extern "C" int sdfhip_render_batch_device(sdfhip_scene *s, const sdfhip_info *infos, uint32_t n_frames,
                                          uint32_t width, uint32_t height, uint32_t band_rows,
                                          uint32_t band_first, uint32_t band_stride, uint32_t nrows_out,
                                          uint32_t flags, float *d_rgba_out, void *stream, sdfhip_stats *stats)
try {
    if (!s || !infos || !d_rgba_out) return fail(SDFHIP_ERR_ARG, "render_batch_device: null argument");
    RenderCall c;
    c.info = infos; c.n_frames = n_frames; c.width = width; c.height = height; c.band_rows = band_rows; c.band_first = band_first;
    c.band_stride = band_stride; c.nrows_out = nrows_out; c.flags = flags; c.d_out = d_rgba_out; c.st = (hipStream_t)stream;
    return render_resident(s, c, stats, "render_batch_device");
}
SDFHIP_ABI_CATCH(sdfhip_render_batch_device)

extern "C" int sdfhip_render_path_device(sdfhip_scene *s, const sdfhip_info *info, const sdfhip_pathtrace *pt,
                                         uint32_t width, uint32_t height, uint32_t band_rows,
                                         uint32_t band_first, uint32_t band_stride, uint32_t nrows_out,
                                         uint32_t flags, float *d_rgba_out, void *stream, sdfhip_stats *stats)
try {
    if (!s || !info || !pt || !d_rgba_out) return fail(SDFHIP_ERR_ARG, "render_path_device: null argument");
    RenderCall c;
    c.info = info; c.pt = pt; c.width = width; c.height = height; c.band_rows = band_rows; c.band_first = band_first;
    c.band_stride = band_stride; c.nrows_out = nrows_out; c.flags = flags; c.d_out = d_rgba_out; c.st = (hipStream_t)stream;
    return render_resident(s, c, stats, "render_path_device");
}
SDFHIP_ABI_CATCH(sdfhip_render_path_device)

extern "C" int sdfhip_render_bands_device(sdfhip_scene *s, const sdfhip_info *infos, uint32_t n_frames,
                                          const sdfhip_pathtrace *pt, uint32_t width, uint32_t height,
                                          uint32_t band_rows, const uint16_t *bands, uint32_t n_bands,
                                          uint32_t nrows_out, uint32_t flags, float *d_rgba_out, void *stream,
                                          sdfhip_stats *stats)
try {
    if (!s || !infos || !bands || !d_rgba_out) return fail(SDFHIP_ERR_ARG, "render_bands_device: null argument");
    if (pt && n_frames != 1) return fail(SDFHIP_ERR_ARG, "render_bands_device: the path-traced mode renders one frame per launch");
    RenderCall c;
    c.info = infos; c.n_frames = n_frames; c.pt = pt; c.width = width; c.height = height; c.band_rows = band_rows;
    c.bands = bands; c.n_bands = n_bands; c.nrows_out = nrows_out; c.flags = flags; c.d_out = d_rgba_out; c.st = (hipStream_t)stream;
    return render_resident(s, c, stats, "render_bands_device");
}
SDFHIP_ABI_CATCH(sdfhip_render_bands_device)

extern "C" int sdfhip_render_sparse_device(sdfhip_scene *s, const sdfhip_info *infos, uint32_t n_frames, uint32_t width,
                                           uint32_t height, uint32_t band_rows, const uint16_t *bands, uint32_t n_bands,
                                           uint32_t nrows_out, uint32_t capacity, uint32_t count_base, uint32_t flags, void *d_share, void *stream)
try {
    if (!s || !infos || !bands || !d_share) return fail(SDFHIP_ERR_ARG, "render_sparse_device: null argument");
    if (capacity == 0) return fail(SDFHIP_ERR_ARG, "render_sparse_device: capacity 0");
    RenderCall c;
    c.info = infos; c.n_frames = n_frames; c.width = width; c.height = height; c.band_rows = band_rows; c.bands = bands; c.n_bands = n_bands;
    c.nrows_out = nrows_out; c.flags = flags; c.d_out = reinterpret_cast<float *>(d_share); c.st = (hipStream_t)stream;
    c.sparse = true; c.sparse_cap = capacity; c.sparse_base = count_base;
    return render_resident(s, c, nullptr, "render_sparse_device");
}
SDFHIP_ABI_CATCH(sdfhip_render_sparse_device)

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
try {
    if (!out || bytes == 0) return fail(SDFHIP_ERR_ARG, "host_alloc: null or zero argument");
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, (size_t)bytes, hipHostMallocPortable | hipHostMallocMapped);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(SDFHIP_ERR_NOMEM, "host_alloc: hipHostMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e)); }
    int rc = host_range_add(p, (size_t)bytes, true, "host_alloc");
    if (rc != SDFHIP_OK) return rc;
    *out = p;
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_host_alloc)
extern "C" int sdfhip_host_register(void *p, uint64_t bytes)
try {
    if (!p || bytes == 0) return fail(SDFHIP_ERR_ARG, "host_register: null or zero argument");
    hipError_t e = hipHostRegister(p, (size_t)bytes, hipHostRegisterPortable | hipHostRegisterMapped);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(SDFHIP_ERR_DEVICE, "host_register: hipHostRegister(%llu bytes) failed: %s", (unsigned long long)bytes, hipGetErrorString(e)); }
    return host_range_add(p, (size_t)bytes, false, "host_register");
}
SDFHIP_ABI_CATCH(sdfhip_host_register)
extern "C" int sdfhip_host_release(void *p)
try {
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
SDFHIP_ABI_CATCH(sdfhip_host_release)

extern "C" int sdfhip_render(sdfhip_scene *s, const sdfhip_info *info, uint32_t width,
                             uint32_t height, uint32_t flags, float *rgba_out, sdfhip_stats *stats)
try {
    if (!s || !info || !rgba_out) return fail(SDFHIP_ERR_ARG, "render: null argument");
    if (width == 0 || height == 0) return fail(SDFHIP_ERR_ARG, "render: zero-sized frame");
#ifdef SDFHIP_EXPERIMENTS
    if (flags & SDFHIP_FLAG_WIRE) return fail(SDFHIP_ERR_ARG, "render: SDFHIP_FLAG_WIRE is for the device-resident entry points");
#endif
    std::lock_guard<std::mutex> lk(s->lock);
    DeviceGuard g(s->device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "render: hipSetDevice(%d) failed", s->device);
    auto t0 = std::chrono::steady_clock::now();
    RenderCall whole;                                  // the whole frame in one launch on the scene's own stream
    whole.info = info; whole.width = width; whole.height = height; whole.band_rows = height; whole.nrows_out = height; whole.flags = flags;
    whole.st = s->stream;
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
    const char *hb = lab_env("SDFHIP_HOST_BANDS");
    // (any flag beyond the output mode -- counting, compaction, the experiments build's A/B knobs -- takes the plain form)
    const bool viewer = !stats && !(flags & ~(uint32_t)(SDFHIP_KERNEL_MASK | SDFHIP_FLAG_DISPLAY | SDFHIP_FLAG_DISPLAY_DEBUG | SDFHIP_FLAG_TILE_ORDER)) &&
                        !(hb && atoi(hb) == 0);
    uint32_t nb = 1;
    if (viewer) {
        // A page-locked destination (sdfhip_host_alloc / _register): the march can store its pixels into the host's array
        // itself -- no device frame, no copy: the stores cross PCIe while the other waves march.  Measured (scripts/host_frame.py
        // --locked, profiles/r03_host_frame.txt in the history, commit 53ee955): into the library's own allocation 1080p 0.676 ms against 0.711-0.731 with band
        // copies into the same memory (RGBA8: 0.260 against 0.292); at 4K the copy engine's 55 GB/s beat the stores' 51 (2.47
        // against 2.58 ms), so frames of 4 M pixels and more go in bands.  Into the caller's own registered array (4 KB pages
        // wherever they happened to lie) the stores are slower -- 1080p RGBA32F 0.758 against 0.714 -- and are used for frames of
        // less than 16 MB only (1080p RGBA8: 0.270 against 0.293).
        bool ours = false;
        const size_t frame_bytes = (size_t)width * height * px_bytes;
        void *const known = host_range_device_pointer(rgba_out, frame_bytes, &ours), *const direct = hb ? nullptr : known;
        const bool locked = known != nullptr;
        if (direct && (ours ? (size_t)width * height < ((size_t)4 << 20) : frame_bytes < ((size_t)16 << 20))) {
            RenderCall c = whole;
            c.flags = flags | SDFHIP_FLAG_TILE_ORDER; c.d_out = reinterpret_cast<float *>(direct); c.out_host = true;
            int rc = render_impl(s, c, nullptr);
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
        HIP_TRY(device_alloc((void **)&s->d_frame, need * sizeof(float4)));
        s->frame_cap = need;
    }
    if (!banded) {
        RenderCall c = whole;
        c.d_out = reinterpret_cast<float *>(s->d_frame);
        TicketGuard tg;
        if (stats) { int rt = take_ticket(s, &tg.t); if (rt != SDFHIP_OK) return rt; }
        int rc = render_impl(s, c, tg.t);
        if (rc != SDFHIP_OK) return rc;
        const hipError_t e = hipMemcpyAsync(rgba_out, s->d_frame, (size_t)width * height * px_bytes, hipMemcpyDeviceToHost, s->stream);
        const hipError_t e2 = hipStreamSynchronize(s->stream);
        if (stats) {
            rc = finish_stats(s, tg.release(), stats);
            stats->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rc != SDFHIP_OK) return rc;
        }
        if (e != hipSuccess || e2 != hipSuccess) return fail(SDFHIP_ERR_DEVICE, "render: copying the frame to the host failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
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
        RenderCall c = whole;
        c.band_rows = rows; c.band_first = b; c.band_stride = nb; c.nrows_out = rows; c.flags = flags | SDFHIP_FLAG_TILE_ORDER;
        c.d_out = reinterpret_cast<float *>(d_base + (size_t)b * rows * width * px_bytes); c.st = s->band_stream[b];
        int rc = render_impl(s, c, nullptr);
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
SDFHIP_ABI_CATCH(sdfhip_render)

extern "C" int sdfhip_render_path(sdfhip_scene *s, const sdfhip_info *info, const sdfhip_pathtrace *pt,
                                  uint32_t width, uint32_t height, uint32_t flags, float *rgba_out,
                                  sdfhip_stats *stats)
try {
    if (!s || !info || !pt || !rgba_out) return fail(SDFHIP_ERR_ARG, "render_path: null argument");
    if (width == 0 || height == 0) return fail(SDFHIP_ERR_ARG, "render_path: zero-sized frame");
    std::lock_guard<std::mutex> lk(s->lock);
    DeviceGuard g(s->device);
    if (!g.ok) return fail(SDFHIP_ERR_DEVICE, "render_path: hipSetDevice(%d) failed", s->device);
    auto t0 = std::chrono::steady_clock::now();
    size_t need = (size_t)width * height;
    if (need > s->frame_cap) {
        if (s->d_frame) { (void)hipFree(s->d_frame); s->d_frame = nullptr; s->frame_cap = 0; }
        HIP_TRY(device_alloc((void **)&s->d_frame, need * sizeof(float4)));
        s->frame_cap = need;
    }
    RenderCall c;
    c.info = info; c.pt = pt; c.width = width; c.height = height; c.band_rows = height; c.nrows_out = height; c.flags = flags;
    c.d_out = reinterpret_cast<float *>(s->d_frame); c.st = s->stream;
    TicketGuard tg;
    if (stats) { int rt = take_ticket(s, &tg.t); if (rt != SDFHIP_OK) return rt; }
    int rc = render_impl(s, c, tg.t);
    if (rc != SDFHIP_OK) return rc;
    HIP_TRY(hipMemcpyAsync(rgba_out, s->d_frame, need * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
    uint32_t overflow = 0;                              // a hit that found no room in its queue (see pt_push): never silently
    for (int i = 0; i < s->n_scratch; i++)
        if (s->scratch[i].stream == s->stream && s->scratch[i].pt_buf)
            HIP_TRY(hipMemcpyAsync(&overflow, s->scratch[i].ctl + sdfhip_scene::CTL_HIT_WORDS + sdfhip_scene::CTL_QUEUE_WORDS + (size_t)2 * HIT_QUEUES * 32,
                                   sizeof overflow, hipMemcpyDeviceToHost, s->stream));
    const hipError_t es = hipStreamSynchronize(s->stream);
    if (stats) {
        rc = finish_stats(s, tg.release(), stats);
        stats->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rc != SDFHIP_OK) return rc;
    }
    if (es != hipSuccess) return fail(SDFHIP_ERR_DEVICE, "render_path: %s", hipGetErrorString(es));
    if (overflow) return fail(SDFHIP_ERR_NOMEM, "render_path: a hit queue of the path-traced pipeline overflowed; the frame is incomplete");
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_render_path)

extern "C" int sdfhip_render_display(sdfhip_scene *s, const sdfhip_info *info, uint32_t width,
                                     uint32_t height, uint32_t flags, int debug, uint8_t *rgba8_out,
                                     sdfhip_stats *stats)
try {
    flags = (flags & ~(uint32_t)(SDFHIP_FLAG_DISPLAY | SDFHIP_FLAG_DISPLAY_DEBUG)) |
            (debug ? SDFHIP_FLAG_DISPLAY_DEBUG : SDFHIP_FLAG_DISPLAY);
    return sdfhip_render(s, info, width, height, flags, reinterpret_cast<float *>(rgba8_out), stats);
}
SDFHIP_ABI_CATCH(sdfhip_render_display)
