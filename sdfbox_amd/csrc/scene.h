// The scene handle of libsdfhip.so and what the host-side translation units share: scene.hip (upload, grids, per-stream
// scratch), render.hip (launches, the render entry points), gather.hip (rank 0's side of the tile gather) and, in the
// experiments build, lab.hip.
#pragma once
#include "raymarch_device.h"
#include "sdfhip_internal.h"

#include <atomic>
#include <mutex>

struct sdfhip_scene {
    int device = 0;
    uint32_t n = 0, depth = 0;
    int stack_ok = 0;
    void *alloc = nullptr;            // hipMalloc'ed block holding the records
    sdfhip::NodeRec *nodes = nullptr; // = alloc + 112: node 1 (first sibling block) starts a 128-B line
    hipStream_t stream = nullptr;
    sdfhip::TopCell *d_top = nullptr; // top grid of the cursor-stack kernels (raymarch_device.h), or null
    int top_level = 0;
    sdfhip::TopCell *d_fine = nullptr;// split grid: blocks of fine cells below the internal cells of d_top, or null
    int fine_bits = 0;
    uint64_t fine_bytes = 0;
    // A second, split grid beside the scene's own, for the kernels whose rays are incoherent (the bounce levels of
    // the path-traced pipeline are HBM-bound: a ray that stays near the surface stays inside one block of fine cells).
    // Same cells, same cursor; built by sdfhip_scene_prepare_path or on the first path-traced render.
    sdfhip::TopCell *d_top2 = nullptr, *d_fine2 = nullptr;
    int top2_level = 0, fine2_bits = 0, fine2_order = 0, scatter_tried = 0;
    int opt_scatter_grid = -1, opt_scatter_order = -1;   // sdfhip_upload_options (-1: choose)
    uint64_t top2_bytes = 0;
    size_t total_mem = 0;
    uint32_t *d_verdict = nullptr;    // k_validate's two words (upload)
    // Per-stream scratch of the render launches: the queues and per-path results of the path-traced pipeline, the tile-queue
    // heads of the compact kernel, the counters of SDFHIP_FLAG_COUNT renders, the launch order of SDFHIP_FLAG_TILE_ORDER.
    // Launches on one stream run in order and may share a scratch; launches on different streams overlap (frames in flight)
    // and must not, so every stream that renders on this handle gets its own.
    struct Scratch {
        hipStream_t stream = nullptr;
        char *hit_buf = nullptr;      // SDFHIP_FLAG_COMPACT: the shadow-ray queue between k_march and k_shadow, `records` x 64 bytes
        size_t records = 0;
        uint32_t *ctl = nullptr;      // control words, see CTL_* below
        uint32_t launches = 0;        // k_march / k_shadow launch pairs so far: its parity selects the set of fill counts
        uint64_t last_use = 0;        // the handle's render count when this scratch was last handed out (the oldest idle one is recycled)
        hipEvent_t idle = nullptr;    // the library's own event behind the last launch that used this scratch: "is it idle?" never asks
                                      // the caller's stream handle, which may have been destroyed since
        char *pt_buf = nullptr;       // path-traced pipeline: two hit queues, then the per-path results
        size_t pt_bytes = 0;
        uint16_t *band_list = nullptr;// a band list longer than the kernel arguments hold (INLINE_BAND_LIST): MAX_BAND_LIST entries in
        uint32_t band_n = 0;          // device memory, and the host's copy of what they hold
        uint16_t band_host[sdfhip::MAX_BAND_LIST];
        // SDFHIP_FLAG_TILE_ORDER: the wave-iterations of every tile of the last frame rendered on this stream, the launch order
        // made from them for the next one, and the frame geometry both belong to
        uint16_t *ord_cost = nullptr;
        uint8_t *ord_class = nullptr;
        uint32_t *ord_perm = nullptr;
        uint32_t ord_tiles = 0, ord_blocks = 0;      // capacity of the two arrays
        uint32_t ord_sig[8] = { 0 };                 // width, height, nrows_out, band_rows, band_first, band_stride, n_band_list, hash of the list
        bool ord_valid = false;
        sdfhip_info ord_info;                        // the camera block of the frame the order was made from
    };
    static constexpr int MAX_SCRATCH = 16;
    // ctl: [the shadow-ray queues' fill counts, two sets] [the compact kernel's 8 tile queues] [the path-traced pipeline's fill counts
    // + its overflow word's line] [the counters of SDFHIP_FLAG_COUNT renders on this stream, 16 x u64: nodes, samples, steps,
    // shadow rays, loads, hits, and from [6] the step classes of sdfhip_debug_step_classes]
    static constexpr size_t CTL_HIT_WORDS = (size_t)2 * sdfhip::MAX_BATCH * sdfhip::HIT_QUEUES * 32, CTL_QUEUE_WORDS = 8 * 32,
                            CTL_PT_WORDS = (size_t)2 * sdfhip::HIT_QUEUES * 32 + 32, CTL_COUNTER_WORDS = 32;
    static constexpr size_t CTL_BYTES = (CTL_HIT_WORDS + CTL_QUEUE_WORDS + CTL_PT_WORDS + CTL_COUNTER_WORDS) * sizeof(uint32_t);
    Scratch scratch[MAX_SCRATCH];
    int n_scratch = 0;
    uint64_t uses = 0;
    // Statistics of a render (sdfhip_stats): an event pair and a pinned landing place for the counters per call in flight, so
    // that the wait for them happens OUTSIDE the handle's lock -- two threads that ask for statistics on two streams of one
    // handle wait side by side, not one behind the other.
    struct StatsTicket {
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        unsigned long long *h_counters = nullptr;    // pinned, 6 x u64
        std::atomic<bool> in_use{false};
        bool counted = false;
        uint32_t kernel_used = 0;
    };
    static constexpr int MAX_TICKETS = 8;
    StatsTicket tickets[MAX_TICKETS];
    float4 *d_frame = nullptr;        // grown on demand by sdfhip_render
    size_t frame_cap = 0;
    // sdfhip_render (the host frame): the frame in HOST_BANDS row bands, each on its own stream (its own scratch: the launch
    // order of SDFHIP_FLAG_TILE_ORDER is kept per stream) behind the band before it; a band's copy to the host runs while
    // the next bands march
    static constexpr int HOST_BANDS = 4;
    hipStream_t band_stream[HOST_BANDS] = { nullptr, nullptr, nullptr, nullptr };
    hipEvent_t band_done[HOST_BANDS] = { nullptr, nullptr, nullptr, nullptr };
    int cu_count = 0;
#ifdef SDFHIP_EXPERIMENTS
    uint32_t *d_d4 = nullptr;         // the grid's second form (CursorFF, lab_device.h): one word per cell of the deepest level
    uint4 *d_recs = nullptr;          // + the sample records of the non-flat leaves; or null
    uint64_t d4_bytes = 0;
    const uint32_t *dbg_tile_perm = nullptr;   // sdfhip_debug_tile_order: experiment hooks for k_march
    uint16_t *dbg_tile_cost = nullptr;
    // sdfhip_debug_touch_begin / _end (lab.hip): the distinct 128-byte lines of the grid arrays that counting renders touch.
    // bits[a]: [8 XCDs][words[a]] bitmaps of array a (0 top, 1 fine, 2 top2, 3 fine2: the bounce levels' grid); a "phase" is a
    // kernel launch whose lines are counted on their own (a whole frame of the default kernel; the camera segments and every bounce
    // level of the path-traced pipeline): result[phase][0..1] = lines of (top, fine) or (top2, fine2) chip-wide, [2..3] = the
    // same summed over the XCDs, [4] = which pair (0 / 2)
    struct Touch {
        static constexpr uint32_t MAX_PHASES = 16;
        uint32_t *bits[4] = { nullptr, nullptr, nullptr, nullptr };
        uint32_t words[4] = { 0, 0, 0, 0 };
        unsigned long long *result = nullptr;      // device, [MAX_PHASES][8]
        uint32_t phase = 0;
        bool on = false;
    } touch;
#endif
    std::mutex lock;                  // render on one handle is single-caller; this makes misuse safe
};

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            (void)hipGetLastError();   /* the runtime's record of it: a later launch check must not report it as its own */ \
            return sdfhip::fail(SDFHIP_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));  \
        }                                                                                   \
    } while (0)

namespace sdfhip {

// Keeps the caller's current device intact (the host process may be PyTorch).
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
        ok = hipSetDevice(dev) == hipSuccess;
        // (a failed runtime call leaves its error for the next hipGetLastError(): a later launch check -- "k_validate launch failed:
        // invalid device ordinal" -- would report THIS failure as its own.  The caller reports it; the runtime's record is cleared.)
        if (!ok) (void)hipGetLastError();
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

constexpr int MAX_TOP_LEVEL = 8;

// scene.hip
// The scratch of stream `st` on this scene (created on the stream's first render), with room for `records` 64-byte hit
// records (0: control words only)
int get_scratch(sdfhip_scene *s, hipStream_t st, size_t records, sdfhip_scene::Scratch **out);
// ... with `bytes` of path-traced pipeline buffers
int get_pt_scratch(sdfhip_scene *s, hipStream_t st, size_t bytes, sdfhip_scene::Scratch **out);
// the bounce levels' second grid, built once (sdfhip_scene_prepare_path, or in front of the first path-traced render)
void ensure_scatter_grid(sdfhip_scene *s);
// A split grid over the scene's records: dense cells of level C whose internal cells (level word 15) name, in `children`,
// the first cell of a block of 8^FB fine cells; built on s->stream.  false (nothing allocated) when memory or the byte limit say no.
bool build_split_grid(sdfhip_scene *s, int C, int FB, int order, uint64_t max_fine_bytes, TopCell **coarse_out, TopCell **fine_out,
                      uint64_t *fine_bytes_out);

// render.hip: every render entry point ends here.  ticket != null: the call's statistics are collected into it (finish_stats,
// after the handle's lock has been released)
struct RenderCall {
    const sdfhip_info *info = nullptr;
    uint32_t width = 0, height = 0, band_rows = 0, band_first = 0, band_stride = 1, nrows_out = 0, flags = 0;
    float *d_out = nullptr;
    hipStream_t st = nullptr;
    const sdfhip_pathtrace *pt = nullptr;
    uint32_t n_frames = 1;
    const uint16_t *bands = nullptr;
    uint32_t n_bands = 0;
    bool sparse = false;
    uint32_t sparse_cap = 0, sparse_base = 0;
    bool out_host = false;            // d_out is page-locked host memory: the kernels store the frame with plain stores (frame_store)
};
int render_impl(sdfhip_scene *s, const RenderCall &call, sdfhip_scene::StatsTicket *ticket);
// k_march's launch grid: the workgroups of a frame's flat numbering (x: 8 * ceil(tiles_y / 8) * tiles_x, y: frames of the batch) as
// (8 tiles_x, frames, ceil(tiles_y / 8)), so that the kernel reads tile column, tile row and XCD label off its block coordinates
// -- or, in tile order (P.tile_perm: one entry per workgroup of a frame's flat numbering), as (8 XCD labels, frames, order slots)
inline dim3 march_grid(const RenderParams &P, dim3 flat)
{
    return P.tile_perm ? dim3(8u, flat.y, flat.x / 8u) : dim3(8u * P.tiles_x, flat.y, (P.tiles_y + 7u) / 8u);
}
// SDFHIP_FLAG_COMPACT on a scene with a full-depth grid: the shadow-ray queue of k_march<..., QUEUE> / k_shadow on the stream's scratch
// (waves holding fewer than hit_min shadow rays queue them); fills P.hit_* and the second kernel's grid
constexpr uint32_t COMPACT_MIN_LANES = 32;
int prepare_shadow_queue(sdfhip_scene *s, const RenderCall &call, RenderParams &P, dim3 grid, uint32_t hit_min, sdfhip_scene::Scratch **sc,
                         dim3 *shade_grid);
int take_ticket(sdfhip_scene *s, sdfhip_scene::StatsTicket **out);           // under the handle's lock
int finish_stats(sdfhip_scene *s, sdfhip_scene::StatsTicket *t, sdfhip_stats *stats);   // outside it

// gather.hip
int deinterleave_impl(int device, const void *d_gathered, void *d_frame, uint32_t width, uint32_t height, uint32_t band_rows, uint32_t world,
                      uint32_t rows_per_rank, const uint8_t *owner, uint32_t pixel_bytes, uint32_t frames, void *stream,
                      uint32_t only_rank = 0xFFFFFFFFu);

#ifdef SDFHIP_EXPERIMENTS
// lab.hip
void build_dense4(sdfhip_scene *s);
// the A/B forms of a render (flags SDFHIP_TUNE_*, the tile-order and workgroup-size knobs, wire pixels, the grid's second form):
// launched = true when one of them took the frame
int launch_experiment(sdfhip_scene *s, const RenderCall &call, RenderParams &P, int cur, bool count, dim3 grid, bool *launched,
                      sdfhip_scene::Scratch **sc);
// sdfhip_debug_touch_*: point P's lookups at the bitmaps of grid pair `pair` (0: the scene's own grid, 2: the bounce levels'), and
// close a phase behind the launches issued so far on `st` (count the pair's lines into result[phase], clear its bitmaps)
void touch_params(const sdfhip_scene *s, RenderParams &P, int pair);
void touch_phase(sdfhip_scene *s, hipStream_t st, int pair);
#endif

}  // namespace sdfhip
