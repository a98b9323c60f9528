// libsdfhip.so, the frame sharded over the GPUs of one node, behind ONE call (new design: the reference
// is single-GPU).
//
// Replaces: what Program.Draw does for the compute pass (SdfBox/Program.cs:79-110: UpdateBuffer(info) :81 +
// DispatchSized(W, H, 1) :94) when the frame is rendered by several devices: the host makes the same one call per frame
// (sdfhip_multi_render) or keeps groups of frames in flight (sdfhip_multi_submit / _wait).
//
// One process, one host thread per device (each issues its own device's work: a group costs ~40 us of API calls per device,
// which one thread would pay eight times over), one HIP stream per device and slot:
//   every rank r   the scene is replicated at create; rank r renders its row bands of the group's frames in ONE launch of the
//                  default kernel, which writes the sparse wire share itself (OUT_SPARSE, raymarch_kernels.h): no dense
//                  intermediate, no compaction kernels; the share's slot count goes to pinned host memory behind it;
//   gather         rank r > 0 pushes its share into rank 0's memory over its own xGMI link: the share's fixed part and as many
//                  packed floats as its last shares needed (x 1.25) -- two peer copies (hipMemcpyPeerAsync) on the rank's
//                  stream, or, with SDFHIP_MULTI_TRANSPORT=rccl, ncclSend / ncclRecv pairs inside ncclGroupStart/End (RCCL is
//                  loaded with dlopen on first use, so the library has no link-time dependency on it);
//   rank 0         waits (on the device) for the ranks' events, expands all shares into the frames in row order
//                  (k_deinterleave_sparse2; rank 0's own share is read where it was rendered) and, for sdfhip_multi_render,
//                  copies the frame to the host.
// A share that needed more floats than were sent is found when the slot is waited for (the counts are in pinned memory by
// then): the missing tail is copied and that rank's rows are expanded again -- inside the library, the caller sees a complete
// frame either way.  The path-traced mode gathers dense RGBA32F bands (its pixels are not wire pixels).
//
// Everything here is written against the library's own C ABI (sdfhip_render_sparse_device, sdfhip_deinterleave_*): the
// multi-device layer adds no kernel.
#include "abi_guard.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <new>
#include <thread>
#include <unistd.h>
#include <vector>

using namespace sdfhip;

namespace {

constexpr uint32_t MAX_RANKS = 16;
constexpr uint32_t MAX_SLOTS = 4;
constexpr uint32_t MAX_GROUP = 8;          // frames per launch (sdfhip_render_batch_device's limit)

#define M_TRY(expr)                                                                           \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                             \
            (void)hipGetLastError();   /* the runtime's record of it: a later launch check must not report it as its own */ \
            return fail(SDFHIP_ERR_DEVICE, "multi: %s failed: %s", #expr, hipGetErrorString(e_)); \
        }                                                                                   \
    } while (0)

// ---- RCCL, loaded on demand ---------------------------------------------------------------------------------------
struct Rccl {
    void *handle = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommAbort)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load()
    {
        if (handle) return true;
        const char *names[] = { getenv("SDFHIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char *n : names) {
            if (!n || !*n) continue;
            handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) return false;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(handle, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(handle, "ncclCommDestroy"));
        CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(handle, "ncclCommAbort"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(handle, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(handle, "ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(dlsym(handle, "ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(dlsym(handle, "ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(handle, "ncclGetErrorString"));
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv;
    }
};
constexpr int NCCL_UINT8 = 1;              // ncclUint8 (nccl.h: ncclInt8 = 0, ncclUint8 = 1)

// ---- one worker thread per rank > 0: runs the jobs the caller posts, one at a time -------------------------------------
// A frame is tens of microseconds of device work per rank, and a thread that sleeps on a condition variable takes about
// as long to wake: a worker polls for the next job for SPIN_US after its last one (a viewer's frames come back to back)
// before it goes to sleep, and the caller polls for the workers' completion the same way.
constexpr int SPIN_US = 200;
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    int (*fn)(void *, uint32_t) = nullptr;     // job: fn(arg, rank)
    void *arg = nullptr;
    uint32_t rank = 0;
    std::atomic<uint32_t> posted{0}, finished{0};
    std::atomic<bool> quit{false};
    int rc = SDFHIP_OK;
    char err[256] = { 0 };                     // the job's sdfhip_last_error text (thread-local there)
    static bool spin(const std::atomic<uint32_t> &a, uint32_t seen, const std::atomic<bool> *stop)
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            for (int i = 0; i < 64; i++) {
                if (a.load(std::memory_order_acquire) != seen || (stop && stop->load(std::memory_order_acquire))) return true;
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#else
                std::this_thread::yield();
#endif
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(SPIN_US)) return false;
        }
    }
    // The thread's body.  Nothing may leave it as an exception (that would be std::terminate in the host's process, the crash the
    // C ABI promises not to cause: abi_guard.h): a job that throws is a job that failed, and a lock or a wait that throws
    // (std::system_error) fails the job in hand and is tried again.
    void run() noexcept
    {
        uint32_t seen = 0;
        for (;;) {
            try {
                if (!spin(posted, seen, &quit)) {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return posted.load(std::memory_order_acquire) != seen || quit.load(); });
                }
                if (quit.load()) return;
                seen = posted.load(std::memory_order_acquire);
                int r;
                try { r = fn(arg, rank); } catch (...) { r = abi_caught("sdfhip_multi: a rank's job"); }
                rc = r;
                if (r != SDFHIP_OK) { strncpy(err, sdfhip_last_error(), sizeof err - 1); err[sizeof err - 1] = 0; }
                { std::lock_guard<std::mutex> lk(mu); finished.store(seen, std::memory_order_release); }
                cv.notify_all();
            } catch (...) {
                rc = abi_caught("sdfhip_multi: a rank's worker thread");
                strncpy(err, sdfhip_last_error(), sizeof err - 1); err[sizeof err - 1] = 0;
                seen = posted.load(std::memory_order_acquire);
                finished.store(seen, std::memory_order_release);
                if (quit.load()) return;
                usleep(1000);
            }
        }
    }
    void post(int (*f)(void *, uint32_t), void *a)
    {
        { std::lock_guard<std::mutex> lk(mu); fn = f; arg = a; posted.fetch_add(1, std::memory_order_release); }
        cv.notify_all();
    }
    int join()
    {
        const uint32_t want = posted.load(std::memory_order_acquire);
        for (;;) {
            if (finished.load(std::memory_order_acquire) == want) return rc;
            if (!spin(finished, want - 1, nullptr)) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return finished.load(std::memory_order_acquire) == want; });
            }
        }
    }
};

struct RankBuf {
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_sent = nullptr;
    uint8_t *d_share = nullptr;            // on the rank's device: the sparse share it renders (its header word 0 is a counter that runs on
                                           // from launch to launch: nothing else may ever be written into this buffer)
    uint8_t *d_bands = nullptr;            // on the rank's device: dense bands (path-traced mode, scenes without a full-depth grid)
    uint8_t *d_gather = nullptr;           // on rank 0's device: where rank r > 0's share lands (sparse mode)
    size_t share_cap = 0, bands_cap = 0, gather_cap = 0;  // bytes allocated
    uint32_t count_base = 0;               // the share's counter before this slot's launch (it is never zeroed, see sdfhip_render_sparse_device)
    uint32_t sent = 0;                     // floats copied with this slot's share
};

struct Slot {
    RankBuf rb[MAX_RANKS];
    hipStream_t rx_stream = nullptr;       // rank 0's device: RCCL receives
    hipEvent_t ev_rx = nullptr, ev_done = nullptr, ev_t0 = nullptr;
    uint8_t *d_frames = nullptr;           // rank 0's device: the assembled frames (when the caller gave no buffer)
    size_t frames_cap = 0;
    uint8_t *d_dense = nullptr;            // rank 0's device: [world] dense shares (path-traced mode, scenes without a full-depth grid)
    size_t dense_cap = 0;
    uint32_t dense_px = 16;                // bytes per pixel of this submission's dense shares (4 with the display pass)
    uint32_t *h_counts = nullptr;          // pinned, [MAX_RANKS]: the shares' counters as the assembly kernel found them
    uint32_t last_used[MAX_RANKS] = { 0 };
    void *out = nullptr;                   // where this submission's frames go (the caller's buffer or d_frames)
    bool timed = false;                    // this submission recorded the ranks' start events
    bool busy = false, path = false, dirty = false;   // path: dense shares (see d_dense); dirty: a submission failed half-way: the shares' counters are re-zeroed before the next one
    uint32_t n_frames = 0, width = 0, height = 0, flags = 0;
    std::chrono::steady_clock::time_point t_submit;
};

struct Layout {
    uint32_t width = 0, height = 0, band_rows = 0, n_bands = 0, rows_per_rank = 0, world = 0;
    float weight = 0.0f;
    std::vector<uint8_t> owner;
    std::vector<uint16_t> bands[MAX_RANKS];
};

// sdfbox_amd/tiles.py BandLayout: round robin, or -- rank 0 also assembles the frame -- bands dealt by largest remaining
// credit with rank 0 weighing `weight` of a peer (integer credits: every caller computes the same deal)
void deal_bands(Layout &L, uint32_t width, uint32_t height, uint32_t world, uint32_t band_rows, float weight)
{
    L.width = width; L.height = height; L.world = world; L.band_rows = band_rows; L.weight = weight;
    L.n_bands = (height + band_rows - 1) / band_rows;
    L.owner.assign(L.n_bands, 0);
    for (uint32_t r = 0; r < MAX_RANKS; r++) L.bands[r].clear();
    if (world > 1 && weight < 1.0f) {
        const long long unit = 1 << 20;
        long long w[MAX_RANKS], credit[MAX_RANKS] = { 0 }, total = 0;
        for (uint32_t r = 0; r < world; r++) { w[r] = r == 0 ? (long long)(weight * (float)unit + 0.5f) : unit; total += w[r]; }
        for (uint32_t b = 0; b < L.n_bands; b++) {
            uint32_t best = 0;
            for (uint32_t r = 0; r < world; r++) { credit[r] += w[r]; if (credit[r] > credit[best]) best = r; }
            credit[best] -= total;
            L.owner[b] = (uint8_t)best;
        }
    } else {
        for (uint32_t b = 0; b < L.n_bands; b++) L.owner[b] = (uint8_t)(b % world);
    }
    uint32_t most = 0;
    for (uint32_t b = 0; b < L.n_bands; b++) L.bands[L.owner[b]].push_back((uint16_t)b);
    for (uint32_t r = 0; r < world; r++) most = (uint32_t)L.bands[r].size() > most ? (uint32_t)L.bands[r].size() : most;
    L.rows_per_rank = most * band_rows;
}

}  // namespace

struct sdfhip_multi {
    uint32_t n = 0;
    int devices[MAX_RANKS];
    sdfhip_scene *scenes[MAX_RANKS];
    Worker *workers[MAX_RANKS];            // [0] unused: the caller's thread plays rank 0
    Slot slots[MAX_SLOTS];
    Layout lay;
    uint32_t band_rows = 16;
    float rank0_weight = 1.0f;
    uint32_t est[MAX_RANKS];               // floats to send with a rank's next share (0: not measured yet, send all)
    bool use_rccl = false, rccl_self = false;
    bool dense_only = false;               // a device's scene has no full-depth grid (depth > 12, inconsistent links, no memory for the grid):
                                           // the default kernel cannot write sparse shares there, so every frame gathers dense bands
    bool broken = false;                   // RCCL transport: a submission failed between its receives and its sends; only sdfhip_multi_free is left
    Rccl rccl;
    void *comms[MAX_RANKS];
    uint64_t resends = 0;
    std::mutex lock;
    // the submission in progress (what the workers' jobs read)
    struct Job {
        sdfhip_multi *m; uint32_t slot; const sdfhip_info *infos; uint32_t n_frames, width, height, flags, capacity;
        const sdfhip_pathtrace *pt;
        bool dense;                         // dense bands instead of sparse shares (pt, or dense_only)
        bool timed;                         // per-rank start events (sdfhip_multi_stats.rank_ms): one API call per rank that a frame alone can do without
        // upload
        const int32_t *structs; const uint8_t *values; uint32_t n_nodes;
    } job;
};

namespace {

struct DevGuard {
    int prev = -1;
    explicit DevGuard(int d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (hipSetDevice(d) != hipSuccess) (void)hipGetLastError(); }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int nccl_fail(sdfhip_multi *m, int rc, const char *what)
{
    return fail(SDFHIP_ERR_DEVICE, "multi: %s failed: %s", what, m->rccl.GetErrorString ? m->rccl.GetErrorString(rc) : "RCCL error");
}

// bytes of rank r's share that are always sent, and the floats behind them
struct ShareShape { size_t bytes, fixed, off_floats; uint32_t capacity; };
ShareShape share_shape(const sdfhip_multi *m, uint32_t n_frames)
{
    ShareShape s;
    const uint64_t cap = (uint64_t)m->lay.rows_per_rank * m->lay.width * n_frames;
    s.capacity = (uint32_t)(cap > 0x7FFFFFFFull ? 0x7FFFFFFFull : cap);       // every pixel lit: nothing is ever dropped
    s.bytes = (size_t)sdfhip_sparse2_bytes(m->lay.width, m->lay.rows_per_rank, n_frames, s.capacity);
    s.off_floats = (size_t)sdfhip_sparse2_floats_offset(m->lay.width, m->lay.rows_per_rank, n_frames);
    s.fixed = s.off_floats;
    return s;
}

// rank r's memory -> rank 0's memory on the sender's stream (several ranks may share a device when the pipeline is rehearsed)
hipError_t push(void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t st)
{
    if (dst_dev == src_dev) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
    return hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, st);
}

int grow(uint8_t **p, size_t *cap, size_t need, hipStream_t drain)
{
    if (need <= *cap) return SDFHIP_OK;
    if (drain) M_TRY(hipStreamSynchronize(drain));
    if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
    M_TRY(device_alloc((void **)p, need));
    *cap = need;
    return SDFHIP_OK;
}

// ---- what rank r does for one submission: render its bands, report the count, push the share to rank 0 -------------------
int rank_submit(void *arg, uint32_t r)
{
    sdfhip_multi *m = static_cast<sdfhip_multi *>(arg);
    const sdfhip_multi::Job &J = m->job;
    Slot &S = m->slots[J.slot];
    RankBuf &B = S.rb[r];
    DevGuard g(m->devices[r]);
    const Layout &L = m->lay;
    if (L.bands[r].empty()) { B.sent = 0; return SDFHIP_OK; }              // more ranks than bands: nothing to do
    if (J.timed) M_TRY(hipEventRecord(B.ev_start, B.stream));
    if (J.dense) {
        // dense bands (RGBA32F, or RGBA8 through the display pass); rank 0 renders straight into its place in the gathered array
        const size_t share = (size_t)J.n_frames * L.rows_per_rank * L.width * S.dense_px;
        uint8_t *dst = S.d_dense + (size_t)r * share;
        uint8_t *out = r == 0 ? dst : B.d_bands;
        int rc = sdfhip_render_bands_device(m->scenes[r], J.infos, J.n_frames, J.pt, L.width, L.height, L.band_rows, L.bands[r].data(),
                                            (uint32_t)L.bands[r].size(), L.rows_per_rank, J.flags, reinterpret_cast<float *>(out),
                                            B.stream, nullptr);
        if (rc != SDFHIP_OK) return rc;
        if (r > 0) {
            if (m->use_rccl) {
                int e = m->rccl.Send(out, share, NCCL_UINT8, 0, m->comms[r], B.stream);
                if (e) return nccl_fail(m, e, "ncclSend");
            } else {
                M_TRY(push(dst, m->devices[0], out, m->devices[r], share, B.stream));
            }
        }
        M_TRY(hipEventRecord(B.ev_sent, B.stream));
        return SDFHIP_OK;
    }
    const ShareShape sh = share_shape(m, J.n_frames);
    int rc = sdfhip_render_sparse_device(m->scenes[r], J.infos, J.n_frames, L.width, L.height, L.band_rows, L.bands[r].data(),
                                         (uint32_t)L.bands[r].size(), L.rows_per_rank, sh.capacity, B.count_base, J.flags, B.d_share, B.stream);
    if (rc != SDFHIP_OK) return rc;
    if (r > 0 || m->rccl_self) {
        // ONE copy: the share's fixed part and, right behind it, as many floats as this rank's last shares needed
        const uint32_t nf = m->est[r] ? (m->est[r] < sh.capacity ? m->est[r] : sh.capacity) : sh.capacity;
        const size_t bytes = sh.fixed + (size_t)nf * 4;
        B.sent = nf;
        if (m->use_rccl) {
            int e = m->rccl.GroupStart();
            if (e) return nccl_fail(m, e, "ncclGroupStart");
            if ((e = m->rccl.Send(B.d_share, bytes, NCCL_UINT8, 0, m->comms[r], B.stream)) != 0) return nccl_fail(m, e, "ncclSend");
            if (r == 0 && (e = m->rccl.Recv(B.d_gather, bytes, NCCL_UINT8, 0, m->comms[0], B.stream)) != 0) return nccl_fail(m, e, "ncclRecv");   // (self test: same group)
            if ((e = m->rccl.GroupEnd()) != 0) return nccl_fail(m, e, "ncclGroupEnd");
        } else {
            M_TRY(push(B.d_gather, m->devices[0], B.d_share, m->devices[r], bytes, B.stream));
        }
    } else {
        B.sent = sh.capacity;
    }
    M_TRY(hipEventRecord(B.ev_sent, B.stream));
    return SDFHIP_OK;
}

int rank_upload(void *arg, uint32_t r)
{
    sdfhip_multi *m = static_cast<sdfhip_multi *>(arg);
    const sdfhip_multi::Job &J = m->job;
    return sdfhip_scene_upload(m->devices[r], J.structs, J.values, J.n_nodes, &m->scenes[r]);
}

// run fn for every rank: ranks > 0 on their threads, rank 0 on the caller's
int on_all_ranks(sdfhip_multi *m, int (*fn)(void *, uint32_t))
{
    for (uint32_t r = 1; r < m->n; r++) m->workers[r]->post(fn, m);
    int rc = fn(m, 0);
    char first[256] = { 0 };
    if (rc != SDFHIP_OK) { strncpy(first, sdfhip_last_error(), sizeof first - 1); }
    for (uint32_t r = 1; r < m->n; r++) {
        const int rr = m->workers[r]->join();
        if (rr != SDFHIP_OK && rc == SDFHIP_OK) { rc = rr; strncpy(first, m->workers[r]->err, sizeof first - 1); }
    }
    return rc == SDFHIP_OK ? SDFHIP_OK : fail(rc, "%s", first);
}

// buffers of one slot for the current layout; everything that may reallocate drains the stream it belongs to first
int prepare_slot(sdfhip_multi *m, Slot &S, uint32_t n_frames, bool path, bool internal_frames, size_t frame_px_bytes)
{
    const Layout &L = m->lay;
    for (uint32_t r = 0; r < m->n; r++) {
        RankBuf &B = S.rb[r];
        DevGuard g(m->devices[r]);
        if (path) {
            if (r > 0) { int rc = grow(&B.d_bands, &B.bands_cap, (size_t)n_frames * L.rows_per_rank * L.width * frame_px_bytes, B.stream); if (rc) return rc; }
        } else {
            const ShareShape sh = share_shape(m, n_frames);
            const uint8_t *before = B.d_share;
            int rc = grow(&B.d_share, &B.share_cap, sh.bytes, B.stream);
            if (rc) return rc;
            if (B.d_share != before || S.dirty) {                       // a new buffer: its counter starts at zero (and is never zeroed again)
                M_TRY(hipMemsetAsync(B.d_share, 0, 64, B.stream));
                B.count_base = 0;
            }
            if (r > 0 || m->rccl_self) {
                DevGuard g0(m->devices[0]);
                rc = grow(&B.d_gather, &B.gather_cap, sh.bytes, S.rb[0].stream);
                if (rc) return rc;
            }
        }
    }
    DevGuard g0(m->devices[0]);
    if (path) { int rc = grow(&S.d_dense, &S.dense_cap, (size_t)m->n * n_frames * L.rows_per_rank * L.width * frame_px_bytes, S.rb[0].stream); if (rc) return rc; }
    if (internal_frames) { int rc = grow(&S.d_frames, &S.frames_cap, (size_t)n_frames * L.width * L.height * frame_px_bytes, S.rb[0].stream); if (rc) return rc; }
    S.dirty = false;
    return SDFHIP_OK;
}

// rank 0, after every rank has issued its work: wait for the shares on the device, expand them into the frames
int assemble(sdfhip_multi *m, Slot &S, int only_rank)
{
    const Layout &L = m->lay;
    DevGuard g0(m->devices[0]);
    hipStream_t st = S.rb[0].stream;
    for (uint32_t r = 1; r < m->n; r++)
        if (!L.bands[r].empty() && (only_rank < 0 || (uint32_t)only_rank == r)) M_TRY(hipStreamWaitEvent(st, S.rb[r].ev_sent, 0));
    if (m->use_rccl && !(m->rccl_self && m->n == 1) && only_rank < 0) M_TRY(hipStreamWaitEvent(st, S.ev_rx, 0));
    const uint8_t *owner = (L.weight < 1.0f && m->n > 1) ? L.owner.data() : nullptr;
    if (S.path)
        return sdfhip_deinterleave_bands_device(m->devices[0], S.d_dense, S.out, L.width, L.height, L.band_rows, m->n, L.rows_per_rank,
                                                L.owner.data(), S.dense_px, S.n_frames, st);
    const void *shares[MAX_RANKS];
    for (uint32_t r = 0; r < m->n; r++) shares[r] = (r == 0 && !m->rccl_self) ? S.rb[0].d_share : S.rb[r].d_gather;
    const ShareShape sh = share_shape(m, S.n_frames);
    return sdfhip_deinterleave_sparse2_device(m->devices[0], shares, S.out, L.width, L.height, L.band_rows, m->n, L.rows_per_rank, owner,
                                              sh.capacity, S.n_frames, S.flags & (SDFHIP_FLAG_DISPLAY | SDFHIP_FLAG_DISPLAY_DEBUG), only_rank,
                                              only_rank < 0 ? S.h_counts : nullptr, st);
}

// rank 0's RCCL receives of one submission (peer copies need none: the senders write into rank 0's memory)
int post_receives(sdfhip_multi *m, Slot &S)
{
    if (!m->use_rccl || m->n == 1) return SDFHIP_OK;
    const Layout &L = m->lay;
    DevGuard g0(m->devices[0]);
    int e = m->rccl.GroupStart();
    if (e) return nccl_fail(m, e, "ncclGroupStart");
    for (uint32_t r = 1; r < m->n; r++) {
        if (L.bands[r].empty()) continue;
        if (S.path) {
            const size_t share = (size_t)S.n_frames * L.rows_per_rank * L.width * S.dense_px;
            if ((e = m->rccl.Recv(S.d_dense + (size_t)r * share, share, NCCL_UINT8, (int)r, m->comms[0], S.rx_stream)) != 0) return nccl_fail(m, e, "ncclRecv");
        } else {
            const ShareShape sh = share_shape(m, S.n_frames);
            const uint32_t nf = m->est[r] ? (m->est[r] < sh.capacity ? m->est[r] : sh.capacity) : sh.capacity;   // what rank r sends (same state, read before the jobs start)
            if ((e = m->rccl.Recv(S.rb[r].d_gather, sh.fixed + (size_t)nf * 4, NCCL_UINT8, (int)r, m->comms[0], S.rx_stream)) != 0) return nccl_fail(m, e, "ncclRecv");
        }
    }
    if ((e = m->rccl.GroupEnd()) != 0) return nccl_fail(m, e, "ncclGroupEnd");
    M_TRY(hipEventRecord(S.ev_rx, S.rx_stream));
    return SDFHIP_OK;
}

// A submission (or the wait for one) failed half-way: whatever was issued is drained, the slot is free again and marked so
// that the shares' counters are re-zeroed before its next use.  With the RCCL transport receives may be outstanding whose
// sends were never issued; they cannot be taken back, so the handle refuses further work (sdfhip_multi_free aborts the comms).
int abort_slot(sdfhip_multi *m, Slot &S, int rc)
{
    char msg[256];
    strncpy(msg, sdfhip_last_error(), sizeof msg - 1); msg[sizeof msg - 1] = 0;
    if (m->use_rccl && m->n > 1) m->broken = true;
    else {
        for (uint32_t r = 0; r < m->n; r++) {
            DevGuard g(m->devices[r]);
            if (S.rb[r].stream) (void)hipStreamSynchronize(S.rb[r].stream);
        }
        (void)hipGetLastError();
    }
    S.busy = false; S.dirty = true;
    return fail(rc, "%s", msg);
}

int submit_locked(sdfhip_multi *m, uint32_t slot, const sdfhip_info *infos, uint32_t n_frames, const sdfhip_pathtrace *pt,
                  uint32_t width, uint32_t height, uint32_t flags, void *d_out, bool timed = true)
{
    if (slot >= MAX_SLOTS) return fail(SDFHIP_ERR_ARG, "multi_submit: slot %u of %u", slot, MAX_SLOTS);
    if (!infos || n_frames == 0 || n_frames > MAX_GROUP || width == 0 || height == 0)
        return fail(SDFHIP_ERR_ARG, "multi_submit: null argument, zero-sized frame or n_frames outside 1..%u", MAX_GROUP);
    if (pt && n_frames != 1) return fail(SDFHIP_ERR_ARG, "multi_submit: the path-traced mode renders one frame per submission");
    if (flags & ~(uint32_t)(SDFHIP_FLAG_DISPLAY | SDFHIP_FLAG_DISPLAY_DEBUG | SDFHIP_FLAG_TILE_ORDER))
        return fail(SDFHIP_ERR_ARG, "multi_submit: flags %#x are not available across devices (0, SDFHIP_FLAG_DISPLAY[_DEBUG], SDFHIP_FLAG_TILE_ORDER)", flags);
    if (pt && (flags & (SDFHIP_FLAG_DISPLAY | SDFHIP_FLAG_DISPLAY_DEBUG)))
        return fail(SDFHIP_ERR_ARG, "multi_submit: the display pass is not available in path-traced mode");
    if (m->broken) return fail(SDFHIP_ERR_DEVICE, "multi_submit: an earlier submission failed with RCCL receives outstanding; free this handle and create a new one");
    Slot &S = m->slots[slot];
    if (S.busy) return fail(SDFHIP_ERR_ARG, "multi_submit: slot %u is in flight (sdfhip_multi_wait it first)", slot);
    const uint32_t frame_bands = (height + m->band_rows - 1) / m->band_rows;
    if (frame_bands > 512) return fail(SDFHIP_ERR_ARG, "multi_submit: %u bands of %u rows (at most 512: choose larger bands)", frame_bands, m->band_rows);
    if (m->lay.width != width || m->lay.height != height || m->lay.band_rows != m->band_rows || m->lay.weight != m->rank0_weight ||
        m->lay.world != m->n) {
        // another geometry: every slot must be idle (their buffers and the float estimates belong to the old one)
        for (uint32_t k = 0; k < MAX_SLOTS; k++)
            if (m->slots[k].busy) return fail(SDFHIP_ERR_ARG, "multi_submit: the frame geometry changed while slot %u is in flight", k);
        // dealt into a layout of its own and moved over the handle's when it is complete: a deal that throws half-way (its vectors;
        // found by tests/test_gpu_fault_injection.py) must not leave a layout that says "this geometry" and owns no rows
        Layout fresh;
        deal_bands(fresh, width, height, m->n, m->band_rows, m->rank0_weight);
        m->lay = std::move(fresh);
        for (uint32_t r = 0; r < MAX_RANKS; r++) m->est[r] = 0;
    }
    const bool display = (flags & (SDFHIP_FLAG_DISPLAY | SDFHIP_FLAG_DISPLAY_DEBUG)) != 0;
    const bool dense = pt != nullptr || m->dense_only;
    int rc = prepare_slot(m, S, n_frames, dense, d_out == nullptr, display ? 4 : 16);
    if (rc != SDFHIP_OK) return rc;
    S.out = d_out ? d_out : S.d_frames;
    S.path = dense; S.dense_px = display ? 4u : 16u; S.n_frames = n_frames; S.width = width; S.height = height; S.flags = flags;
    S.t_submit = std::chrono::steady_clock::now();
    m->job.m = m; m->job.slot = slot; m->job.infos = infos; m->job.n_frames = n_frames; m->job.width = width; m->job.height = height;
    // sparse shares: the display pass runs where the frame is assembled; dense bands: where they are rendered
    m->job.flags = dense ? flags : flags & ~(uint32_t)(SDFHIP_FLAG_DISPLAY | SDFHIP_FLAG_DISPLAY_DEBUG);
    m->job.pt = pt; m->job.dense = dense; m->job.timed = timed;
    S.timed = timed;
    S.dirty = true;                                   // until everything below has been issued
    rc = post_receives(m, S);
    if (rc == SDFHIP_OK) rc = on_all_ranks(m, rank_submit);
    if (rc == SDFHIP_OK) rc = assemble(m, S, -1);
    if (rc == SDFHIP_OK) {
        DevGuard g0(m->devices[0]);
        const hipError_t e = hipEventRecord(S.ev_done, S.rb[0].stream);
        if (e != hipSuccess) rc = fail(SDFHIP_ERR_DEVICE, "multi: queueing the frame's completion failed: %s", hipGetErrorString(e));
    }
    if (rc != SDFHIP_OK) return abort_slot(m, S, rc);
    S.dirty = false;
    S.busy = true;
    return SDFHIP_OK;
}

int wait_body(sdfhip_multi *m, Slot &S, uint32_t &resent)
{
    const Layout &L = m->lay;
    {
        DevGuard g0(m->devices[0]);
        M_TRY(hipEventSynchronize(S.ev_done));
    }
    if (!S.path) {
        const ShareShape sh = share_shape(m, S.n_frames);
        for (uint32_t r = 0; r < m->n; r++) {
            if (L.bands[r].empty()) continue;
            RankBuf &B = S.rb[r];
            const uint32_t handed = S.h_counts[r] - B.count_base;          // slots this launch handed out (the counter runs on, modulo 2^32)
            B.count_base = S.h_counts[r];
            const uint32_t used = handed < sh.capacity ? handed : sh.capacity;
            if ((r > 0 || m->rccl_self) && used > B.sent) {
                // the share needed more floats than were sent with it: the tail now, and this rank's rows again
                {
                    DevGuard g(m->devices[r]);
                    M_TRY(push(B.d_gather + sh.off_floats + (size_t)B.sent * 4, m->devices[0],
                               B.d_share + sh.off_floats + (size_t)B.sent * 4, m->devices[r], (size_t)(used - B.sent) * 4, B.stream));
                    M_TRY(hipEventRecord(B.ev_sent, B.stream));
                }
                int rc = assemble(m, S, (int)r);
                if (rc != SDFHIP_OK) return rc;
                DevGuard g0(m->devices[0]);
                M_TRY(hipStreamSynchronize(S.rb[0].stream));
                resent++;
                B.sent = used;
            }
            // the next share of this rank: a quarter more than this one used, in steps of 1024
            const uint64_t want = ((uint64_t)used + used / 4 + 1024 + 1023) / 1024 * 1024;
            const uint32_t e = (uint32_t)(want < sh.capacity ? want : sh.capacity);
            m->est[r] = (m->est[r] == 0 || e > m->est[r]) ? e : (m->est[r] - (m->est[r] - e) / 8);   // up at once, down slowly
            S.last_used[r] = used;
        }
    }
    return SDFHIP_OK;
}

int wait_locked(sdfhip_multi *m, uint32_t slot, void **d_frames, sdfhip_multi_stats *stats)
{
    if (slot >= MAX_SLOTS) return fail(SDFHIP_ERR_ARG, "multi_wait: slot %u of %u", slot, MAX_SLOTS);
    Slot &S = m->slots[slot];
    if (!S.busy) return fail(SDFHIP_ERR_ARG, "multi_wait: nothing was submitted to slot %u", slot);
    const Layout &L = m->lay;
    uint32_t resent = 0;
    const int rcw = wait_body(m, S, resent);
    if (rcw != SDFHIP_OK) return abort_slot(m, S, rcw);
    m->resends += resent;
    S.busy = false;
    if (d_frames) *d_frames = S.out;
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->n_devices = m->n;
        stats->resends = resent;
        stats->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - S.t_submit).count();
        for (uint32_t r = 0; r < m->n && r < 16; r++) {
            if (L.bands[r].empty()) continue;
            DevGuard g(m->devices[r]);
            float ms = 0.0f;
            if (S.timed && hipEventElapsedTime(&ms, S.rb[r].ev_start, S.rb[r].ev_sent) == hipSuccess) stats->rank_ms[r] = ms;
            (void)hipGetLastError();
            if (!S.path) {
                const ShareShape sh = share_shape(m, S.n_frames);
                stats->floats_used[r] = S.last_used[r];
                if (r > 0) stats->gathered_bytes += sh.fixed + (uint64_t)S.rb[r].sent * 4;
            } else if (r > 0) {
                stats->gathered_bytes += (uint64_t)S.n_frames * L.rows_per_rank * L.width * S.dense_px;
            }
        }
    }
    return SDFHIP_OK;
}

}  // namespace

extern "C" int sdfhip_multi_free(sdfhip_multi *m)
try {
    if (!m) return SDFHIP_OK;
    for (uint32_t r = 1; r < m->n; r++) {
        Worker *w = m->workers[r];
        if (!w) continue;
        { std::lock_guard<std::mutex> lk(w->mu); w->quit.store(true); }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
        delete w;
    }
    for (uint32_t k = 0; k < MAX_SLOTS; k++) {
        Slot &S = m->slots[k];
        for (uint32_t r = 0; r < m->n; r++) {
            RankBuf &B = S.rb[r];
            DevGuard g(m->devices[r]);
            if (B.stream) (void)hipStreamSynchronize(B.stream);
            if (B.d_share) (void)hipFree(B.d_share);
            if (B.d_bands) (void)hipFree(B.d_bands);
            if (B.ev_start) (void)hipEventDestroy(B.ev_start);
            if (B.ev_sent) (void)hipEventDestroy(B.ev_sent);
            if (B.stream) (void)hipStreamDestroy(B.stream);
        }
        DevGuard g0(m->devices[0]);
        for (uint32_t r = 0; r < m->n; r++) if (S.rb[r].d_gather) (void)hipFree(S.rb[r].d_gather);
        if (S.rx_stream) { (void)hipStreamSynchronize(S.rx_stream); (void)hipStreamDestroy(S.rx_stream); }
        if (S.d_frames) (void)hipFree(S.d_frames);
        if (S.d_dense) (void)hipFree(S.d_dense);
        if (S.h_counts) (void)hipHostFree(S.h_counts);
        if (S.ev_rx) (void)hipEventDestroy(S.ev_rx);
        if (S.ev_done) (void)hipEventDestroy(S.ev_done);
        if (S.ev_t0) (void)hipEventDestroy(S.ev_t0);
    }
    if (m->use_rccl)
        for (uint32_t r = 0; r < m->n; r++)
            if (m->comms[r]) (void)((m->broken && m->rccl.CommAbort) ? m->rccl.CommAbort(m->comms[r]) : m->rccl.CommDestroy(m->comms[r]));
    for (uint32_t r = 0; r < m->n; r++) if (m->scenes[r]) (void)sdfhip_scene_free(m->scenes[r]);
    delete m;
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_multi_free)

extern "C" int sdfhip_multi_create(const int *devices, uint32_t n_devices, const int32_t *structs, const uint8_t *values,
                                   uint32_t n, sdfhip_multi **out)
try {
    if (!devices || !structs || !values || !out || n == 0 || n_devices == 0)
        return fail(SDFHIP_ERR_ARG, "multi_create: null argument, empty scene or empty device list");
    *out = nullptr;
    if (n_devices > MAX_RANKS) return fail(SDFHIP_ERR_ARG, "multi_create: %u devices (at most %u)", n_devices, MAX_RANKS);
    int ndev = 0;
    M_TRY(hipGetDeviceCount(&ndev));
    bool distinct = true;
    for (uint32_t r = 0; r < n_devices; r++) {
        if (devices[r] < 0 || devices[r] >= ndev) return fail(SDFHIP_ERR_DEVICE, "multi_create: device %d of %d does not exist", devices[r], ndev);
        for (uint32_t q = 0; q < r; q++) if (devices[q] == devices[r]) distinct = false;   // (allowed: several ranks on one device rehearse the pipeline)
    }
    sdfhip_multi *m = new (std::nothrow) sdfhip_multi();
    if (!m) return fail(SDFHIP_ERR_NOMEM, "multi_create: out of host memory");
    m->n = n_devices;
    for (uint32_t r = 0; r < MAX_RANKS; r++) { m->devices[r] = r < n_devices ? devices[r] : 0; m->scenes[r] = nullptr; m->workers[r] = nullptr; m->est[r] = 0; m->comms[r] = nullptr; }
    if (const char *e = lab_env("SDFHIP_MULTI_BAND_ROWS")) { const int v = atoi(e); if (v >= 8 && v <= 4096 && v % 8 == 0) m->band_rows = (uint32_t)v; }
    if (const char *e = lab_env("SDFHIP_MULTI_RANK0_WEIGHT")) { const float v = (float)atof(e); if (v > 0.0f && v <= 1.0f) m->rank0_weight = v; }
    auto bail = [&](int rc) { char msg[256]; strncpy(msg, sdfhip_last_error(), sizeof msg - 1); msg[sizeof msg - 1] = 0; sdfhip_multi_free(m); return fail(rc, "%s", msg); };
    for (uint32_t r = 1; r < n_devices; r++) {
        Worker *w = new (std::nothrow) Worker();
        if (!w) return bail(fail(SDFHIP_ERR_NOMEM, "multi_create: out of host memory"));
        w->rank = r;
        m->workers[r] = w;
        try { w->th = std::thread([w] { w->run(); }); }
        catch (...) { return bail(abi_caught("multi_create: the worker thread of a rank")); }     // (std::system_error: no more threads)
    }
    // peer access into rank 0's memory (the gather writes there)
    for (uint32_t r = 1; r < n_devices; r++) {
        if (devices[r] == devices[0]) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devices[r], devices[0]) != hipSuccess || !can) continue;      // hipMemcpyPeerAsync then stages through the host
        DevGuard g(devices[r]);
        const hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        (void)hipGetLastError();
    }
    for (uint32_t k = 0; k < MAX_SLOTS; k++) {
        Slot &S = m->slots[k];
        for (uint32_t r = 0; r < n_devices; r++) {
            RankBuf &B = S.rb[r];
            DevGuard g(devices[r]);
            hipError_t e;
            if ((e = hipStreamCreateWithFlags(&B.stream, hipStreamNonBlocking)) != hipSuccess ||
                (e = hipEventCreate(&B.ev_start)) != hipSuccess || (e = hipEventCreate(&B.ev_sent)) != hipSuccess)
                return bail(fail(SDFHIP_ERR_DEVICE, "multi_create: stream / event on device %d: %s", devices[r], hipGetErrorString(e)));
        }
        DevGuard g0(devices[0]);
        hipError_t e;
        if ((e = hipHostMalloc((void **)&S.h_counts, MAX_RANKS * sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess)
            return bail(fail(SDFHIP_ERR_DEVICE, "multi_create: pinned memory: %s", hipGetErrorString(e)));
        memset(S.h_counts, 0, MAX_RANKS * sizeof(uint32_t));
        if ((e = hipStreamCreateWithFlags(&S.rx_stream, hipStreamNonBlocking)) != hipSuccess || (e = hipEventCreate(&S.ev_rx)) != hipSuccess ||
            (e = hipEventCreate(&S.ev_done)) != hipSuccess || (e = hipEventCreate(&S.ev_t0)) != hipSuccess)
            return bail(fail(SDFHIP_ERR_DEVICE, "multi_create: stream / event on device %d: %s", devices[0], hipGetErrorString(e)));
    }
    // the scene on every device, uploaded by the ranks' own threads at the same time
    m->job.structs = structs; m->job.values = values; m->job.n_nodes = n;
    int rc = on_all_ranks(m, rank_upload);
    if (rc != SDFHIP_OK) return bail(rc);
    // sparse shares come from the default kernel, which needs a grid as deep as the tree; a scene without one (deeper than 12
    // levels, inconsistent links, or no memory for the grid on some device) is gathered as dense bands instead
    for (uint32_t r = 0; r < n_devices; r++) if (!scene_has_full_depth_grid(m->scenes[r])) m->dense_only = true;
    // transport of the gather
    const char *tr = getenv("SDFHIP_MULTI_TRANSPORT");
    if (tr && strcmp(tr, "rccl") == 0) {
        m->rccl_self = n_devices == 1 && getenv("SDFHIP_MULTI_RCCL_SELF") && atoi(getenv("SDFHIP_MULTI_RCCL_SELF")) != 0;
        if (!distinct) return bail(fail(SDFHIP_ERR_ARG, "multi_create: SDFHIP_MULTI_TRANSPORT=rccl needs distinct devices (one RCCL rank per GPU)"));
        if (n_devices > 1 || m->rccl_self) {
            if (!m->rccl.load()) return bail(fail(SDFHIP_ERR_DEVICE, "multi_create: librccl.so could not be loaded: %s", dlerror()));
            const int e = m->rccl.CommInitAll(m->comms, (int)n_devices, m->devices);
            if (e) return bail(nccl_fail(m, e, "ncclCommInitAll"));
            m->use_rccl = true;
        }
    }
    // first contact with the links, before any frame depends on them
    rc = sdfhip_multi_selftest(m, nullptr);
    if (rc != SDFHIP_OK) return bail(rc);
    *out = m;
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_multi_create)

extern "C" int sdfhip_multi_selftest(sdfhip_multi *m, sdfhip_multi_link *links)
try {
    if (!m) return fail(SDFHIP_ERR_ARG, "multi_selftest: null handle");
    std::lock_guard<std::mutex> lk(m->lock);
    for (uint32_t k = 0; k < MAX_SLOTS; k++)
        if (m->slots[k].busy) return fail(SDFHIP_ERR_ARG, "multi_selftest: slot %u is in flight", k);
    if (m->broken) return fail(SDFHIP_ERR_DEVICE, "multi_selftest: the handle's RCCL transport is in an undefined state");
    constexpr size_t BYTES = 1u << 20;
    std::vector<uint8_t> pattern(BYTES), back(BYTES);
    Slot &S = m->slots[0];
    uint8_t *d_rx = nullptr;
    { DevGuard g0(m->devices[0]); M_TRY(device_alloc((void **)&d_rx, BYTES)); }
    int rc = SDFHIP_OK;
    char first[256] = { 0 };
    for (uint32_t r = 0; r < m->n; r++) {
        sdfhip_multi_link L;
        memset(&L, 0, sizeof L);
        L.device = m->devices[r];
        (void)hipDeviceGetPCIBusId(L.pci_bus_id, (int)sizeof L.pci_bus_id, m->devices[r]);
        (void)hipGetLastError();
        L.ok = 1; L.peer_access = -1;
        if (r > 0 || m->rccl_self) {
            if (m->devices[r] != m->devices[0]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, m->devices[r], m->devices[0]) != hipSuccess) { (void)hipGetLastError(); can = 0; }
                L.peer_access = can ? 1 : 0;
            }
            for (size_t i = 0; i < BYTES; i++) pattern[i] = (uint8_t)((i * 2654435761u + r * 97u + (i >> 11)) >> 7);
            uint8_t *d_tx = nullptr;
            hipError_t e = hipSuccess;
            int ne = 0;
            float ms = 0.0f;
            {
                DevGuard g(m->devices[r]);
                RankBuf &B = S.rb[r];
                if ((e = device_alloc((void **)&d_tx, BYTES)) == hipSuccess &&
                    (e = hipMemcpyAsync(d_tx, pattern.data(), BYTES, hipMemcpyHostToDevice, B.stream)) == hipSuccess) {
                    { DevGuard g0(m->devices[0]); e = hipMemsetAsync(d_rx, 0, BYTES, S.rb[0].stream); if (e == hipSuccess) e = hipStreamSynchronize(S.rb[0].stream); }
                    if (e == hipSuccess) e = hipEventRecord(B.ev_start, B.stream);
                    if (e == hipSuccess) {
                        if (m->use_rccl) {
                            // the gather's own pattern: the receive on devices[0]'s receive stream, the send on the rank's stream, each in a group
                            // (one group for both when the rank IS devices[0]: the self test of one-GPU boxes)
                            DevGuard g0(m->devices[0]);
                            if (r == 0) {
                                ne = m->rccl.GroupStart();
                                if (!ne) ne = m->rccl.Send(d_tx, BYTES, NCCL_UINT8, 0, m->comms[0], B.stream);
                                if (!ne) ne = m->rccl.Recv(d_rx, BYTES, NCCL_UINT8, 0, m->comms[0], B.stream);
                                if (!ne) ne = m->rccl.GroupEnd();
                            } else {
                                ne = m->rccl.GroupStart();
                                if (!ne) ne = m->rccl.Recv(d_rx, BYTES, NCCL_UINT8, (int)r, m->comms[0], S.rx_stream);
                                if (!ne) ne = m->rccl.GroupEnd();
                                DevGuard gr(m->devices[r]);
                                if (!ne) ne = m->rccl.GroupStart();
                                if (!ne) ne = m->rccl.Send(d_tx, BYTES, NCCL_UINT8, 0, m->comms[r], B.stream);
                                if (!ne) ne = m->rccl.GroupEnd();
                            }
                        } else {
                            e = push(d_rx, m->devices[0], d_tx, m->devices[r], BYTES, B.stream);
                        }
                    }
                    if (e == hipSuccess && !ne) e = hipEventRecord(B.ev_sent, B.stream);
                    if (e == hipSuccess && !ne) e = hipStreamSynchronize(B.stream);
                    if (e == hipSuccess && !ne) { (void)hipEventElapsedTime(&ms, B.ev_start, B.ev_sent); (void)hipGetLastError(); }
                }
                if (d_tx) (void)hipFree(d_tx);
            }
            if (e == hipSuccess && !ne) {
                DevGuard g0(m->devices[0]);
                if (m->use_rccl && r > 0) e = hipStreamSynchronize(S.rx_stream);
                if (e == hipSuccess) e = hipMemcpy(back.data(), d_rx, BYTES, hipMemcpyDeviceToHost);
            }
            L.push_ms = ms;
            L.ok = (e == hipSuccess && !ne && memcmp(back.data(), pattern.data(), BYTES) == 0) ? 1u : 0u;
            if (!L.ok && rc == SDFHIP_OK) {
                size_t bad = 0;
                while (bad < BYTES && back[bad] == pattern[bad]) bad++;
                rc = SDFHIP_ERR_DEVICE;
                snprintf(first, sizeof first, "multi_selftest: device %d (%s) -> device %d by %s: %s", m->devices[r], L.pci_bus_id, m->devices[0],
                         m->use_rccl ? "ncclSend / ncclRecv" : (L.peer_access == 1 ? "hipMemcpyPeerAsync over a peer link" : "hipMemcpyPeerAsync (staged: no peer access)"),
                         e != hipSuccess ? hipGetErrorString(e) : ne ? (m->rccl.GetErrorString ? m->rccl.GetErrorString(ne) : "RCCL error") :
                         bad < BYTES ? "the pattern arrived damaged" : "?");
                if (e == hipSuccess && !ne && bad < BYTES) { const size_t l = strlen(first); snprintf(first + l, sizeof first - l, " (first wrong byte at %zu of %zu)", bad, BYTES); }
                (void)hipGetLastError();
                if (m->use_rccl && (e != hipSuccess || ne)) m->broken = true;
            }
        }
        if (links) links[r] = L;
    }
    { DevGuard g0(m->devices[0]); (void)hipFree(d_rx); }
    return rc == SDFHIP_OK ? SDFHIP_OK : fail(rc, "%s", first);
}
SDFHIP_ABI_CATCH(sdfhip_multi_selftest)

extern "C" int sdfhip_multi_configure(sdfhip_multi *m, uint32_t band_rows, float rank0_weight)
try {
    if (!m) return fail(SDFHIP_ERR_ARG, "multi_configure: null handle");
    if (band_rows == 0 || band_rows % 8 != 0) return fail(SDFHIP_ERR_ARG, "multi_configure: band_rows %u must be a positive multiple of 8 (whole 8x8 wave tiles per band)", band_rows);
    if (!(rank0_weight > 0.0f && rank0_weight <= 1.0f)) return fail(SDFHIP_ERR_ARG, "multi_configure: rank0_weight must be in (0, 1]");
    std::lock_guard<std::mutex> lk(m->lock);
    for (uint32_t k = 0; k < MAX_SLOTS; k++)
        if (m->slots[k].busy) return fail(SDFHIP_ERR_ARG, "multi_configure: slot %u is in flight", k);
    m->band_rows = band_rows; m->rank0_weight = rank0_weight;
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_multi_configure)

#ifdef SDFHIP_EXPERIMENTS
#include "../../include/sdfhip_experimental.h"
extern "C" int sdfhip_multi_debug_floats_sent(sdfhip_multi *m, uint32_t floats)
try {
    if (!m) return fail(SDFHIP_ERR_ARG, "multi_debug_floats_sent: null handle");
    std::lock_guard<std::mutex> lk(m->lock);
    for (uint32_t r = 0; r < MAX_RANKS; r++) m->est[r] = floats;
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_multi_debug_floats_sent)
#endif

extern "C" int sdfhip_multi_info(const sdfhip_multi *m, uint32_t *n_devices, int *devices, uint32_t *band_rows, float *rank0_weight, int *transport)
try {
    if (!m) return fail(SDFHIP_ERR_ARG, "multi_info: null handle");
    if (n_devices) *n_devices = m->n;
    if (devices) for (uint32_t r = 0; r < m->n; r++) devices[r] = m->devices[r];
    if (band_rows) *band_rows = m->band_rows;
    if (rank0_weight) *rank0_weight = m->rank0_weight;
    if (transport) *transport = m->use_rccl ? 1 : 0;
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_multi_info)

extern "C" int sdfhip_multi_submit(sdfhip_multi *m, uint32_t slot, const sdfhip_info *infos, uint32_t n_frames, uint32_t width,
                                   uint32_t height, uint32_t flags, void *d_frames_out)
try {
    if (!m) return fail(SDFHIP_ERR_ARG, "multi_submit: null handle");
    std::lock_guard<std::mutex> lk(m->lock);
    return submit_locked(m, slot, infos, n_frames, nullptr, width, height, flags, d_frames_out);
}
SDFHIP_ABI_CATCH(sdfhip_multi_submit)

extern "C" int sdfhip_multi_submit_path(sdfhip_multi *m, uint32_t slot, const sdfhip_info *info, const sdfhip_pathtrace *pt,
                                        uint32_t width, uint32_t height, uint32_t flags, void *d_frame_out)
try {
    if (!m || !pt) return fail(SDFHIP_ERR_ARG, "multi_submit_path: null argument");
    std::lock_guard<std::mutex> lk(m->lock);
    return submit_locked(m, slot, info, 1, pt, width, height, flags, d_frame_out);
}
SDFHIP_ABI_CATCH(sdfhip_multi_submit_path)

extern "C" int sdfhip_multi_wait(sdfhip_multi *m, uint32_t slot, void **d_frames, sdfhip_multi_stats *stats)
try {
    if (!m) return fail(SDFHIP_ERR_ARG, "multi_wait: null handle");
    std::lock_guard<std::mutex> lk(m->lock);
    return wait_locked(m, slot, d_frames, stats);
}
SDFHIP_ABI_CATCH(sdfhip_multi_wait)

static int multi_render_host(sdfhip_multi *m, const sdfhip_info *info, const sdfhip_pathtrace *pt, uint32_t width, uint32_t height,
                             uint32_t flags, void *out, sdfhip_multi_stats *stats)
{
    if (!m || !info || !out) return fail(SDFHIP_ERR_ARG, "multi_render: null argument");
    std::lock_guard<std::mutex> lk(m->lock);
    const auto t0 = std::chrono::steady_clock::now();
    // (per-rank start events only when somebody asked for statistics.  The copy to the host array is issued AFTER the wait: queued
    // behind the assembly in its stream -- one wait for both -- a copy into pageable memory made the call slower, 0.072 -> 0.082 ms
    // for a 64x64 frame: the runtime stages it with the host waiting inside the copy call)
    int rc = submit_locked(m, 0, info, 1, pt, width, height, flags, nullptr, stats != nullptr);
    if (rc != SDFHIP_OK) return rc;
    void *d = nullptr;
    rc = wait_locked(m, 0, &d, stats);
    if (rc != SDFHIP_OK) return rc;
    const size_t px = (flags & (SDFHIP_FLAG_DISPLAY | SDFHIP_FLAG_DISPLAY_DEBUG)) ? 4 : 16;
    DevGuard g0(m->devices[0]);
    M_TRY(hipMemcpyAsync(out, d, (size_t)width * height * px, hipMemcpyDeviceToHost, m->slots[0].rb[0].stream));
    M_TRY(hipStreamSynchronize(m->slots[0].rb[0].stream));
    if (stats) stats->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return SDFHIP_OK;
}

extern "C" int sdfhip_multi_render(sdfhip_multi *m, const sdfhip_info *info, uint32_t width, uint32_t height, uint32_t flags,
                                   float *rgba_out, sdfhip_multi_stats *stats)
try {
    return multi_render_host(m, info, nullptr, width, height, flags, rgba_out, stats);
}
SDFHIP_ABI_CATCH(sdfhip_multi_render)

extern "C" int sdfhip_multi_render_path(sdfhip_multi *m, const sdfhip_info *info, const sdfhip_pathtrace *pt, uint32_t width,
                                        uint32_t height, uint32_t flags, float *rgba_out, sdfhip_multi_stats *stats)
try {
    if (!pt) return fail(SDFHIP_ERR_ARG, "multi_render_path: null argument");
    return multi_render_host(m, info, pt, width, height, flags, rgba_out, stats);
}
SDFHIP_ABI_CATCH(sdfhip_multi_render_path)
