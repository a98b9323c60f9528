// libsdfhip.so, device half, host side: the scene handle -- upload (fused records, validation on the device, the lookup
// grids), the per-stream scratch of the render launches, the path tracer's second grid.
//
// Replaces the reference's resource creation and binding:
//   SdfBox/Program.cs:543-572,147-152     StructBuffer/ValueTexture + binding -> sdfhip_scene_upload
#include "upload_kernels.h"
#include "scene.h"
#include "abi_guard.h"

#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <new>
#include <utility>

using namespace sdfhip;

#ifndef FULL_GRID_SHARE
#define FULL_GRID_SHARE 64          // a grid as deep as the tree may take 1/FULL_GRID_SHARE of the device's memory
#endif
#ifndef DENSE_GRID_MAX_BYTES
#define DENSE_GRID_MAX_BYTES (512ull << 20)   // ... and by default no more than this (depth <= 8): deeper trees get a split grid
#endif

extern "C" int sdfhip_device_count(int *count)
try {
    if (!count) return fail(SDFHIP_ERR_ARG, "device_count: null argument");
    *count = 0;
    HIP_TRY(hipGetDeviceCount(count));
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_device_count)

extern "C" int sdfhip_device_pci_bus_id(int device, char *out, uint32_t len)
try {
    if (!out || len < 16) return fail(SDFHIP_ERR_ARG, "device_pci_bus_id: the buffer must hold 16 bytes");
    out[0] = 0;
    HIP_TRY(hipDeviceGetPCIBusId(out, (int)len, device));
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_device_pci_bus_id)

extern "C" int sdfhip_scene_free(sdfhip_scene *s)
try {
    if (!s) return SDFHIP_OK;
    {
        DeviceGuard g(s->device);
        if (s->stream) (void)hipStreamSynchronize(s->stream);
        if (s->alloc) (void)hipFree(s->alloc);
        if (s->d_verdict) (void)hipFree(s->d_verdict);
        if (s->d_top) (void)hipFree(s->d_top);
        if (s->d_fine) (void)hipFree(s->d_fine);
#ifdef SDFHIP_EXPERIMENTS
        if (s->d_d4) (void)hipFree(s->d_d4);
        for (int a = 0; a < 4; a++) if (s->touch.bits[a]) (void)hipFree(s->touch.bits[a]);      // (a sdfhip_debug_touch_begin without its _end)
        if (s->touch.result) (void)hipFree(s->touch.result);
        if (s->d_recs) (void)hipFree(s->d_recs);
#endif
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
            if (s->scratch[i].band_list) (void)hipFree(s->scratch[i].band_list);
        }
        for (int i = 0; i < sdfhip_scene::MAX_TICKETS; i++) {
            if (s->tickets[i].ev0) (void)hipEventDestroy(s->tickets[i].ev0);
            if (s->tickets[i].ev1) (void)hipEventDestroy(s->tickets[i].ev1);
            if (s->tickets[i].h_counters) (void)hipHostFree(s->tickets[i].h_counters);
        }
        if (s->d_frame) (void)hipFree(s->d_frame);
        for (int b = 0; b < sdfhip_scene::HOST_BANDS; b++) {
            if (s->band_stream[b]) { (void)hipStreamSynchronize(s->band_stream[b]); (void)hipStreamDestroy(s->band_stream[b]); }
            if (s->band_done[b]) (void)hipEventDestroy(s->band_done[b]);
        }
        if (s->stream) (void)hipStreamDestroy(s->stream);
    }
    delete s;
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_scene_free)

// (see scene.h)
bool sdfhip::build_split_grid(sdfhip_scene *s, int C, int FB, int order, uint64_t max_fine_bytes, TopCell **coarse_out, TopCell **fine_out,
                             uint64_t *fine_bytes_out)
{
    const size_t ncell = (size_t)1 << (3 * C);
    const uint32_t n_chunks = (uint32_t)((ncell + 255) / 256);
    uint32_t *d_block_node = nullptr, *d_chunks = nullptr;
    TopCell *d_coarse = nullptr, *d_fine = nullptr;
    bool ok = false;
    do {
        if (device_alloc((void **)&d_coarse, ncell * sizeof(TopCell)) != hipSuccess) break;
        if (device_alloc((void **)&d_chunks, ((size_t)n_chunks + 1) * 4) != hipSuccess) break;
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
            if (device_alloc((void **)&d_fine, fine_bytes) != hipSuccess) break;
            if (device_alloc((void **)&d_block_node, nblocks * 4) != hipSuccess) break;
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

extern "C" int sdfhip_scene_upload(int device, const int32_t *structs, const uint8_t *values,
                                   uint32_t n, sdfhip_scene **out)
try {
    return sdfhip::scene_from_arrays(device, structs, values, n, false, nullptr, out);
}
SDFHIP_ABI_CATCH(sdfhip_scene_upload)

extern "C" void sdfhip_upload_options_default(sdfhip_upload_options *opt)
try {
    if (!opt) return;
    opt->size = (uint32_t)sizeof *opt;
    opt->top_grid_level = opt->top_grid_split = opt->scatter_grid = opt->scatter_order = -1;
}
SDFHIP_ABI_CATCH_VOID(sdfhip_upload_options_default)

extern "C" int sdfhip_scene_upload_ex(int device, const int32_t *structs, const uint8_t *values, uint32_t n,
                                      const sdfhip_upload_options *opt, sdfhip_scene **out)
try {
    // `size` is what lets the struct grow: a caller built against an OLDER header passes a smaller struct -- its fields are
    // taken and the ones it does not know stay at -1 ("choose"); a caller built against a NEWER header passes a larger one -- accepted
    // when the part this library does not know is all -1, i.e. asks for nothing it cannot give.  Version 1 ends behind scatter_order.
    constexpr uint32_t V1_BYTES = (uint32_t)(offsetof(sdfhip_upload_options, scatter_order) + sizeof(int32_t));
    sdfhip_upload_options local;
    if (opt) {
        if (opt->size < V1_BYTES)
            return fail(SDFHIP_ERR_ARG, "scene_upload_ex: options of %u bytes (version 1 has %u: start from sdfhip_upload_options_default)", opt->size, V1_BYTES);
        //  ... but `size` is also the one field a caller can leave uninitialised: a struct this library does not know is
        // scanned up to `size` below, so a garbage value must not send that scan through the caller's memory.  No header of this
        // library will be 16 times today's.
        if (opt->size > 16u * (uint32_t)sizeof local)
            return fail(SDFHIP_ERR_ARG, "scene_upload_ex: options of %u bytes (this library's are %zu; start from sdfhip_upload_options_default)",
                        opt->size, sizeof local);
        memset(&local, 0xFF, sizeof local);                  // every field -1
        const size_t take = opt->size < sizeof local ? opt->size : sizeof local;
        memcpy(&local, opt, take);
        local.size = (uint32_t)sizeof local;
        for (size_t i = sizeof local; i + sizeof(int32_t) <= opt->size; i += sizeof(int32_t)) {
            int32_t v;
            memcpy(&v, reinterpret_cast<const char *>(opt) + i, sizeof v);
            if (v != -1) return fail(SDFHIP_ERR_ARG, "scene_upload_ex: options of %u bytes set a field at offset %zu that this library (%zu bytes) does not know", opt->size, i, sizeof local);
        }
        opt = &local;
    }
    if (opt && (opt->top_grid_level < -1 || opt->top_grid_level > 10 || opt->top_grid_split < -1 || opt->top_grid_split > 8 ||
                opt->scatter_grid < -1 || opt->scatter_grid > 4 || opt->scatter_order < -1 || opt->scatter_order > 1))
        return fail(SDFHIP_ERR_ARG, "scene_upload_ex: an option out of range (top_grid_level -1..10, top_grid_split -1..8, scatter_grid -1..4, scatter_order -1..1)");
    return sdfhip::scene_from_arrays(device, structs, values, n, false, opt, out);
}
SDFHIP_ABI_CATCH(sdfhip_scene_upload_ex)

// structs / values on the host (sdfhip_scene_upload), or already in `device`'s memory (sdfhip_sdfgen_scene: the tree the GPU
// builder has just made never leaves HBM)
int sdfhip::scene_from_arrays(int device, const int32_t *structs, const uint8_t *values, uint32_t n, bool resident, const sdfhip_upload_options *opt,
                              sdfhip_scene **out, int trusted_depth)
{
    // the grid choices: the caller's (sdfhip_scene_upload_ex), else -- laboratory library only -- the environment's, else ours
    auto choice = [](int32_t given, const char *env_name, int lo, int hi) {
        if (given >= 0) return (int)given;
        if (const char *e = lab_env(env_name)) { const int v = atoi(e); if (v >= lo && v <= hi) return v; }
        return -1;
    };
    const int opt_level = choice(opt ? opt->top_grid_level : -1, "SDFHIP_TOP_GRID_LEVEL", 0, 10);
    const int opt_split = choice(opt ? opt->top_grid_split : -1, "SDFHIP_TOP_GRID_SPLIT", 0, 8);
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
    if (!g.ok) return (void)hipGetLastError(), fail(SDFHIP_ERR_DEVICE, "scene_upload: hipSetDevice(%d) failed", device);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SDFHIP_ERR_DEVICE, "scene_upload: device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);

    sdfhip_scene *s = new (std::nothrow) sdfhip_scene();
    if (!s) return fail(SDFHIP_ERR_NOMEM, "scene_upload: out of host memory");
    s->device = device; s->n = n;                                       // (depth and stack_ok are set once the tree has been validated, below)
    s->total_mem = prop.totalGlobalMem;
    s->cu_count = prop.multiProcessorCount;
    s->opt_scatter_grid = opt ? opt->scatter_grid : -1;
    s->opt_scatter_order = opt ? opt->scatter_order : -1;

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
    if ((e = device_alloc(&s->alloc, (size_t)n * 16 + 128)) != hipSuccess) return bail(e, "device_alloc(records)");
    s->nodes = reinterpret_cast<NodeRec *>(static_cast<char *>(s->alloc) + 112);
    if ((e = device_alloc((void **)&s->d_verdict, 2 * sizeof(uint32_t))) != hipSuccess) return bail(e, "device_alloc(verdict)");
    if (resident) {
        d_s = const_cast<int32_t *>(structs); d_v = const_cast<uint8_t *>(values);
    } else {
        if ((e = device_alloc(&d_s, bytes)) != hipSuccess) return bail(e, "device_alloc(structs)");
        if ((e = device_alloc(&d_v, bytes)) != hipSuccess) return bail(e, "device_alloc(values)");
        if ((e = hipMemcpyAsync(d_s, structs, bytes, hipMemcpyHostToDevice, s->stream)) != hipSuccess) return bail(e, "hipMemcpy(structs)");
        if ((e = hipMemcpyAsync(d_v, values, bytes, hipMemcpyHostToDevice, s->stream)) != hipSuccess) return bail(e, "hipMemcpy(values)");
    }
    //  the laboratory library does not take the builder's word: it validates the builder's tree like any other and
    // refuses a scene whose verdict differs from what the builder said -- every GPU test of the builder runs on both flavours, so
    // a regression there is SDFHIP_ERR_BAD_TREE in the test suite, not wrong pixels behind a grid sized for another depth
#ifdef SDFHIP_EXPERIMENTS
    const bool trust = false;
#else
    const bool trust = resident && trusted_depth >= 0;
#endif
    if (trust) {
        // the point-cloud builder's own tree (sdfhip_sdfgen_scene): every block of eight children was appended by k_emit under its
        // parent, the depth is the number of levels it built -- nothing to find out (0.3 ms of a 10 ms build, and a wait on the stream)
        consistent = 1;
        depth = (uint32_t)trusted_depth;
        s->depth = depth;
        s->stack_ok = depth <= (uint32_t)LM ? 1 : 0;
    } else {   // validation (sdfhip_octdata_validate's verdicts, on the device: see k_validate)
        uint32_t *d_verdict = s->d_verdict, verdict[2] = { 0u, 0u };
        if ((e = hipMemsetAsync(d_verdict, 0, sizeof verdict, s->stream)) != hipSuccess) return bail(e, "hipMemset(verdict)");
        // (scratch: the allocation of the fused records, 16 bytes per node, not written before k_fuse)
        Jump *ja = reinterpret_cast<Jump *>(s->alloc), *jb = ja + n;
        const dim3 vg((n + 255u) / 256u), vb(256);
        hipLaunchKernelGGL(k_validate_init, vg, vb, 0, s->stream, (const int2 *)d_s, n, d_verdict, ja);
        for (int round = 0; round < 3; round++) { hipLaunchKernelGGL((k_validate_jump<false>), vg, vb, 0, s->stream, ja, jb, n, d_verdict); std::swap(ja, jb); }
        hipLaunchKernelGGL((k_validate_jump<true>), vg, vb, 0, s->stream, ja, jb, n, d_verdict);
        std::swap(ja, jb);
        if ((e = hipGetLastError()) != hipSuccess) return bail(e, "k_validate launch");
        if ((e = hipMemcpyAsync(verdict, d_verdict, sizeof verdict, hipMemcpyDeviceToHost, s->stream)) != hipSuccess) return bail(e, "hipMemcpy(verdict)");
        if ((e = hipStreamSynchronize(s->stream)) != hipSuccess) return bail(e, "k_validate");
        if ((verdict[0] & 4u) && !(verdict[0] & 1u)) {
            // chains of more than 16 links: three more rounds reach 128; what is open after those is a cycle or longer than 64 links
            if ((e = hipMemsetAsync(d_verdict, 0, sizeof verdict, s->stream)) != hipSuccess) return bail(e, "hipMemset(verdict)");
            const uint32_t keep = verdict[0] & 3u;
            for (int round = 0; round < 2; round++) { hipLaunchKernelGGL((k_validate_jump<false>), vg, vb, 0, s->stream, ja, jb, n, d_verdict); std::swap(ja, jb); }
            hipLaunchKernelGGL((k_validate_jump<true>), vg, vb, 0, s->stream, ja, jb, n, d_verdict);
            if ((e = hipGetLastError()) != hipSuccess) return bail(e, "k_validate launch");
            if ((e = hipMemcpyAsync(verdict, d_verdict, sizeof verdict, hipMemcpyDeviceToHost, s->stream)) != hipSuccess) return bail(e, "hipMemcpy(verdict)");
            if ((e = hipStreamSynchronize(s->stream)) != hipSuccess) return bail(e, "k_validate");
            verdict[0] = keep | (verdict[0] & 1u) | ((verdict[0] & 4u) ? 1u : 0u);      // still open: endless or too long
        }
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
        if (resident && trusted_depth >= 0 && (!consistent || depth != (uint32_t)trusted_depth)) {
            sdfhip_scene_free(s);
            return fail(SDFHIP_ERR_BAD_TREE, "scene_from_arrays: the builder says its tree is consistent and %d levels deep; the validation finds it %s, depth %u",
                        trusted_depth, consistent ? "consistent" : "INCONSISTENT", depth);
        }
        s->depth = depth;
        s->stack_ok = (consistent && depth <= (uint32_t)LM) ? 1 : 0;   // LM: the shader's own descent limit (Compute.hlsl:98)
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
    const bool dense_asked = opt_level >= (int)depth && dense_bytes <= prop.totalGlobalMem / FULL_GRID_SHARE;
    if (depth >= 1 && depth <= 10 && (dense_bytes <= DENSE_GRID_MAX_BYTES || dense_asked)) {
        top_level = (int)depth;
    } else {
        const size_t budget = (size_t)n * 16 > ((size_t)1 << 16) ? (size_t)n * 16 : ((size_t)1 << 16);
        while (top_level < MAX_TOP_LEVEL && (uint32_t)top_level < depth &&
               (sizeof(TopCell) << (3 * (top_level + 1))) <= budget)
            top_level++;
    }
    if (opt_level >= 0) top_level = opt_level < (int)depth ? opt_level : (int)depth;
    // Split grid for trees too deep for a dense grid of their depth (10-12 levels): a dense coarse level C whose
    // internal cells point at dense blocks of the remaining FB = depth - C levels.  Every leaf is one or two
    // loads away (CursorF kernels), the coarse level stays cache-resident, and the blocks exist only where the
    // tree is deep.  Taken when the blocks fit 1/16 of the device's memory; sdfhip_upload_options.top_grid_split
    // forces a coarse level (0 = never).
    int split = 0;
    if ((uint32_t)top_level < depth && depth <= (uint32_t)LM) {
        // coarse level: as deep as 8, no larger than the tree's own records, leaving at most 4 levels to the blocks
        const size_t budget = (size_t)n * 16 > ((size_t)1 << 16) ? (size_t)n * 16 : ((size_t)1 << 16);
        int C = 0;
        while (C < MAX_TOP_LEVEL && C + 1 < (int)depth && (sizeof(TopCell) << (3 * (C + 1))) <= budget) C++;
        if (C >= 1 && (int)depth - C <= 4) split = C;
    }
    if (opt_split >= 0) split = (opt_split >= 1 && opt_split < (int)depth && (int)depth - opt_split <= 6 && opt_split <= 8) ? opt_split : 0;
    if (opt_level >= 0) split = opt_split >= 0 ? split : 0;       // an explicit level means a plain grid
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
        while (top_level > 0 && device_alloc((void **)&s->d_top, sizeof(TopCell) << (3 * top_level)) != hipSuccess) {
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
#ifdef SDFHIP_EXPERIMENTS
    build_dense4(s);
#endif
    if (!resident) { (void)hipFree(d_s); (void)hipFree(d_v); }
    d_s = nullptr; d_v = nullptr;
    *out = s;
    return SDFHIP_OK;
}

extern "C" int sdfhip_scene_top_grid(const sdfhip_scene *s, int32_t *level, uint64_t *bytes)
try {
    if (!s) return fail(SDFHIP_ERR_ARG, "scene_top_grid: null scene");
    if (level) *level = s->d_top ? s->top_level : 0;
    uint64_t extra = s->top2_bytes;
#ifdef SDFHIP_EXPERIMENTS
    extra += s->d4_bytes;
#endif
    if (bytes) *bytes = s->d_top ? ((uint64_t)sizeof(TopCell) << (3 * s->top_level)) + s->fine_bytes + extra : 0;
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_scene_top_grid)

bool sdfhip::scene_has_full_depth_grid(const sdfhip_scene *s)
{
    return s && s->stack_ok && s->d_top && (s->fine_bits || (uint32_t)s->top_level >= s->depth);     // render_impl's `two`
}

extern "C" int sdfhip_scene_info(const sdfhip_scene *s, uint32_t *n, uint32_t *depth,
                                 int *stack_kernel_ok, int *device)
try {
    if (!s) return fail(SDFHIP_ERR_ARG, "scene_info: null scene");
    if (n) *n = s->n;
    if (depth) *depth = s->depth;
    if (stack_kernel_ok) *stack_kernel_ok = s->stack_ok;
    if (device) *device = s->device;
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_scene_info)

// Beside the scene's own grid, the bounce levels of the path-traced pipeline read a split grid of the same cells with larger,
// sub-cube-ordered blocks -- unless the scene's grid already has that coarse level.  Built once: by
// sdfhip_scene_prepare_path (at load time: allocations and two stream synchronisations), or else in front of the first
// path-traced render, before its clock starts.  SDFHIP_SCATTER_GRID=0 turns it off, 1..4 sets the blocks' levels (default 4, 3 for shallow trees);
// SDFHIP_SCATTER_ORDER=0 stores the blocks in x-y-z order.  Without memory for it (1/32 of the device's) the bounce levels
// read the scene's own grid.
void sdfhip::ensure_scatter_grid(sdfhip_scene *s)
{
    if (s->scatter_tried) return;
    s->scatter_tried = 1;
    if (!s->stack_ok || !s->d_top || !((s->fine_bits && s->d_fine) || (uint32_t)s->top_level >= s->depth)) return;   // the pipeline needs a full-depth grid
    // blocks of 8^FB fine cells; 0 = off.  Default 16^3-cell blocks (64 KB each) for trees of depth 6 and more: cfg-5 21.9 ms
    // against 22.3 with 8^3 (and 24.6 with 4^3) for 1.00 instead of 0.83 GB at depth 9, on a 288 GB device
    // (unless a size was asked for, blocks that do not fit the memory share fall back to the next smaller size)
    int asked = s->opt_scatter_grid, order = s->opt_scatter_order;
    if (asked < 0) if (const char *e = lab_env("SDFHIP_SCATTER_GRID")) asked = atoi(e);
    if (order < 0) if (const char *e = lab_env("SDFHIP_SCATTER_ORDER")) order = atoi(e);
    for (int FB = asked >= 0 ? asked : ((int)s->depth >= 6 ? 4 : 3); FB >= 1; FB = asked >= 0 ? 0 : FB - 1) {
        if (!(FB <= 4 && (int)s->depth - FB >= 1 && (int)s->depth - FB <= MAX_TOP_LEVEL)) continue;
        if (s->fine_bits && s->top_level == (int)s->depth - FB) return;            // the scene's own grid is that grid
        uint64_t fbytes = 0;
        s->fine2_order = (FB >= 2 && order != 0) ? 1 : 0;
        if (build_split_grid(s, (int)s->depth - FB, FB, s->fine2_order, s->total_mem / 32, &s->d_top2, &s->d_fine2, &fbytes)) {
            s->top2_level = (int)s->depth - FB; s->fine2_bits = FB;
            s->top2_bytes = ((uint64_t)sizeof(TopCell) << (3 * s->top2_level)) + fbytes;
            return;
        }
    }
}

extern "C" int sdfhip_scene_prepare_path(sdfhip_scene *s)
try {
    if (!s) return fail(SDFHIP_ERR_ARG, "scene_prepare_path: null scene");
    std::lock_guard<std::mutex> lk(s->lock);
    DeviceGuard g(s->device);
    if (!g.ok) return (void)hipGetLastError(), fail(SDFHIP_ERR_DEVICE, "scene_prepare_path: hipSetDevice(%d) failed", s->device);
    ensure_scatter_grid(s);
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_scene_prepare_path)

// The scratch of stream `st` on this scene, with room for `records` hit records (0: control words only).
// Created on a stream's first render; grown (after the stream has drained) when a larger frame comes.
int sdfhip::get_scratch(sdfhip_scene *s, hipStream_t st, size_t records, sdfhip_scene::Scratch **out)
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
        sc->stream = st; sc->launches = 0; sc->ord_valid = false; memset(sc->ord_sig, 0, sizeof sc->ord_sig); sc->band_n = 0;
        HIP_TRY(hipMemsetAsync(sc->ctl, 0, sdfhip_scene::CTL_BYTES, st));
    }
    if (!sc) {
        sc = &s->scratch[s->n_scratch];
        *sc = sdfhip_scene::Scratch();
        sc->stream = st;
        HIP_TRY(hipEventCreateWithFlags(&sc->idle, hipEventDisableTiming));
        { const hipError_t em = device_alloc((void **)&sc->ctl, sdfhip_scene::CTL_BYTES); if (em != hipSuccess) { (void)hipEventDestroy(sc->idle); sc->idle = nullptr; return fail(SDFHIP_ERR_DEVICE, "device_alloc(scratch) failed: %s", hipGetErrorString(em)); } }
        // zeroed IN the stream that will use it: a hipMemset on the null stream is not ordered against a non-blocking
        // stream (the first frame on a new scratch would, now and then, have met counters that were not zero yet)
        s->n_scratch++;
        HIP_TRY(hipMemsetAsync(sc->ctl, 0, sdfhip_scene::CTL_BYTES, st));
    }
    sc->last_use = ++s->uses;
    if (records > sc->records) {
        HIP_TRY(hipStreamSynchronize(st));
        if (sc->hit_buf) { (void)hipFree(sc->hit_buf); sc->hit_buf = nullptr; sc->records = 0; }
        HIP_TRY(device_alloc((void **)&sc->hit_buf, records * 64));
        sc->records = records;
    }
    *out = sc;
    return SDFHIP_OK;
}

// the path-traced pipeline's buffers on a stream's scratch
int sdfhip::get_pt_scratch(sdfhip_scene *s, hipStream_t st, size_t bytes, sdfhip_scene::Scratch **out)
{
    int rc = get_scratch(s, st, 0, out);
    if (rc != SDFHIP_OK) return rc;
    sdfhip_scene::Scratch *sc = *out;
    if (bytes > sc->pt_bytes) {
        HIP_TRY(hipStreamSynchronize(st));
        if (sc->pt_buf) { (void)hipFree(sc->pt_buf); sc->pt_buf = nullptr; sc->pt_bytes = 0; }
        HIP_TRY(device_alloc((void **)&sc->pt_buf, bytes));
        sc->pt_bytes = bytes;
    }
    return SDFHIP_OK;
}

