// Point cloud -> ASDF on the GPU: SdfGen's builder (SdfGen/dllmain.cpp:67-207,295-319),
// level-synchronous instead of recursive, one wavefront per octree node.
//
// What the reference does per node (construct, dllmain.cpp:163-190), and where it is here:
//   centerValue = distance from the cell centre to the nearest candidate   k_center
//   possible    = candidates within (centerValue + sqrt(3)/2 * scale)       k_center (count),
//                 of the centre, in list order                              k_corners (stable compaction)
//   corner i    = signed distance to the nearest *surviving* point, unless  k_corners
//                 inherited from the parent (child i inherits corner i)
//   split       = centerValue < 2 * scale && depth < MaxDepth               k_corners
//   children    = 8 new nodes whose candidates are this node's `possible`   k_children
// The recursion becomes a loop over levels; the reference's node order (children blocks
// appended in depth-first pre-order) is restored on the host afterwards, and the values
// are quantised there with FromFloat / WriteBytes (dllmain.cpp:192-207).
//
// Bit-exactness with oracle/sdfgen_oracle.c (the CPU restatement): same fp32 expressions
// in the same order, no contraction (-ffp-contract=off), IEEE sqrt and divide; ties in the
// nearest-point search go to the earliest list position, as the reference's strict `<`
// scan does; `minDistance < 0.015` is a double comparison, as there.
#include "sdfhip_internal.h"

#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

namespace sdfhip {

struct GenParams {
    const float *verts;        // n x {pos xyz, normal xyz}
    float gs, gox, goy, goz;   // GlobalScale, GlobalOffset
    float scale;               // 2^-depth of this level
    int depth, max_depth;
};

struct LevelArrays {           // one entry per node of the level
    float *px, *py, *pz;       // cell lower corner (unit-cube coordinates)
    float *inherit;            // value of the inherited corner (slot), unused for the root
    int32_t *slot;             // which child of its parent (-1: root)
    int32_t *parent;           // parent's index in its level
    uint32_t *cand_off, *cand_cnt;
    // results
    float *center_value;
    uint32_t *pcount;          // size of `possible`
    float *vals;               // 8 per node
    uint32_t *split;           // 0 / 1
};

__device__ __forceinline__ void transform(const GenParams &P, float wx, float wy, float wz, float &x, float &y, float &z)
{
    wy = 1 - wy;                                         // Transform, dllmain.cpp:82-87
    x = (wx - .5f) * P.gs + P.gox;
    y = (wy - .5f) * P.gs + P.goy;
    z = (wz - .5f) * P.gs + P.goz;
}
__device__ __forceinline__ float lensq(float x, float y, float z) { return x * x + y * y + z * z; }

struct Best { float d; uint32_t k; };
__device__ __forceinline__ bool better(const Best &a, const Best &b) { return a.d < b.d || (a.d == b.d && a.k < b.k); }
__device__ __forceinline__ Best wave_min(Best b)
{
    for (int off = 32; off > 0; off >>= 1) {
        Best o;
        o.d = __shfl_xor(b.d, off);
        o.k = (uint32_t)__shfl_xor((int)b.k, off);
        if (better(o, b)) b = o;
    }
    return b;
}

// centerValue and |possible| of every node of the level (TrueDistanceAt + the count of GetPossible)
__global__ __launch_bounds__(64) void k_center(GenParams P, LevelArrays L, const uint32_t *__restrict__ cand,
                                               uint32_t n_nodes, uint32_t *err)
{
    const uint32_t node = blockIdx.x, lane = threadIdx.x;
    if (node >= n_nodes) return;
    const float h = 0.5f * P.scale;                      // Vector3(1) * 0.5f * scale
    float cx, cy, cz;
    transform(P, L.px[node] + h, L.py[node] + h, L.pz[node] + h, cx, cy, cz);
    const uint32_t off = L.cand_off[node], cnt = L.cand_cnt[node];
    Best b{INFINITY, 0xFFFFFFFFu};
    for (uint32_t k = lane; k < cnt; k += 64) {
        const float *v = P.verts + 6 * (size_t)cand[off + k];
        float d = lensq(v[0] - cx, v[1] - cy, v[2] - cz);
        if (d < b.d) { b.d = d; b.k = k; }
    }
    b = wave_min(b);
    if (b.k == 0xFFFFFFFFu || isinf(b.d) || isnan(b.d)) {    // "Did not find" / "NaN distance"
        if (lane == 0) atomicExch(err, 2u);
        if (lane == 0) { L.center_value[node] = 0.0f; L.pcount[node] = 0; }
        return;
    }
    const float center_value = sqrtf(b.d) / P.gs;
    float r = center_value + 0.866025404f * P.scale;     // GetPossible, dllmain.cpp:151-162
    r *= P.gs;
    r *= r;
    uint32_t count = 0;
    for (uint32_t k = lane; k < cnt; k += 64) {
        const float *v = P.verts + 6 * (size_t)cand[off + k];
        if (lensq(v[0] - cx, v[1] - cy, v[2] - cz) < r) count++;
    }
    for (int o = 32; o > 0; o >>= 1) count += __shfl_xor(count, o);
    if (lane == 0) { L.center_value[node] = center_value; L.pcount[node] = count; }
}

// `possible` list (stable), the corner values, the split decision
__global__ __launch_bounds__(64) void k_corners(GenParams P, LevelArrays L, const uint32_t *__restrict__ cand,
                                                const uint32_t *__restrict__ poff, uint32_t *__restrict__ possible,
                                                uint32_t n_nodes, uint32_t *err)
{
    const uint32_t node = blockIdx.x, lane = threadIdx.x;
    if (node >= n_nodes) return;
    const float px = L.px[node], py = L.py[node], pz = L.pz[node];
    const float h = 0.5f * P.scale;
    float cx, cy, cz;
    transform(P, px + h, py + h, pz + h, cx, cy, cz);
    const float center_value = L.center_value[node];
    float r = center_value + 0.866025404f * P.scale;
    r *= P.gs;
    r *= r;
    // the 8 corner positions, transformed: pos + split(i) * scale
    float qx[8], qy[8], qz[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
        transform(P, px + (float)(i % 2) * P.scale, py + (float)((i / 2) % 2) * P.scale,
                  pz + (float)((i / 2 / 2) % 2) * P.scale, qx[i], qy[i], qz[i]);
    Best best[8];
#pragma unroll
    for (int i = 0; i < 8; i++) best[i] = Best{INFINITY, 0xFFFFFFFFu};
    const uint32_t off = L.cand_off[node], cnt = L.cand_cnt[node], out = poff[node];
    uint32_t base = 0;
    for (uint32_t k0 = 0; k0 < cnt; k0 += 64) {
        const uint32_t k = k0 + lane;
        bool keep = false;
        uint32_t vi = 0;
        float vx = 0, vy = 0, vz = 0;
        if (k < cnt) {
            vi = cand[off + k];
            const float *v = P.verts + 6 * (size_t)vi;
            vx = v[0]; vy = v[1]; vz = v[2];
            keep = lensq(vx - cx, vy - cy, vz - cz) < r;
        }
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            possible[out + base + rank] = vi;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float d = lensq(vx - qx[i], vy - qy[i], vz - qz[i]);
                if (d < best[i].d) { best[i].d = d; best[i].k = k; }
            }
        }
        base += (uint32_t)__popcll(m);
    }
    const int slot = L.slot[node];
    float myval = 0.0f;                                  // lane i < 8 ends up holding corner i
#pragma unroll
    for (int i = 0; i < 8; i++) {
        Best b = wave_min(best[i]);
        float val;
        if (i == slot) {
            val = L.inherit[node];                       // n[i] = vals[insert][i], dllmain.cpp:181
        } else if (b.k == 0xFFFFFFFFu || isinf(b.d) || isnan(b.d)) {
            if (lane == 0) atomicExch(err, 2u);
            val = 0.0f;
        } else {                                         // DistanceAt, dllmain.cpp:119-149
            const float *v = P.verts + 6 * (size_t)cand[off + b.k];
            float md = sqrtf(b.d);
            const float ex = qx[i] - v[0], ey = qy[i] - v[1], ez = qz[i] - v[2];      // p - closest.Position
            if ((double)md < 0.015) {
                const float nl = sqrtf(lensq(v[3], v[4], v[5]));
                md = (v[3] / nl) * ex + (v[4] / nl) * ey + (v[5] / nl) * ez;
            } else if (v[3] * (v[0] - qx[i]) + v[4] * (v[1] - qy[i]) + v[5] * (v[2] - qz[i]) > 0) {   // Inside
                md *= -1;
            }
            val = md / P.gs;
        }
        if ((int)lane == i) myval = val;
    }
    if (lane < 8) L.vals[8 * (size_t)node + lane] = myval;
    if (lane == 0) L.split[node] = (center_value < P.scale * 2 && P.depth < P.max_depth) ? 1u : 0u;
}

// the 8 children of every split node (construct's push_back loop + the arguments of its recursion)
__global__ void k_children(LevelArrays L, LevelArrays N, const uint32_t *__restrict__ block_of,
                           const uint32_t *__restrict__ poff, float half_scale, uint32_t n_nodes)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t node = t >> 3, i = t & 7u;
    if (node >= n_nodes || !L.split[node]) return;
    const uint32_t c = 8 * block_of[node] + i;
    N.px[c] = L.px[node] + (float)(i % 2) * half_scale;
    N.py[c] = L.py[node] + (float)((i / 2) % 2) * half_scale;
    N.pz[c] = L.pz[node] + (float)((i / 2 / 2) % 2) * half_scale;
    N.inherit[c] = L.vals[8 * (size_t)node + i];
    N.slot[c] = (int32_t)i;
    N.parent[c] = (int32_t)node;
    N.cand_off[c] = poff[node];
    N.cand_cnt[c] = L.pcount[node];
}

__global__ void k_iota(uint32_t *p, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i;
}

// exclusive scan of n uint32 (one workgroup walks the array in 1024-element chunks); total -> *total
__global__ __launch_bounds__(1024) void k_scan(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                uint32_t n, unsigned long long *total)
{
    __shared__ uint32_t wsum[16];
    __shared__ unsigned long long carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + tid;
        uint32_t v = i < n ? in[i] : 0u, x = v;
        for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o); if ((int)lane >= o) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; w++) woff += wsum[w];
        const unsigned long long c = carry;
        if (i < n) out[i] = (uint32_t)(c + woff + x - v);
        __syncthreads();
        if (tid == 1023) carry = c + woff + x;
        __syncthreads();
    }
    if (tid == 0) *total = carry;
}

}  // namespace sdfhip

using namespace sdfhip;

namespace {

struct DevBuf {                  // frees what it owns
    std::vector<void *> ptrs;
    template <class T> T *alloc(size_t n)
    {
        void *p = nullptr;
        if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return (T *)p;
    }
    void release(void *p)
    {
        for (auto &q : ptrs) if (q == p) { (void)hipFree(q); q = nullptr; }
    }
    ~DevBuf() { for (void *p : ptrs) if (p) (void)hipFree(p); }
};

bool alloc_level(DevBuf &d, LevelArrays &L, size_t n)
{
    L.px = d.alloc<float>(n); L.py = d.alloc<float>(n); L.pz = d.alloc<float>(n);
    L.inherit = d.alloc<float>(n); L.slot = d.alloc<int32_t>(n); L.parent = d.alloc<int32_t>(n);
    L.cand_off = d.alloc<uint32_t>(n); L.cand_cnt = d.alloc<uint32_t>(n);
    L.center_value = d.alloc<float>(n); L.pcount = d.alloc<uint32_t>(n);
    L.vals = d.alloc<float>(8 * n); L.split = d.alloc<uint32_t>(n);
    return L.px && L.py && L.pz && L.inherit && L.slot && L.parent && L.cand_off && L.cand_cnt &&
           L.center_value && L.pcount && L.vals && L.split;
}
void free_level(DevBuf &d, LevelArrays &L)
{
    void *all[] = { L.px, L.py, L.pz, L.inherit, L.slot, L.parent, L.cand_off, L.cand_cnt, L.center_value,
                    L.pcount, L.vals, L.split };
    for (void *p : all) d.release(p);
}

float saturate(float x) { return x > 1 ? 1 : (x < 0 ? 0 : x); }
uint8_t from_float(float f, float scale)               // dllmain.cpp:192-196
{
    float normd = f / 2 / scale;
    return (uint8_t)floorf(saturate(normd + 0.25f) * 255);
}

struct HostLevel { std::vector<int32_t> parent; std::vector<uint32_t> split; std::vector<float> vals; };

}  // namespace

#define GEN_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(SDFHIP_ERR_DEVICE, "sdfgen: %s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" int sdfhip_sdfgen(int device, const float *verts6, uint32_t n, int32_t depth, sdfhip_octdata *out,
                             sdfhip_sdfgen_stats *stats)
{
    if (!verts6 || !out || n == 0) return fail(SDFHIP_ERR_ARG, "sdfgen: null argument or empty point cloud");
    if (depth < 0 || depth > 12) return fail(SDFHIP_ERR_ARG, "sdfgen: depth %d outside 0..12", depth);
    out->length = 0; out->structs = nullptr; out->values = nullptr;
    auto t0 = std::chrono::steady_clock::now();

    // FindDimensions, dllmain.cpp:67-80 (host: one pass over the points)
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (uint32_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            lo[k] = fminf(lo[k], verts6[6 * (size_t)i + k]);
            hi[k] = fmaxf(hi[k], verts6[6 * (size_t)i + k]);
        }
    GenParams P;
    P.gox = (lo[0] + hi[0]) * 0.5f + 0.003f;
    P.goy = (lo[1] + hi[1]) * 0.5f + 0.003f;
    P.goz = (lo[2] + hi[2]) * 0.5f + 0.003f;
    const float lowest = fminf(lo[0], fminf(lo[1], lo[2])), highest = fmaxf(hi[0], fmaxf(hi[1], hi[2]));
    P.gs = (highest - lowest) * 1.1f;
    P.max_depth = depth;

    int ndev = 0;
    GEN_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(SDFHIP_ERR_DEVICE, "sdfgen: device %d of %d does not exist", device, ndev);
    int prev = -1;
    (void)hipGetDevice(&prev);
    GEN_TRY(hipSetDevice(device));
    struct Restore { int prev; ~Restore() { if (prev >= 0) (void)hipSetDevice(prev); } } restore{prev};

    try {
        DevBuf d;
        float *d_verts = d.alloc<float>(6 * (size_t)n);
        uint32_t *d_err = d.alloc<uint32_t>(1);
        unsigned long long *d_total = d.alloc<unsigned long long>(1);
        uint32_t *cand = d.alloc<uint32_t>(n);
        if (!d_verts || !d_err || !d_total || !cand) return fail(SDFHIP_ERR_NOMEM, "sdfgen: out of device memory");
        GEN_TRY(hipMemcpy(d_verts, verts6, 6 * (size_t)n * sizeof(float), hipMemcpyHostToDevice));
        GEN_TRY(hipMemset(d_err, 0, sizeof(uint32_t)));
        hipLaunchKernelGGL(k_iota, dim3(1024), dim3(256), 0, 0, cand, n);
        P.verts = d_verts;

        LevelArrays L;
        if (!alloc_level(d, L, 1)) return fail(SDFHIP_ERR_NOMEM, "sdfgen: out of device memory");
        {   // the root: construct(all, 0, 0, -1, 0)
            const float z = 0.0f; const int32_t m1 = -1; const uint32_t zero = 0;
            GEN_TRY(hipMemcpy(L.px, &z, 4, hipMemcpyHostToDevice)); GEN_TRY(hipMemcpy(L.py, &z, 4, hipMemcpyHostToDevice));
            GEN_TRY(hipMemcpy(L.pz, &z, 4, hipMemcpyHostToDevice)); GEN_TRY(hipMemcpy(L.inherit, &z, 4, hipMemcpyHostToDevice));
            GEN_TRY(hipMemcpy(L.slot, &m1, 4, hipMemcpyHostToDevice)); GEN_TRY(hipMemcpy(L.parent, &m1, 4, hipMemcpyHostToDevice));
            GEN_TRY(hipMemcpy(L.cand_off, &zero, 4, hipMemcpyHostToDevice)); GEN_TRY(hipMemcpy(L.cand_cnt, &n, 4, hipMemcpyHostToDevice));
        }
        std::vector<HostLevel> levels;
        uint32_t n_nodes = 1;
        unsigned long long cand_entries = n;
        for (int lvl = 0;; lvl++) {
            P.depth = lvl;
            P.scale = ldexpf(1.0f, -lvl);                 // powf(0.5, depth)
            uint32_t *poff = d.alloc<uint32_t>(n_nodes), *block_of = d.alloc<uint32_t>(n_nodes);
            if (!poff || !block_of) return fail(SDFHIP_ERR_NOMEM, "sdfgen: out of device memory");
            hipLaunchKernelGGL(k_center, dim3(n_nodes), dim3(64), 0, 0, P, L, cand, n_nodes, d_err);
            hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, 0, L.pcount, poff, n_nodes, d_total);
            unsigned long long total = 0;
            GEN_TRY(hipMemcpy(&total, d_total, sizeof total, hipMemcpyDeviceToHost));
            if (total > 0xFFFFFFF0ull) return fail(SDFHIP_ERR_NOMEM, "sdfgen: candidate lists of level %d exceed 2^32 entries", lvl);
            uint32_t *possible = d.alloc<uint32_t>((size_t)total);
            if (!possible) return fail(SDFHIP_ERR_NOMEM, "sdfgen: out of device memory for %llu candidate entries", total);
            cand_entries += total;
            hipLaunchKernelGGL(k_corners, dim3(n_nodes), dim3(64), 0, 0, P, L, cand, poff, possible, n_nodes, d_err);
            hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, 0, L.split, block_of, n_nodes, d_total);
            unsigned long long n_split = 0;
            GEN_TRY(hipMemcpy(&n_split, d_total, sizeof n_split, hipMemcpyDeviceToHost));
            uint32_t err = 0;
            GEN_TRY(hipMemcpy(&err, d_err, sizeof err, hipMemcpyDeviceToHost));
            if (err) return fail(SDFHIP_ERR_ARG, "sdfgen: a cell at depth %d has no candidate point left (the reference throws \"Did not find\")", lvl);
            // keep this level's results for the host-side assembly
            HostLevel hl;
            hl.parent.resize(n_nodes); hl.split.resize(n_nodes); hl.vals.resize(8 * (size_t)n_nodes);
            GEN_TRY(hipMemcpy(hl.parent.data(), L.parent, n_nodes * 4, hipMemcpyDeviceToHost));
            GEN_TRY(hipMemcpy(hl.split.data(), L.split, n_nodes * 4, hipMemcpyDeviceToHost));
            GEN_TRY(hipMemcpy(hl.vals.data(), L.vals, 8 * (size_t)n_nodes * 4, hipMemcpyDeviceToHost));
            levels.push_back(std::move(hl));
            if (n_split == 0) break;
            if (8 * n_split > 0x7FFFFFF0ull) return fail(SDFHIP_ERR_NOMEM, "sdfgen: more than 2^31 nodes");
            LevelArrays N;
            if (!alloc_level(d, N, (size_t)(8 * n_split))) return fail(SDFHIP_ERR_NOMEM, "sdfgen: out of device memory");
            hipLaunchKernelGGL(k_children, dim3((8 * n_nodes + 255) / 256), dim3(256), 0, 0, L, N, block_of, poff,
                               P.scale / 2, n_nodes);
            GEN_TRY(hipDeviceSynchronize());
            free_level(d, L);
            d.release(cand); d.release(poff); d.release(block_of);
            L = N;
            cand = possible;
            n_nodes = (uint32_t)(8 * n_split);
        }
        GEN_TRY(hipDeviceSynchronize());

        // ---- host: the reference's node order and bytes ------------------------------------
        // children block of split node j of level l = nodes 8*b .. 8*b+7 of level l+1, b = rank of j
        // among the split nodes of its level.  The reference appends a node's block when it
        // processes the node and then recurses into the children in order: pre-order.
        size_t total_nodes = 0;
        for (auto &hl : levels) total_nodes += hl.parent.size();
        std::vector<std::vector<uint32_t>> first_child(levels.size());
        for (size_t l = 0; l < levels.size(); l++) {
            first_child[l].assign(levels[l].split.size(), 0xFFFFFFFFu);
            uint32_t b = 0;
            for (size_t j = 0; j < levels[l].split.size(); j++)
                if (levels[l].split[j]) first_child[l][j] = 8 * b++;
        }
        int32_t *S = (int32_t *)malloc(total_nodes * 8);
        uint8_t *V = (uint8_t *)malloc(total_nodes * 8);
        if (!S || !V) { free(S); free(V); return fail(SDFHIP_ERR_NOMEM, "sdfgen: out of host memory for %zu nodes", total_nodes); }
        struct Item { uint32_t level, j; int32_t new_index, new_parent; };
        std::vector<Item> stack;
        stack.push_back(Item{ 0, 0, 0, -1 });
        size_t next = 1;
        while (!stack.empty()) {
            Item it = stack.back(); stack.pop_back();
            const HostLevel &hl = levels[it.level];
            const float scale = ldexpf(1.0f, -(int)it.level);
            for (int k = 0; k < 8; k++) V[(size_t)it.new_index * 8 + k] = from_float(hl.vals[8 * (size_t)it.j + k], scale);
            S[2 * (size_t)it.new_index] = it.new_parent;
            const uint32_t fc = first_child[it.level][it.j];
            if (fc == 0xFFFFFFFFu) { S[2 * (size_t)it.new_index + 1] = -1; continue; }
            const int32_t block = (int32_t)next;
            next += 8;
            S[2 * (size_t)it.new_index + 1] = block;
            // pre-order: child 0's subtree is numbered first -> push the children in reverse
            for (int k = 7; k >= 0; k--) stack.push_back(Item{ it.level + 1, fc + (uint32_t)k, block + k, it.new_index });
        }
        out->length = (uint32_t)total_nodes; out->structs = S; out->values = V;
        if (stats) {
            stats->nodes = (uint32_t)total_nodes;
            stats->levels = (uint32_t)levels.size();
            stats->candidate_entries = cand_entries;
            stats->global_scale = P.gs;
            stats->global_offset[0] = P.gox; stats->global_offset[1] = P.goy; stats->global_offset[2] = P.goz;
            stats->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        return SDFHIP_OK;
    } catch (const std::bad_alloc &) {
        return fail(SDFHIP_ERR_NOMEM, "sdfgen: out of host memory");
    }
}
