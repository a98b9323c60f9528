// Point cloud -> ASDF on the GPU: SdfGen's builder (SdfGen/dllmain.cpp:67-207,295-319),
// level-synchronous instead of recursive: a wavefront per block of eight siblings (which share one candidate list), several
// workgroups per node on the first levels.
//
// What the reference does per node (construct, dllmain.cpp:163-190), and where it is here:
//   centerValue = distance from the cell centre to the nearest candidate   k_center_*
//   possible    = candidates within (centerValue + sqrt(3)/2 * scale)       k_center_* (count),
//                 of the centre, in list order                              k_corners_* (stable compaction)
//   corner i    = signed distance to the nearest *surviving* point, unless  k_corners_*
//                 inherited from the parent (child i inherits corner i)
//   split       = centerValue < 2 * scale && depth < MaxDepth               k_corners_*
// (levels of fewer than 16 384 nodes: k_center_seg_min / _seg_count, k_corners_seg / _fin -- several workgroups per node;
//  the others: k_center_sib / k_corners_sib -- a wavefront per sibling block, the shared list staged through LDS)
//   children    = 8 new nodes whose candidates are this node's `possible`   k_children
// The recursion becomes a loop over levels; the reference's node order (children blocks
// appended in depth-first pre-order) is restored afterwards, still on the GPU, from subtree
// counts (k_subtree, bottom-up) and pre-order ranks (k_emit, top-down), and k_emit also
// quantises the values as FromFloat / WriteBytes do (dllmain.cpp:192-207): the host receives
// the finished {parent, children} and byte arrays in one copy each.
// Device memory comes from three bump arenas (results that live until the end; per-level
// scratch and candidate lists, ping-ponged between levels), not from one hipMalloc per array.
//
// Bit-exactness with oracle/sdfgen_oracle.c (the CPU restatement): same fp32 expressions
// in the same order, no contraction (-ffp-contract=off), IEEE sqrt and divide; ties in the
// nearest-point search go to the earliest list position, as the reference's strict `<`
// scan does; `minDistance < 0.015` is a double comparison, as there.
#include "abi_guard.h"

#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <sys/mman.h>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace sdfhip {

struct GenParams {
    const float *verts;        // n x {pos xyz, normal xyz}
    float gs, gox, goy, goz;   // GlobalScale, GlobalOffset
    float scale;               // 2^-depth of this level
    int depth, max_depth;
};

// A candidate list entry carries its point's position: the passes over a list stream 16-byte entries instead of chasing an
// index into the point array (the lists of the 1 M-point knot: 741 M entries at depth 10, the passes were chains of dependent
// gathers); the index (w, as bits) is needed once per corner, for the winner's normal.
typedef float4 Cand;

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
    uint32_t *block_of;        // rank among the split nodes of the level: children are 8*block_of .. +7 of the next
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

// Reduction over the BT = 1024 threads of a workgroup that works on (a segment of) one node's list.
template <int BT> __device__ __forceinline__ Best node_min(Best b, Best *sh)
{
    b = wave_min(b);
    if constexpr (BT > 64) {
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        __syncthreads();                                 // sh may still be read from the previous use
        if (lane == 0) sh[wave] = b;
        __syncthreads();
        b = sh[0];
        for (int w = 1; w < BT / 64; w++) if (better(sh[w], b)) b = sh[w];
    }
    return b;
}

// ---- several workgroups per node (the first levels) ---------------------------------------------------
// Levels 0 .. 4 have 1 .. 4 095 nodes with up to a million candidates each: one workgroup per node leaves the chip idle (the 1 M-
// point knot: 16 ms of a 40 ms build).  Here a node's list is cut into S segments of whole 1 024-entry chunks, one workgroup
// each; the nearest-candidate searches meet in 64-bit atomic minima of {distance bits, list position} -- for non-negative
// floats the bit patterns order as the values do, and equal distances go to the earlier position: the same winner as the
// strict `<` scan -- and the `possible` list stays in list order: segment s writes behind the survivors of the segments
// before it (their counts are known from the centre pass).
constexpr uint32_t SEG_BT = 256;   // threads of a workgroup of the segment kernels: four wavefronts -- five workgroups share a CU, where one of
                                   // sixteen wavefronts had it alone and nothing covered its chains of dependent loads (10 us a workgroup)
constexpr uint32_t SEG_U = 4;      // entries a thread has in flight per trip: a trip takes SEG_U x SEG_BT = 1 024 list entries
constexpr uint32_t SEG_W = SEG_BT / 64;
static_assert(SEG_U * SEG_BT == 1024, "segment_of cuts lists into whole trips of 1 024 entries");
struct SegArrays {
    unsigned long long *best;      // per node: the candidate nearest to the cell centre
    unsigned long long *corner;    // 8 per node: ... to every corner, among the survivors
    uint32_t *count;               // per node and segment: survivors
    uint32_t S;
};
__device__ __forceinline__ unsigned long long pack_best(const Best &b) { return ((unsigned long long)__float_as_uint(b.d) << 32) | b.k; }
__device__ __forceinline__ Best unpack_best(unsigned long long p) { return Best{__uint_as_float((uint32_t)(p >> 32)), (uint32_t)p}; }
__device__ __forceinline__ void segment_of(uint32_t cnt, uint32_t S, uint32_t s, uint32_t &lo, uint32_t &hi)
{
    const uint32_t len = ((cnt + S - 1u) / S + 1023u) & ~1023u;          // whole chunks
    lo = min(cnt, s * len); hi = min(cnt, lo + len);
}

// Which (node, segment) a workgroup of the segment kernels takes.  Workgroups go to the eight XCDs in turn (blockIdx mod 8), each
// with an L2 of its own, and the eight siblings of a block read the SAME list: numbered node-major, neighbouring nodes land on eight
// different XCDs and every list comes out of HBM up to eight times (measured: three times its bytes).  So: XCD x takes the x-th
// eighth of the work, and within the work the eight siblings' segment s are neighbours -- the same L2, at about the same time.
__device__ __forceinline__ bool seg_work(const SegArrays &A, uint32_t n_nodes, uint32_t &node, uint32_t &s)
{
    const uint32_t per = gridDim.x >> 3;                 // the grid is padded to a multiple of 8
    const uint32_t v = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (v >= n_nodes * A.S) return false;
    if (n_nodes >= 8u) {                                 // below the root: blocks of eight siblings
        const uint32_t w = v >> 3;
        node = (w / A.S) * 8u + (v & 7u); s = w % A.S;
    } else {
        node = v / A.S; s = v % A.S;
    }
    return true;
}

__global__ __launch_bounds__(SEG_BT) void k_center_seg_min(GenParams P, LevelArrays L, SegArrays A, const Cand *__restrict__ cand, uint32_t n_nodes)
{
    __shared__ Best sh[SEG_W];
    const uint32_t tid = threadIdx.x;
    uint32_t node, s;
    if (!seg_work(A, n_nodes, node, s)) return;
    const float h = 0.5f * P.scale;
    float cx, cy, cz;
    transform(P, L.px[node] + h, L.py[node] + h, L.pz[node] + h, cx, cy, cz);
    const uint32_t off = L.cand_off[node], cnt = L.cand_cnt[node];
    uint32_t lo, hi;
    segment_of(cnt, A.S, s, lo, hi);
    Best b{INFINITY, 0xFFFFFFFFu};
    // SEG_U loads in flight per thread (one per trip left the passes waiting for a single load's latency: 1.6 TB/s out of L2),
    // and the next trip's are issued before this trip's entries are looked at
    Cand nx[SEG_U];
    if (lo < hi) {
#pragma unroll
        for (uint32_t u = 0; u < SEG_U; u++) nx[u] = cand[off + min(lo + tid + u * SEG_BT, hi - 1u)];
    }
    for (uint32_t k0 = lo + tid; k0 < hi; k0 += SEG_U * SEG_BT) {
        Cand v[SEG_U];
#pragma unroll
        for (uint32_t u = 0; u < SEG_U; u++) { v[u] = nx[u]; nx[u] = cand[off + min(k0 + (SEG_U + u) * SEG_BT, hi - 1u)]; }
#pragma unroll
        for (uint32_t u = 0; u < SEG_U; u++) {
            const uint32_t k = k0 + u * SEG_BT;
            float d = lensq(v[u].x - cx, v[u].y - cy, v[u].z - cz);
            if (k < hi && d < b.d) { b.d = d; b.k = k; }
        }
    }
    b = node_min<SEG_BT>(b, sh);
    if (tid == 0 && b.k != 0xFFFFFFFFu) atomicMin(&A.best[node], pack_best(b));
}

__global__ __launch_bounds__(SEG_BT) void k_center_seg_count(GenParams P, LevelArrays L, SegArrays A, const Cand *__restrict__ cand,
                                                           uint32_t n_nodes, uint32_t *err)
{
    __shared__ uint32_t shc[SEG_W];
    const uint32_t tid = threadIdx.x;
    uint32_t node, s;
    if (!seg_work(A, n_nodes, node, s)) return;
    const Best b = unpack_best(A.best[node]);
    if (b.k == 0xFFFFFFFFu || isinf(b.d) || isnan(b.d)) {    // "Did not find" / "NaN distance"
        if (tid == 0) {
            A.count[node * A.S + s] = 0;
            if (s == 0) { atomicExch(err, 2u); L.center_value[node] = 0.0f; }
        }
        return;
    }
    const float h = 0.5f * P.scale;
    float cx, cy, cz;
    transform(P, L.px[node] + h, L.py[node] + h, L.pz[node] + h, cx, cy, cz);
    const uint32_t off = L.cand_off[node], cnt = L.cand_cnt[node];
    uint32_t lo, hi;
    segment_of(cnt, A.S, s, lo, hi);
    const float center_value = sqrtf(b.d) / P.gs;
    float r = center_value + 0.866025404f * P.scale;     // GetPossible, dllmain.cpp:151-162
    r *= P.gs;
    r *= r;
    uint32_t count = 0;
    Cand nx[SEG_U];
    if (lo < hi) {
#pragma unroll
        for (uint32_t u = 0; u < SEG_U; u++) nx[u] = cand[off + min(lo + tid + u * SEG_BT, hi - 1u)];
    }
#pragma unroll 1
    for (uint32_t k0 = lo + tid; k0 < hi; k0 += SEG_U * SEG_BT) {
        Cand v[SEG_U];
#pragma unroll
        for (uint32_t u = 0; u < SEG_U; u++) { v[u] = nx[u]; nx[u] = cand[off + min(k0 + (SEG_U + u) * SEG_BT, hi - 1u)]; }
#pragma unroll
        for (uint32_t u = 0; u < SEG_U; u++)
            if (k0 + u * SEG_BT < hi && lensq(v[u].x - cx, v[u].y - cy, v[u].z - cz) < r) count++;
    }
    for (int o = 32; o > 0; o >>= 1) count += __shfl_xor(count, o);
    if ((tid & 63u) == 0) shc[tid >> 6] = count;
    __syncthreads();
    if (tid == 0) {
        count = 0;
        for (uint32_t w = 0; w < SEG_W; w++) count += shc[w];
        A.count[node * A.S + s] = count;
        if (count) atomicAdd(&L.pcount[node], count);
        if (s == 0) L.center_value[node] = center_value;
    }
}

__global__ __launch_bounds__(SEG_BT) void k_corners_seg(GenParams P, LevelArrays L, SegArrays A, const Cand *__restrict__ cand,
                                                      const uint32_t *__restrict__ poff, Cand *__restrict__ possible, uint32_t n_nodes)
{
    __shared__ Best sh[SEG_W * 8];
    __shared__ uint32_t kept[2][SEG_U][SEG_W];              // survivors per wavefront and chunk, double-buffered over trips
    __shared__ uint32_t ahead[SEG_W];
    __shared__ Cand surv[SEG_U * SEG_BT];                // the survivors of a trip ({position, place in the list})
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t node, s;
    if (!seg_work(A, n_nodes, node, s)) return;
    const Best c = unpack_best(A.best[node]);
    if (c.k == 0xFFFFFFFFu || isinf(c.d) || isnan(c.d)) return;          // no centre value: the build fails (k_center_seg_count said so)
    const float px = L.px[node], py = L.py[node], pz = L.pz[node];
    const float h = 0.5f * P.scale;
    float cx, cy, cz;
    transform(P, px + h, py + h, pz + h, cx, cy, cz);
    const float center_value = L.center_value[node];
    float r = center_value + 0.866025404f * P.scale;
    r *= P.gs;
    r *= r;
    float qx[8], qy[8], qz[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
        transform(P, px + (float)(i % 2) * P.scale, py + (float)((i / 2) % 2) * P.scale,
                  pz + (float)((i / 2 / 2) % 2) * P.scale, qx[i], qy[i], qz[i]);
    Best best[8];
#pragma unroll
    for (int i = 0; i < 8; i++) best[i] = Best{INFINITY, 0xFFFFFFFFu};
    const uint32_t off = L.cand_off[node], cnt = L.cand_cnt[node];
    uint32_t lo, hi;
    segment_of(cnt, A.S, s, lo, hi);
    if (lo >= hi) return;                                // (the whole workgroup: a short list leaves most segments empty)
    // where this segment's survivors go: behind those of the segments before it
    uint32_t before_seg = 0;
    for (uint32_t j = tid; j < s; j += SEG_BT) before_seg += A.count[node * A.S + j];
    for (int o = 32; o > 0; o >>= 1) before_seg += __shfl_xor(before_seg, o);
    if (lane == 0) ahead[wave] = before_seg;
    __syncthreads();
    before_seg = 0;
    for (uint32_t w = 0; w < SEG_W; w++) before_seg += ahead[w];
    const bool last = P.depth >= P.max_depth;
    const uint32_t out = last ? 0u : poff[node] + before_seg;
    uint32_t base = 0, flip = 0;
    Cand nx[SEG_U];                                      // the next trip's entries, on their way while this trip's are worked on
#pragma unroll
    for (uint32_t u = 0; u < SEG_U; u++) nx[u] = cand[off + min(lo + u * SEG_BT + tid, hi - 1u)];
    for (uint32_t k0 = lo; k0 < hi; k0 += SEG_U * SEG_BT, flip ^= 1u) {
        // SEG_U entries per thread and trip: their loads are in flight together, and one barrier serves them all
        Cand vi[SEG_U];
        bool keep[SEG_U];
        unsigned long long m[SEG_U];
#pragma unroll
        for (uint32_t u = 0; u < SEG_U; u++) { vi[u] = nx[u]; nx[u] = cand[off + min(k0 + (SEG_U + u) * SEG_BT + tid, hi - 1u)]; }
#pragma unroll
        for (uint32_t u = 0; u < SEG_U; u++) {
            keep[u] = k0 + u * SEG_BT + tid < hi && lensq(vi[u].x - cx, vi[u].y - cy, vi[u].z - cz) < r;
            m[u] = __ballot(keep[u]);
            if (lane == 0) kept[flip][u][wave] = (uint32_t)__popcll(m[u]);
        }
        __syncthreads();                                 // (also: the corner search of the trip before has read surv)
        // The survivors go out in list order -- chunk by chunk, wavefront by wavefront -- and into LDS, from where they are dealt to the
        // threads again, one each, for the eight corner distances: every lane busy, where the distances inside `if (keep)` ran with
        // the fifth of the lanes whose entry had survived (most of the kernel's instructions, once its lists came from one L2).
        uint32_t all = 0;
#pragma unroll
        for (uint32_t u = 0; u < SEG_U; u++) {
            uint32_t before = 0, chunk = 0;
            for (uint32_t w = 0; w < SEG_W; w++) { const uint32_t n = kept[flip][u][w]; before += w < wave ? n : 0u; chunk += n; }
            if (keep[u]) {
                const uint32_t at = all + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(m[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[u], 0u));
                if (!last) possible[out + base + at] = vi[u];
                surv[at] = make_float4(vi[u].x, vi[u].y, vi[u].z, __uint_as_float(k0 + u * SEG_BT + tid));      // w: its place in the list
            }
            all += chunk;
        }
        __syncthreads();
        for (uint32_t at = tid; at < all; at += SEG_BT) {    // (in list order per thread, and `better` orders the threads: the earliest of equals)
            const Cand e = surv[at];
            const uint32_t k = __float_as_uint(e.w);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float d = lensq(e.x - qx[i], e.y - qy[i], e.z - qz[i]);
                if (d < best[i].d) { best[i].d = d; best[i].k = k; }
            }
        }
        base += all;
    }
    // the eight minima of the workgroup with ONE barrier: every wavefront leaves its own in LDS, thread i of the first eight
    // takes corner i over the wavefronts (eight reductions one after the other, two barriers each, were most of what a
    // workgroup with little to do cost: 17 us)
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const Best b = wave_min(best[i]);
        if (lane == 0) sh[wave * 8 + i] = b;
    }
    __syncthreads();
    if (tid < 8u) {
        Best b = sh[tid];
        for (uint32_t w = 1; w < SEG_W; w++) if (better(sh[w * 8 + tid], b)) b = sh[w * 8 + tid];
        if (b.k != 0xFFFFFFFFu) atomicMin(&A.corner[8 * (size_t)node + tid], pack_best(b));
    }
}

// the corner values and the split decision from the segments' minima: thread i of a node's eight, corner i
__global__ __launch_bounds__(256) void k_corners_fin(GenParams P, LevelArrays L, SegArrays A, const Cand *__restrict__ cand,
                                                     uint32_t n_nodes, uint32_t *err)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x, node = g >> 3;
    const int i = (int)(g & 7u);
    if (node >= n_nodes) return;
    const Best c = unpack_best(A.best[node]);
    const bool no_centre = c.k == 0xFFFFFFFFu || isinf(c.d) || isnan(c.d);
    const float px = L.px[node], py = L.py[node], pz = L.pz[node];
    float qx, qy, qz;
    transform(P, px + (float)(i % 2) * P.scale, py + (float)((i / 2) % 2) * P.scale, pz + (float)((i / 2 / 2) % 2) * P.scale, qx, qy, qz);
    const Best b = unpack_best(A.corner[8 * (size_t)node + i]);
    const uint32_t off = L.cand_off[node];
    float val;
    if (i == L.slot[node]) {
        val = L.inherit[node];                           // n[i] = vals[insert][i], dllmain.cpp:181
    } else if (no_centre || b.k == 0xFFFFFFFFu || isinf(b.d) || isnan(b.d)) {
        atomicExch(err, 2u);
        val = 0.0f;
    } else {                                             // DistanceAt, dllmain.cpp:119-149
        const float *v = P.verts + 6 * (size_t)__float_as_uint(cand[off + b.k].w);
        float md = sqrtf(b.d);
        const float ex = qx - v[0], ey = qy - v[1], ez = qz - v[2];      // p - closest.Position
        if ((double)md < 0.015) {
            const float nl = sqrtf(lensq(v[3], v[4], v[5]));
            md = (v[3] / nl) * ex + (v[4] / nl) * ey + (v[5] / nl) * ez;
        } else if (v[3] * (v[0] - qx) + v[4] * (v[1] - qy) + v[5] * (v[2] - qz) > 0) {   // Inside
            md *= -1;
        }
        val = md / P.gs;
    }
    L.vals[8 * (size_t)node + i] = val;
    if (i == 0) L.split[node] = (L.center_value[node] < P.scale * 2 && P.depth < P.max_depth) ? 1u : 0u;
}

// minimum over the G lanes of a group (ties: the earlier list position)
template <int G> __device__ __forceinline__ Best sub_min(Best b)
{
    for (int off = G / 2; off > 0; off >>= 1) {
        Best o;
        o.d = __shfl_xor(b.d, off);
        o.k = (uint32_t)__shfl_xor((int)b.k, off);
        if (better(o, b)) b = o;
    }
    return b;
}

// ---- a wavefront per sibling block --------------------------------------------------------------------
// The eight children of a node share ONE candidate list (construct passes `possible` to all of them, dllmain.cpp:183-189; here
// cand_off / cand_cnt of the eight are equal).  A wavefront takes a whole sibling block: it brings the list into LDS once, 64
// entries per load instruction, fully coalesced and SIB_CH / 64 loads in flight per lane, and eight lanes per child walk it
// there -- where the sixteen-lanes-per-node form asks L1 for every entry once per child and waits for it.  A short list (the
// deepest levels: ~100 entries) stays in LDS for both passes of the centre kernel.  The epilogue is lane-parallel: a butterfly
// that exchanges halves (4 + 2 + 1 candidates per lane instead of 8 x 3) leaves lane j of a child with the nearest survivor of
// corner j, and the lane computes that corner's value alone (the other form computes all eight on every lane).
// Same arithmetic per candidate, same tie-breaking (earliest list position), same stable order of the survivors: same bytes.
constexpr int SIB_CH = 256;                               // list entries per LDS chunk (4 KB per wavefront)
constexpr int SIB_LD = SIB_CH / 64;                       // loads per lane and chunk
// a chunk on its way into the lanes' registers (the next one, while the wavefront works on the one in LDS) ...
struct SibChunk { Cand a, b, c, d; };
static_assert(SIB_LD == 4, "SibChunk holds four entries per lane");
__device__ __forceinline__ SibChunk sib_fetch(const Cand *__restrict__ list, uint32_t k0, uint32_t cnt, uint32_t lane)
{
    const uint32_t last = cnt - 1u, at = k0 < cnt ? k0 + lane : last;                  // (behind the list's end: its last entry, not used)
    return SibChunk{ list[min(at, last)], list[min(at + 64u, last)], list[min(at + 128u, last)], list[min(at + 192u, last)] };
}
// ... and from there into LDS
__device__ __forceinline__ void sib_stage(Cand *buf, const SibChunk &pre, uint32_t lane)
{
    buf[lane] = pre.a; buf[lane + 64u] = pre.b; buf[lane + 128u] = pre.c; buf[lane + 192u] = pre.d;
}

// PIPE: the next chunk is fetched while this one is worked on (lists of several chunks; for the short lists of the deepest levels the
// registers it takes cost more than it brings: 2.7 against 2.2 ms on the last level of the 1 M-point knot)
template <bool PIPE>
__global__ __launch_bounds__(64) void k_center_sib(GenParams P, LevelArrays L, const Cand *__restrict__ cand, uint32_t n_nodes, uint32_t *err)
{
    __shared__ Cand buf[SIB_CH];
    const uint32_t lane = threadIdx.x, j = lane & 7u, node = blockIdx.x * 8u + (lane >> 3);      // n_nodes is a multiple of 8 below the root
    if (blockIdx.x * 8u >= n_nodes) return;
    const float h = 0.5f * P.scale;
    float cx, cy, cz;
    transform(P, L.px[node] + h, L.py[node] + h, L.pz[node] + h, cx, cy, cz);
    const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.cand_off[node]), cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.cand_cnt[node]);
    Best b{INFINITY, 0xFFFFFFFFu};
    SibChunk pre{};
    if (PIPE && cnt) pre = sib_fetch(cand + off, 0, cnt, lane);
    for (uint32_t k0 = 0; k0 < cnt; k0 += SIB_CH) {
        const uint32_t n = min((uint32_t)SIB_CH, cnt - k0);
        if (k0) __syncthreads();                         // the chunk before has been read
        if (PIPE) {
            sib_stage(buf, pre, lane);
            pre = sib_fetch(cand + off, k0 + SIB_CH, cnt, lane);
        } else {
            for (uint32_t u = lane; u < n; u += 64u) buf[u] = cand[off + k0 + u];
        }
        __syncthreads();
        for (uint32_t t = j; t < n; t += 8u) {
            const Cand e = buf[t];
            const float d = lensq(e.x - cx, e.y - cy, e.z - cz);
            if (d < b.d) { b.d = d; b.k = k0 + t; }
        }
    }
    b = sub_min<8>(b);
    const bool bad = b.k == 0xFFFFFFFFu || isinf(b.d) || isnan(b.d);      // "Did not find" / "NaN distance"
    const float center_value = bad ? 0.0f : sqrtf(b.d) / P.gs;
    float r = center_value + 0.866025404f * P.scale;     // GetPossible, dllmain.cpp:151-162
    r *= P.gs;
    r *= r;
    uint32_t count = 0;
    const bool last = P.depth >= P.max_depth;            // no node of the last level splits: nobody will read its survivors' list,
    if (PIPE && cnt > (uint32_t)SIB_CH && !last) pre = sib_fetch(cand + off, 0, cnt, lane);      // so nobody needs its length
    for (uint32_t k0 = 0; k0 < (last ? 0u : cnt); k0 += SIB_CH) {
        const uint32_t n = min((uint32_t)SIB_CH, cnt - k0);
        if (cnt > (uint32_t)SIB_CH) {                    // (a list of one chunk is still there)
            __syncthreads();
            if (PIPE) {
                sib_stage(buf, pre, lane);
                pre = sib_fetch(cand + off, k0 + SIB_CH, cnt, lane);
            } else {
                for (uint32_t u = lane; u < n; u += 64u) buf[u] = cand[off + k0 + u];
            }
            __syncthreads();
        }
        if (!bad)
            for (uint32_t t = j; t < n; t += 8u) {
                const Cand e = buf[t];
                if (lensq(e.x - cx, e.y - cy, e.z - cz) < r) count++;
            }
    }
    for (int o = 4; o > 0; o >>= 1) count += __shfl_xor(count, o);
    if (j == 0) {
        if (bad) atomicExch(err, 2u);
        L.center_value[node] = center_value; L.pcount[node] = bad ? 0u : count;
    }
}

// one step of the epilogue's butterfly: of 2 n candidates a lane keeps the n whose corner has this bit as the lane has it, and
// takes its partner's candidates for the same corners
template <int N>
__device__ __forceinline__ void halve(const Best (&in)[2 * N], Best (&out)[N], bool upper, int off)
{
#pragma unroll
    for (int i = 0; i < N; i++) {
        const Best mine = upper ? in[i + N] : in[i], send = upper ? in[i] : in[i + N];
        Best o;
        o.d = __shfl_xor(send.d, off);
        o.k = (uint32_t)__shfl_xor((int)send.k, off);
        out[i] = better(o, mine) ? o : mine;
    }
}

// FUSED (the last level, whose survivors' lists nobody needs -- so nothing waits for a scan between the two searches): the centre
// search of k_center_sib runs first, in this kernel, and a list of one chunk is read once for both.
template <bool PIPE, bool FUSED = false>
__global__ __launch_bounds__(64) void k_corners_sib(GenParams P, LevelArrays L, const Cand *__restrict__ cand,
                                                    const uint32_t *__restrict__ poff, Cand *__restrict__ possible,
                                                    uint32_t n_nodes, uint32_t *err)
{
    static_assert(!(PIPE && FUSED), "the fused form is for the short lists of the last level");
    __shared__ Cand buf[SIB_CH];
    const uint32_t lane = threadIdx.x, s = lane >> 3, j = lane & 7u, node = blockIdx.x * 8u + s;
    if (blockIdx.x * 8u >= n_nodes) return;
    const float px = L.px[node], py = L.py[node], pz = L.pz[node];
    const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.cand_off[node]), cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.cand_cnt[node]);
    float center_value;
    bool staged = false;                                 // buf holds the list's only chunk
    if (FUSED) {
        const float hh = 0.5f * P.scale;
        float mx, my, mz;
        transform(P, px + hh, py + hh, pz + hh, mx, my, mz);
        Best c{INFINITY, 0xFFFFFFFFu};
        for (uint32_t k0 = 0; k0 < cnt; k0 += SIB_CH) {
            const uint32_t n = min((uint32_t)SIB_CH, cnt - k0);
            if (k0) __syncthreads();
            for (uint32_t u = lane; u < n; u += 64u) buf[u] = cand[off + k0 + u];
            __syncthreads();
            for (uint32_t t = j; t < n; t += 8u) {
                const Cand e = buf[t];
                const float d = lensq(e.x - mx, e.y - my, e.z - mz);
                if (d < c.d) { c.d = d; c.k = k0 + t; }
            }
        }
        c = sub_min<8>(c);
        const bool bad = c.k == 0xFFFFFFFFu || isinf(c.d) || isnan(c.d);      // "Did not find" / "NaN distance"
        center_value = bad ? 0.0f : sqrtf(c.d) / P.gs;
        if (j == 0) {
            if (bad) atomicExch(err, 2u);
            L.center_value[node] = center_value; L.pcount[node] = 0u;
        }
        staged = cnt <= (uint32_t)SIB_CH;
    } else {
        center_value = L.center_value[node];
    }
    const bool last = P.depth >= P.max_depth;            // the last level's survivors are looked at (the corner values), not kept
    const uint32_t out = last ? 0u : poff[node];
    const int slot = L.slot[node];
    const float h = 0.5f * P.scale;
    float cx, cy, cz;
    transform(P, px + h, py + h, pz + h, cx, cy, cz);
    float r = center_value + 0.866025404f * P.scale;
    r *= P.gs;
    r *= r;
    float qx[8], qy[8], qz[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
        transform(P, px + (float)(i % 2) * P.scale, py + (float)((i / 2) % 2) * P.scale,
                  pz + (float)((i / 2 / 2) % 2) * P.scale, qx[i], qy[i], qz[i]);
    Best best[8];
#pragma unroll
    for (int i = 0; i < 8; i++) best[i] = Best{INFINITY, 0xFFFFFFFFu};
    uint32_t base = 0;
    SibChunk pre{};
    if (PIPE && cnt) pre = sib_fetch(cand + off, 0, cnt, lane);
    for (uint32_t k0 = 0; k0 < cnt; k0 += SIB_CH) {
        const uint32_t n = min((uint32_t)SIB_CH, cnt - k0);
        if (!(FUSED && staged)) {
            if (k0 || FUSED) __syncthreads();
            if (PIPE) {
                sib_stage(buf, pre, lane);
                pre = sib_fetch(cand + off, k0 + SIB_CH, cnt, lane);
            } else {
                for (uint32_t u = lane; u < n; u += 64u) buf[u] = cand[off + k0 + u];
            }
            __syncthreads();
        }
        for (uint32_t t0 = 0; t0 < n; t0 += 8u) {        // eight entries per child and round, in list order
            const uint32_t t = t0 + j;
            const Cand vi = buf[t < n ? t : 0u];
            const float vx = vi.x, vy = vi.y, vz = vi.z;
            const bool keep = t < n && lensq(vx - cx, vy - cy, vz - cz) < r;
            const unsigned long long m = __ballot(keep);
            const uint32_t gm = (uint32_t)(m >> (8u * s)) & 0xFFu;                   // this child's survivors of the round
            if (keep) {
                if (!last) possible[out + base + (uint32_t)__popc(gm & ((1u << j) - 1u))] = vi;    // stable: list order
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    float d = lensq(vx - qx[i], vy - qy[i], vz - qz[i]);
                    if (d < best[i].d) { best[i].d = d; best[i].k = k0 + t; }
                }
            }
            base += (uint32_t)__popc(gm);
        }
    }
    // lane j of a child ends up with corner j's nearest survivor
    Best b4[4], b2[2], b1[1];
    halve<4>(best, b4, (j & 4u) != 0u, 4);
    halve<2>(b4, b2, (j & 2u) != 0u, 2);
    halve<1>(b2, b1, (j & 1u) != 0u, 1);
    const Best b = b1[0];
    const int i = (int)j;
    float q0, q1, q2;
    transform(P, px + (float)(i % 2) * P.scale, py + (float)((i / 2) % 2) * P.scale, pz + (float)((i / 2 / 2) % 2) * P.scale, q0, q1, q2);
    float val;
    if (i == slot) {
        val = L.inherit[node];                           // n[i] = vals[insert][i], dllmain.cpp:181
    } else if (b.k == 0xFFFFFFFFu || isinf(b.d) || isnan(b.d)) {
        atomicExch(err, 2u);
        val = 0.0f;
    } else {                                             // DistanceAt, dllmain.cpp:119-149
        const float *v = P.verts + 6 * (size_t)__float_as_uint(cand[off + b.k].w);
        float md = sqrtf(b.d);
        const float ex = q0 - v[0], ey = q1 - v[1], ez = q2 - v[2];      // p - closest.Position
        if ((double)md < 0.015) {
            const float nl = sqrtf(lensq(v[3], v[4], v[5]));
            md = (v[3] / nl) * ex + (v[4] / nl) * ey + (v[5] / nl) * ez;
        } else if (v[3] * (v[0] - q0) + v[4] * (v[1] - q1) + v[5] * (v[2] - q2) > 0) {   // Inside
            md *= -1;
        }
        val = md / P.gs;
    }
    L.vals[8 * (size_t)node + j] = val;
    if (j == 0) L.split[node] = (center_value < P.scale * 2 && P.depth < P.max_depth) ? 1u : 0u;
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

// the root: construct(all, 0, 0, -1, 0)
__global__ void k_root(LevelArrays L, uint32_t n_points)
{
    L.px[0] = L.py[0] = L.pz[0] = L.inherit[0] = 0.0f;
    L.slot[0] = L.parent[0] = -1;
    L.cand_off[0] = 0; L.cand_cnt[0] = n_points;
}

// FindDimensions (dllmain.cpp:67-80) over the points as they lie in device memory: per workgroup the minima and maxima of x, y, z
// (fminf / fmaxf as on the host: a NaN coordinate is passed over), the host folds the BOUNDS_WG partial results
constexpr uint32_t BOUNDS_WG = 256;
__global__ __launch_bounds__(256) void k_bounds(const float *__restrict__ verts, uint32_t n, float *__restrict__ partial)
{
    __shared__ float sh[4][6];
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += BOUNDS_WG * 256u)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float v = verts[6 * (size_t)i + k];
            lo[k] = fminf(lo[k], v); hi[k] = fmaxf(hi[k], v);
        }
#pragma unroll
    for (int k = 0; k < 3; k++)
        for (int o = 32; o > 0; o >>= 1) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], o)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o)); }
    if ((threadIdx.x & 63u) == 0)
        for (int k = 0; k < 3; k++) { sh[threadIdx.x >> 6][k] = lo[k]; sh[threadIdx.x >> 6][3 + k] = hi[k]; }
    __syncthreads();
    if (threadIdx.x < 6u) {
        float v = sh[0][threadIdx.x];
        for (int w = 1; w < 4; w++) v = threadIdx.x < 3u ? fminf(v, sh[w][threadIdx.x]) : fmaxf(v, sh[w][threadIdx.x]);
        partial[6 * blockIdx.x + threadIdx.x] = v;
    }
}

// the root's list: every point, in input order
__global__ void k_cand_init(Cand *p, const float *__restrict__ verts, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        p[i] = make_float4(verts[6 * (size_t)i], verts[6 * (size_t)i + 1], verts[6 * (size_t)i + 2], __uint_as_float(i));
}

// exclusive scan of n uint32 in two launches: per-chunk sums, then every workgroup adds the sums of
// the chunks before it to the scan of its own chunk; total -> *total
constexpr uint32_t SCAN_CHUNK = 1024;
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *wsum)   // all threads get the sum
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if (lane == 0) wsum[wave] = v;
    __syncthreads();
    uint32_t s = 0;
    for (uint32_t w = 0; w < SCAN_CHUNK / 64; w++) s += wsum[w];
    return s;
}
__global__ __launch_bounds__(SCAN_CHUNK) void k_scan_sums(const uint32_t *__restrict__ in, uint32_t *__restrict__ sums, uint32_t n)
{
    __shared__ uint32_t wsum[SCAN_CHUNK / 64];
    const uint32_t i = blockIdx.x * SCAN_CHUNK + threadIdx.x;
    const uint32_t s = block_sum(i < n ? in[i] : 0u, wsum);
    if (threadIdx.x == 0) sums[blockIdx.x] = s;
}
// What the host waits for at the two points of a level where it must know a total before it can go on (the size of the next
// lists; the number of nodes of the next level, and whether a cell ran out of candidates): written by the scan's last thread
// straight into page-locked host memory -- one stream synchronisation per point instead of a blocking copy per word.
// What the host waits for twice per level, in page-locked memory it can read without a call: the scan's total, the error word, and
// `seq` -- the number the host gave this scan, written LAST behind a system-scope fence: the host spins on it instead of
// synchronising the stream (a round trip through the runtime's wait costs tens of microseconds; 21 of them per build).
struct Report { unsigned long long total; uint32_t err; volatile uint32_t seq; };
__global__ __launch_bounds__(SCAN_CHUNK) void k_scan_apply(const uint32_t *__restrict__ in, const uint32_t *__restrict__ sums,
                                                            uint32_t *__restrict__ out, uint32_t n, const uint32_t *err, Report *report, uint32_t seq)
{
    __shared__ uint32_t wsum[SCAN_CHUNK / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    __shared__ unsigned long long w64[SCAN_CHUNK / 64];
    unsigned long long before = 0;                       // totals may pass 2^32 (the host checks)
    {
        unsigned long long mine = 0;
        for (uint32_t b = tid; b < blockIdx.x; b += SCAN_CHUNK) mine += sums[b];
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
        if (lane == 0) w64[wave] = mine;
        __syncthreads();
        for (uint32_t w = 0; w < SCAN_CHUNK / 64; w++) before += w64[w];
    }
    const uint32_t i = blockIdx.x * SCAN_CHUNK + tid;
    uint32_t v = i < n ? in[i] : 0u, x = v;
    for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o); if ((int)lane >= o) x += y; }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < wave; w++) woff += wsum[w];
    if (i < n) out[i] = (uint32_t)(before + woff + x - v);
    if (i == n - 1) {
        report->total = before + woff + x; report->err = *err;
        __threadfence_system();
        report->seq = seq;
    }
}

// ---- the reference's node order, on the GPU ------------------------------------------------------
// cnt = number of split nodes in the subtree of a node (itself included); levels bottom-up
__global__ void k_subtree(const uint32_t *__restrict__ split, const uint32_t *__restrict__ block_of,
                          const uint32_t *__restrict__ cnt_next, uint32_t *__restrict__ cnt, uint32_t n_nodes)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_nodes) return;
    uint32_t c = 0;
    if (split[j]) {
        c = 1;
        const uint32_t first = 8 * block_of[j];
        for (int k = 0; k < 8; k++) c += cnt_next[first + k];
    }
    cnt[j] = c;
}
// The reference appends the children block of a node when it processes the node, then recurses into
// the children in order (dllmain.cpp:183-189): the block of a split node with r split nodes before it
// in pre-order starts at 1 + 8r.  rank(child k of p) = rank(p) + 1 + sum of cnt over its siblings j < k.
__global__ void k_emit(const uint32_t *__restrict__ split, const int32_t *__restrict__ parent, const float *__restrict__ vals,
                       const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ rank_up, const int32_t *__restrict__ index_up,
                       uint32_t *__restrict__ rank, int32_t *__restrict__ index, float scale, uint32_t n_nodes,
                       int32_t *__restrict__ S, uint8_t *__restrict__ V)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_nodes) return;
    uint32_t r = 0;
    int32_t idx = 0, up = -1;
    if (rank_up) {                                       // not the root
        const uint32_t p = (uint32_t)parent[c], k = c & 7u;
        r = rank_up[p] + 1;
        for (uint32_t j = 0; j < k; j++) r += cnt[c - k + j];
        idx = (int32_t)(1 + 8 * rank_up[p] + k);
        up = index_up[p];
    }
    rank[c] = r;
    index[c] = idx;
    S[2 * (size_t)idx] = up;
    S[2 * (size_t)idx + 1] = split[c] ? (int32_t)(1 + 8 * r) : -1;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {                        // FromFloat, dllmain.cpp:192-196
        const float normd = vals[8 * (size_t)c + k] / 2 / scale;
        const float sat = fminf(fmaxf(normd + 0.25f, 0.0f), 1.0f);
        const uint32_t b = (uint32_t)floorf(sat * 255);
        if (k < 4) lo |= b << (8 * k); else hi |= b << (8 * (k - 4));
    }
    ((uint2 *)V)[idx] = make_uint2(lo, hi);
}

}  // namespace sdfhip

using namespace sdfhip;

namespace {

// Bump allocator over a few large hipMalloc chunks.  reset() makes the memory reusable; work on
// the (single, in-order) stream that still reads the old contents was launched before whatever
// is launched to overwrite them, so no synchronisation is needed.
// The arenas' chunks outlive a build: they go back to a per-process pool (by device) instead of to hipFree, and the next build
// takes them from there.  On this stack the first hipMalloc after a build had freed its ~10 GB took 340 ms -- every build after
// the first one in a process: 31 ms became 370 -- and a build's own allocations are 5 ms of it.  The pool keeps at most
// POOL_MAX_BYTES per device (what a depth-10 build of a million points needs); SDFHIP_GEN_POOL=0 turns it off.
struct ChunkPool {
    struct Item { int device; char *base; size_t size; };
    std::mutex mu;
    std::vector<Item> items;
    size_t held = 0;
    static constexpr size_t POOL_MAX_BYTES = (size_t)24 << 30;
    static bool enabled() { const char *e = getenv("SDFHIP_GEN_POOL"); return !(e && atoi(e) == 0); }
    char *take(int device, size_t want, size_t *size_out)
    {
        std::lock_guard<std::mutex> lk(mu);
        size_t best = items.size();
        for (size_t i = 0; i < items.size(); i++)        // the smallest chunk that fits, and no more than twice as large
            if (items[i].device == device && items[i].size >= want && items[i].size <= 2 * want + ((size_t)64 << 20) &&
                (best == items.size() || items[i].size < items[best].size)) best = i;
        if (best == items.size()) return nullptr;
        char *p = items[best].base;
        *size_out = items[best].size;
        held -= items[best].size;
        items.erase(items.begin() + (long)best);
        return p;
    }
    void give(int device, char *base, size_t size)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            size_t held_here = 0;                            // the limit is per device, as include/sdfhip.h says
            for (const Item &it : items) if (it.device == device) held_here += it.size;
            if (enabled() && held_here + size <= POOL_MAX_BYTES) { items.push_back(Item{device, base, size}); held += size; return; }
        }
        (void)hipFree(base);
    }
    size_t trim(int device)                              // device < 0: every device's chunks
    {
        std::vector<Item> out;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < items.size();)
                if (device < 0 || items[i].device == device) { out.push_back(items[i]); held -= items[i].size; items.erase(items.begin() + (long)i); }
                else i++;
        }
        size_t freed = 0;
        for (auto &it : out) { (void)hipFree(it.base); freed += it.size; }
        return freed;
    }
};
ChunkPool g_pool;

struct Arena {
    struct Chunk { char *base; size_t size, used; };
    std::vector<Chunk> chunks;
    size_t grow;
    int device = 0;
    explicit Arena(size_t grow) : grow(grow) { (void)hipGetDevice(&device); }
    Arena(const Arena &) = delete;
    Arena &operator=(const Arena &) = delete;
    void *take(size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes == 0) bytes = 256;
        for (auto &c : chunks)
            if (c.size - c.used >= bytes) { void *p = c.base + c.used; c.used += bytes; return p; }
        size_t size = bytes > grow ? bytes : grow;
        void *p = ChunkPool::enabled() ? g_pool.take(device, size, &size) : nullptr;
        if (!p) {
            const auto t0 = std::chrono::steady_clock::now();
            if (hipMalloc(&p, size) != hipSuccess) {     // the pool may be holding what this allocation needs, in chunks of other sizes
                (void)hipGetLastError();
                if (g_pool.trim(device) == 0 || hipMalloc(&p, size) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            }
            const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms > 5.0f && lab_env("SDFHIP_GEN_LEVELS")) fprintf(stderr, "sdfgen: hipMalloc(%zu MB) took %.1f ms\n", size >> 20, ms);
        }
        chunks.push_back(Chunk{ (char *)p, size, bytes });
        grow = grow < ((size_t)1 << 30) ? grow * 2 : grow;
        return p;
    }
    template <class T> T *alloc(size_t n) { return (T *)take(n * sizeof(T)); }
    void reset() { for (auto &c : chunks) c.used = 0; }
    ~Arena() { for (auto &c : chunks) g_pool.give(device, c.base, c.size); }
};

// scratch: arrays only the level itself and k_children read; keep: what the final ordering needs
bool alloc_level(Arena &scratch, Arena &keep, LevelArrays &L, size_t n)
{
    L.px = scratch.alloc<float>(n); L.py = scratch.alloc<float>(n); L.pz = scratch.alloc<float>(n);
    L.inherit = scratch.alloc<float>(n); L.slot = scratch.alloc<int32_t>(n);
    L.cand_off = scratch.alloc<uint32_t>(n); L.cand_cnt = scratch.alloc<uint32_t>(n);
    L.center_value = scratch.alloc<float>(n); L.pcount = scratch.alloc<uint32_t>(n);
    L.parent = keep.alloc<int32_t>(n); L.vals = keep.alloc<float>(8 * n);
    L.split = keep.alloc<uint32_t>(n); L.block_of = keep.alloc<uint32_t>(n);
    return L.px && L.py && L.pz && L.inherit && L.slot && L.parent && L.cand_off && L.cand_cnt &&
           L.center_value && L.pcount && L.vals && L.split && L.block_of;
}

bool scan_u32(Arena &scratch, const uint32_t *in, uint32_t *out, uint32_t n, const uint32_t *d_err, sdfhip::Report *report, uint32_t seq)
{
    const uint32_t chunks = (n + sdfhip::SCAN_CHUNK - 1) / sdfhip::SCAN_CHUNK;
    uint32_t *sums = scratch.alloc<uint32_t>(chunks);
    if (!sums) return false;
    hipLaunchKernelGGL(sdfhip::k_scan_sums, dim3(chunks), dim3(sdfhip::SCAN_CHUNK), 0, 0, in, sums, n);
    hipLaunchKernelGGL(sdfhip::k_scan_apply, dim3(chunks), dim3(sdfhip::SCAN_CHUNK), 0, 0, in, sums, out, n, d_err, report, seq);
    return true;
}

// Wait for the scan numbered `seq` to have reported: spin on the page-locked word (the kernels in front of it are microseconds to a
// millisecond of work); after 20 ms of that, ask the runtime -- a kernel that faulted never reports, and the runtime says why.
hipError_t wait_report(const sdfhip::Report *report, uint32_t seq)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; spins++) {
        if (report->seq == seq) return hipSuccess;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
    }
    const hipError_t e = hipStreamSynchronize(0);
    if (e != hipSuccess) return e;
    return report->seq == seq ? hipSuccess : hipErrorUnknown;
}

constexpr uint32_t WIDE_LEVEL = 16384;
constexpr unsigned long long WIDE_MAX_LIST = 8192;
struct KeptLevel { LevelArrays L; uint32_t n; uint32_t *cnt, *rank; int32_t *index; };

}  // namespace

#define GEN_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                             \
            (void)hipGetLastError();   /* the runtime's record of it: a later launch check must not report it as its own */ \
            return fail(SDFHIP_ERR_DEVICE, "sdfgen: %s failed: %s", #expr, hipGetErrorString(e_)); \
        }                                                                                   \
    } while (0)
#define GEN_NOMEM() fail(SDFHIP_ERR_NOMEM, "sdfgen: out of device memory")

// out: the tree on the host (may be null); scene_out: the tree as a scene handle on `device`, made from the builder's own device
// arrays (may be null)
static int sdfgen_impl(int device, const float *verts6, uint32_t n, int32_t depth, sdfhip_octdata *out, sdfhip_scene **scene_out,
                       sdfhip_sdfgen_stats *stats)
{
    if (!verts6 || (!out && !scene_out) || n == 0) return fail(SDFHIP_ERR_ARG, "sdfgen: null argument or empty point cloud");
    if (depth < 0 || depth > 12) return fail(SDFHIP_ERR_ARG, "sdfgen: depth %d outside 0..12", depth);
    if (out) { out->length = 0; out->structs = nullptr; out->values = nullptr; }
    if (scene_out) *scene_out = nullptr;
    auto t0 = std::chrono::steady_clock::now();

    GenParams P;
    P.max_depth = depth;

    int ndev = 0;
    GEN_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(SDFHIP_ERR_DEVICE, "sdfgen: device %d of %d does not exist", device, ndev);
    int prev = -1;
    (void)hipGetDevice(&prev);
    GEN_TRY(hipSetDevice(device));
    struct Restore { int prev; ~Restore() { if (prev >= 0) (void)hipSetDevice(prev); } } restore{prev};

    try {
        Arena keep((size_t)64 << 20), scratch[2] = { Arena((size_t)32 << 20), Arena((size_t)32 << 20) },
              lists[2] = { Arena((size_t)64 << 20), Arena((size_t)64 << 20) };
        float *d_verts = keep.alloc<float>(6 * (size_t)n);
        uint32_t *d_err = keep.alloc<uint32_t>(1);
        // (page-locked and mapped; one per host thread, kept for the life of the process: allocating one costs as much as a level)
        static thread_local Report *report = nullptr;
        if (!report) {
            //  COHERENT, said explicitly: the host spins on `seq` while the kernel is still running -- without fine-grained
            // coherence (HIP_HOST_COHERENT=0 makes that the default) every wait would burn its whole spin budget before the stream
            // synchronisation behind it; and the numbering starts from zeroed bytes, not from whatever the allocation held
            GEN_TRY(hipHostMalloc((void **)&report, sizeof(Report) + BOUNDS_WG * 6 * sizeof(float),
                                  hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
            memset(report, 0, sizeof(Report) + BOUNDS_WG * 6 * sizeof(float));
        }
        float *bounds = reinterpret_cast<float *>(report + 1);
        uint32_t report_seq = report->seq;                   // (the scans of this build are numbered on from the last build's)
        Cand *cand = lists[0].alloc<Cand>(n);
        if (!d_verts || !d_err || !cand) return GEN_NOMEM();
        GEN_TRY(hipMemcpy(d_verts, verts6, 6 * (size_t)n * sizeof(float), hipMemcpyHostToDevice));
        GEN_TRY(hipMemsetAsync(d_err, 0, sizeof(uint32_t), 0));
        hipLaunchKernelGGL(k_bounds, dim3(BOUNDS_WG), dim3(256), 0, 0, (const float *)d_verts, n, bounds);
        GEN_TRY(hipStreamSynchronize(0));
        {   // FindDimensions, dllmain.cpp:67-80 (the pass over the points ran on the device: 2 ms of a 15 ms build on the host)
            float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
            for (uint32_t w = 0; w < BOUNDS_WG; w++)
                for (int k = 0; k < 3; k++) {
                    lo[k] = fminf(lo[k], ((volatile float *)bounds)[6 * w + k]);
                    hi[k] = fmaxf(hi[k], ((volatile float *)bounds)[6 * w + 3 + k]);
                }
            P.gox = (lo[0] + hi[0]) * 0.5f + 0.003f;
            P.goy = (lo[1] + hi[1]) * 0.5f + 0.003f;
            P.goz = (lo[2] + hi[2]) * 0.5f + 0.003f;
            const float lowest = fminf(lo[0], fminf(lo[1], lo[2])), highest = fmaxf(hi[0], fmaxf(hi[1], hi[2]));
            P.gs = (highest - lowest) * 1.1f;
        }
        P.verts = d_verts;
        hipLaunchKernelGGL(k_cand_init, dim3(1024), dim3(256), 0, 0, cand, (const float *)d_verts, n);

        LevelArrays L;
        if (!alloc_level(scratch[0], keep, L, 1)) return GEN_NOMEM();
        hipLaunchKernelGGL(k_root, dim3(1), dim3(1), 0, 0, L, n);
        std::vector<KeptLevel> levels;
        uint32_t n_nodes = 1;
        size_t total_nodes = 0;
        unsigned long long cand_entries = n, list_entries = n;      // all lists so far; the lists of the level at hand
        // (SDFHIP_GEN_WIDE=n: the sibling-block form from n nodes on -- tests set 8 to run small trees through it)
        const bool wide_forced = lab_env("SDFHIP_GEN_WIDE") != nullptr;
        const uint32_t wide_level = wide_forced ? (uint32_t)atoi(lab_env("SDFHIP_GEN_WIDE")) : WIDE_LEVEL;
        const uint32_t seg_target = lab_env("SDFHIP_GEN_SEGS") ? (uint32_t)max(1, atoi(lab_env("SDFHIP_GEN_SEGS"))) : 4096u;
        const bool level_timing = lab_env("SDFHIP_GEN_LEVELS") != nullptr;      // debug aid: nodes, entries and time of every level on stderr
        for (int lvl = 0;; lvl++) {
            auto tl = std::chrono::steady_clock::now();
            Arena &mine = scratch[lvl & 1], &other = scratch[(lvl + 1) & 1];
            P.depth = lvl;
            P.scale = ldexpf(1.0f, -lvl);                 // powf(0.5, depth)
            uint32_t *poff = mine.alloc<uint32_t>(n_nodes);
            if (!poff) return GEN_NOMEM();
            // which form: a wavefront per sibling block (k_center_sib / k_corners_sib) where that fills the chip -- 16 384 nodes = 2 048
            // wavefronts and more; below the root a level's nodes come in blocks of eight -- else a node's list over S workgroups of
            // 1 024 threads (k_center_seg_min, ...).  (1 M-point knot, depth 10: level 5, 10 776 nodes with lists of 50 000: 2.9 ms
            // in segments, 3.9 by sibling blocks; level 6, 37 896 nodes: 4.7 against 1.9)
            // ... and lists short enough that one wavefront per block is not the level's critical path (level 6 of the same knot:
            // 4 737 blocks with lists of 17 000 -- 1.6 ms by sibling blocks, 1.2 in segments; level 7, lists of 5 000: 1.1 against 2.2)
            const bool wide = n_nodes >= wide_level && lvl >= 1 && (n_nodes & 7u) == 0 &&
                              (wide_forced || list_entries / (n_nodes / 8) <= WIDE_MAX_LIST);
            SegArrays A{ nullptr, nullptr, nullptr, wide ? 0u : (seg_target + n_nodes - 1) / n_nodes };      // segments per node: ~4 096 workgroups per level
            // (the lists of a level are `list_entries` long together, one per block of eight siblings)
            const bool pipe = wide && list_entries / (n_nodes / 8) > 2u * (unsigned long long)SIB_CH;
            const bool fused = wide && !pipe && lvl >= depth;          // the last level: centre and corner searches in one kernel
            if (fused) ;
            else if (wide && pipe) hipLaunchKernelGGL(k_center_sib<true>, dim3(n_nodes / 8), dim3(64), 0, 0, P, L, cand, n_nodes, d_err);
            else if (wide) hipLaunchKernelGGL(k_center_sib<false>, dim3(n_nodes / 8), dim3(64), 0, 0, P, L, cand, n_nodes, d_err);
            else {
                A.best = mine.alloc<unsigned long long>(n_nodes); A.corner = mine.alloc<unsigned long long>(8 * (size_t)n_nodes);
                A.count = mine.alloc<uint32_t>((size_t)n_nodes * A.S);
                if (!A.best || !A.corner || !A.count) return GEN_NOMEM();
                GEN_TRY(hipMemsetAsync(A.best, 0xFF, (size_t)n_nodes * 8, 0));
                GEN_TRY(hipMemsetAsync(A.corner, 0xFF, (size_t)n_nodes * 64, 0));
                GEN_TRY(hipMemsetAsync(L.pcount, 0, (size_t)n_nodes * 4, 0));
                hipLaunchKernelGGL(k_center_seg_min, dim3((n_nodes * A.S + 7u) & ~7u), dim3(SEG_BT), 0, 0, P, L, A, cand, n_nodes);
                hipLaunchKernelGGL(k_center_seg_count, dim3((n_nodes * A.S + 7u) & ~7u), dim3(SEG_BT), 0, 0, P, L, A, cand, n_nodes, d_err);
            }
            // The survivors' lists of this level are the candidate lists of the next: their lengths are scanned into offsets and the
            // host allocates them -- except on the last level (lvl == MaxDepth), where no node splits: its survivors are looked at
            // for the corner values and written nowhere (1.9 GB of stores, a scan and a host round trip on the 1 M-point knot).
            const bool last_level = lvl >= depth;
            unsigned long long total = 0;
            Cand *possible = nullptr;
            if (!last_level) {
                if (!scan_u32(mine, L.pcount, poff, n_nodes, d_err, report, ++report_seq)) return GEN_NOMEM();
                GEN_TRY(wait_report(report, report_seq));
                total = ((volatile Report *)report)->total;
                if (total > 0xFFFFFFF0ull) return fail(SDFHIP_ERR_NOMEM, "sdfgen: candidate lists of level %d exceed 2^32 entries", lvl);
                lists[(lvl + 1) & 1].reset();             // the lists of level lvl-1: dead since k_children of lvl-1
                possible = lists[(lvl + 1) & 1].alloc<Cand>((size_t)total);
                if (!possible) return fail(SDFHIP_ERR_NOMEM, "sdfgen: out of device memory for %llu candidate entries", total);
            }
            cand_entries += total;
            list_entries = total;
            if (fused) hipLaunchKernelGGL((k_corners_sib<false, true>), dim3(n_nodes / 8), dim3(64), 0, 0, P, L, cand, poff, possible, n_nodes, d_err);
            else if (wide && pipe) hipLaunchKernelGGL(k_corners_sib<true>, dim3(n_nodes / 8), dim3(64), 0, 0, P, L, cand, poff, possible, n_nodes, d_err);
            else if (wide) hipLaunchKernelGGL(k_corners_sib<false>, dim3(n_nodes / 8), dim3(64), 0, 0, P, L, cand, poff, possible, n_nodes, d_err);
            else {
                hipLaunchKernelGGL(k_corners_seg, dim3((n_nodes * A.S + 7u) & ~7u), dim3(SEG_BT), 0, 0, P, L, A, cand, poff, possible, n_nodes);
                hipLaunchKernelGGL(k_corners_fin, dim3((8 * n_nodes + 255) / 256), dim3(256), 0, 0, P, L, A, cand, n_nodes, d_err);
            }
            if (!scan_u32(mine, L.split, L.block_of, n_nodes, d_err, report, ++report_seq)) return GEN_NOMEM();
            GEN_TRY(wait_report(report, report_seq));
            struct { unsigned long long n_split; uint32_t err; } back;
            back.n_split = ((volatile Report *)report)->total;
            back.err = ((volatile Report *)report)->err;
            if (back.err) return fail(SDFHIP_ERR_ARG, "sdfgen: a cell at depth %d has no candidate point left (the reference throws \"Did not find\")", lvl);
            if (level_timing) {
                (void)hipDeviceSynchronize();
                fprintf(stderr, "sdfgen: level %2d: %9u nodes, %11llu entries kept for the next level, %8.3f ms\n", lvl, n_nodes, total,
                        std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tl).count());
            }
            levels.push_back(KeptLevel{ L, n_nodes, nullptr, nullptr, nullptr });
            total_nodes += n_nodes;
            const unsigned long long n_split = back.n_split;
            if (n_split == 0) break;
            if (total_nodes + 8 * n_split > 0x7FFFFFF0ull) return fail(SDFHIP_ERR_NOMEM, "sdfgen: more than 2^31 nodes");
            other.reset();                                // scratch of level lvl-1
            LevelArrays N;
            if (!alloc_level(other, keep, N, (size_t)(8 * n_split))) return GEN_NOMEM();
            hipLaunchKernelGGL(k_children, dim3((8 * n_nodes + 255) / 256), dim3(256), 0, 0, L, N, L.block_of, poff,
                               P.scale / 2, n_nodes);
            L = N;
            cand = possible;
            n_nodes = (uint32_t)(8 * n_split);
        }

        const bool timing = lab_env("SDFHIP_GEN_TIMING") != nullptr;       // debug aid: phase times on stderr
        auto lap = [&](const char *what) {
            if (!timing) return;
            (void)hipDeviceSynchronize();
            fprintf(stderr, "sdfgen: %-28s at %.2f ms\n", what, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count());
        };
        lap("levels done");
        // ---- the reference's node order and bytes (k_subtree bottom-up, k_emit top-down) ----
        int32_t *d_S = keep.alloc<int32_t>(2 * total_nodes);
        uint8_t *d_V = keep.alloc<uint8_t>(8 * total_nodes);
        if (!d_S || !d_V) return GEN_NOMEM();
        for (auto &kl : levels) {
            kl.cnt = keep.alloc<uint32_t>(kl.n); kl.rank = keep.alloc<uint32_t>(kl.n); kl.index = keep.alloc<int32_t>(kl.n);
            if (!kl.cnt || !kl.rank || !kl.index) return GEN_NOMEM();
        }
        for (int l = (int)levels.size() - 1; l >= 0; l--) {
            KeptLevel &kl = levels[l];
            const uint32_t *next = l + 1 < (int)levels.size() ? levels[l + 1].cnt : nullptr;   // the last level splits nothing
            hipLaunchKernelGGL(k_subtree, dim3((kl.n + 255) / 256), dim3(256), 0, 0, kl.L.split, kl.L.block_of, next, kl.cnt, kl.n);
        }
        for (size_t l = 0; l < levels.size(); l++) {
            KeptLevel &kl = levels[l];
            hipLaunchKernelGGL(k_emit, dim3((kl.n + 255) / 256), dim3(256), 0, 0, kl.L.split, kl.L.parent, kl.L.vals, kl.cnt,
                               l ? levels[l - 1].rank : nullptr, l ? levels[l - 1].index : nullptr, kl.rank, kl.index,
                               ldexpf(1.0f, -(int)l), kl.n, d_S, d_V);
        }
        GEN_TRY(hipGetLastError());
        lap("order + bytes done");
        // the result arrays (the caller frees them with free()): large ones on 2 MB pages where the kernel offers them -- the
        // copy from the device writes every page for the first time, and 47 000 page faults of 4 KB take five times as long as
        // the 192 MB of a 12 M-node tree take over the link
        auto result_alloc = [](size_t bytes) -> void * {
            if (bytes < ((size_t)8 << 20)) return malloc(bytes);
            void *p = nullptr;
            if (posix_memalign(&p, (size_t)2 << 20, bytes) != 0) return malloc(bytes);
            (void)madvise(p, bytes, MADV_HUGEPAGE);
            return p;
        };
        if (scene_out) {
            // the viewer's generate -> upload flow (Program.cs:613-650 -> :147-152) without the round trip: the scene's records
            // and grids are made from the arrays the builder has just written (12 ms of copy to the host and 8 ms back, at depth 10)
            GEN_TRY(hipDeviceSynchronize());
            const int rc = sdfhip::scene_from_arrays(device, d_S, d_V, (uint32_t)total_nodes, true, nullptr, scene_out,
                                                     (int)levels.size() - 1);     // (the depth it built: levels 0 .. size - 1)
            if (rc != SDFHIP_OK) return rc;
            lap("scene made on the device");
        }
        if (out) {
            int32_t *S = (int32_t *)result_alloc(total_nodes * 8);
            uint8_t *V = (uint8_t *)result_alloc(total_nodes * 8);
            hipError_t e2 = hipSuccess;
            if (!S || !V) e2 = hipErrorOutOfMemory;
            else {
                e2 = hipMemcpy(S, d_S, total_nodes * 8, hipMemcpyDeviceToHost);
                if (e2 == hipSuccess) e2 = hipMemcpy(V, d_V, total_nodes * 8, hipMemcpyDeviceToHost);
            }
            if (e2 != hipSuccess) {
                free(S); free(V);
                if (scene_out && *scene_out) { (void)sdfhip_scene_free(*scene_out); *scene_out = nullptr; }
                return fail(SDFHIP_ERR_DEVICE, "sdfgen: copying the tree back failed: %s", hipGetErrorString(e2));
            }
            lap("copied to the host");
            out->length = (uint32_t)total_nodes; out->structs = S; out->values = V;
        }
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

hipError_t sdfhip::device_alloc_bytes(void **p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipErrorOutOfMemory) return e;
    (void)hipGetLastError();
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || g_pool.trim(device) == 0) return e;       // nothing to give back: the failure stands
    return hipMalloc(p, bytes);
}

extern "C" int sdfhip_sdfgen_trim(void)
try {
    (void)g_pool.trim(-1);
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_sdfgen_trim)

extern "C" int sdfhip_sdfgen(int device, const float *verts6, uint32_t n, int32_t depth, sdfhip_octdata *out,
                             sdfhip_sdfgen_stats *stats)
try {
    if (!out) return fail(SDFHIP_ERR_ARG, "sdfgen: null argument or empty point cloud");
    return sdfgen_impl(device, verts6, n, depth, out, nullptr, stats);
}
SDFHIP_ABI_CATCH(sdfhip_sdfgen)

extern "C" int sdfhip_sdfgen_scene(int device, const float *verts6, uint32_t n, int32_t depth, sdfhip_scene **scene,
                                   sdfhip_octdata *out, sdfhip_sdfgen_stats *stats)
try {
    if (!scene) return fail(SDFHIP_ERR_ARG, "sdfgen_scene: null argument");
    return sdfgen_impl(device, verts6, n, depth, out, scene, stats);
}
SDFHIP_ABI_CATCH(sdfhip_sdfgen_scene)
