// The upload's kernels (gfx950): fused 16-byte node records, the tree's validation, the lookup grids that turn find() into
// one or two loads.  Non-template kernels: included by scene.hip ONLY.
#pragma once
#include "raymarch_device.h"

namespace sdfhip {

// ---- small helper kernels -----------------------------------------------------
// {parent, children}[N] + bytes[N][8] -> fused 16-byte records (upload).
__global__ void k_fuse(const int2 *__restrict__ structs, const uint2 *__restrict__ values,
                       NodeRec *__restrict__ nodes, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int2 s = structs[i];
        uint2 v = values[i];
        nodes[i] = make_uint4((uint32_t)s.x, (uint32_t)s.y, v.x, v.y);
    }
}

// A leaf cell of a grid that serves CursorFT (t.level = LM - level): when its 8 bytes are equal, mark it flat and
// replace v1 by the distance it returns (raymarch_device.h, CursorFT)
__device__ __forceinline__ void flat_cell(TopCell &t)
{
    if (t.v0 == t.v1 && t.v0 == __builtin_amdgcn_alignbit(t.v0, t.v0, 8)) {
        t.v1 = flat_cell_distance_bits(t.v0 & 0xFFu, t.level);
        t.level |= FLAT_BIT;
    }
}

// The top grid of the cursor-stack kernels (raymarch_device.h): one thread per level-TG cell walks
// from the root by the cell's octant bits and stores the record it ends at.
// full = 0: cells hold the level (CursorS); 1: LM - level (CursorF, grid as deep as the tree); 2: like 1, but a
// cell whose node is still internal holds level 15 and the node's index: the coarse half of a split grid
__global__ void k_top_grid(const NodeRec *__restrict__ nodes, TopCell *__restrict__ top, int TG, int full)
{
    const uint32_t total = 1u << (3 * TG), mask = (1u << TG) - 1u;
    for (uint32_t cell = blockIdx.x * blockDim.x + threadIdx.x; cell < total; cell += gridDim.x * blockDim.x) {
        const uint32_t cx = cell & mask, cy = (cell >> TG) & mask, cz = cell >> (2 * TG);
        NodeRec r = nodes[0];
        uint32_t level = 0, index = 0;
        while (level < (uint32_t)TG && (int32_t)r.y >= 0) {
            const uint32_t sb = (uint32_t)TG - 1u - level;
            index = r.y + ((cx >> sb & 1u) | ((cy >> sb & 1u) << 1) | ((cz >> sb & 1u) << 2));
            r = nodes[index];
            level++;
        }
        TopCell t;
        // a grid as deep as the tree serves CursorFT, which wants LM - level
        t.level = full ? (uint32_t)LM - level : level; t.v0 = r.z; t.v1 = r.w; t.children = (int32_t)r.y;
        if (full == 2 && (int32_t)r.y >= 0) { t.level = 15u; t.children = (int32_t)index; }
        else if (full) flat_cell(t);
        top[top_index(cx, cy, cz, TG)] = t;
    }
}

// The fine half of a split grid: block b holds the 8^FB cells below the internal node block_node[b] of level TG,
// each the leaf that contains it (LM - level, values), in x-y-z order.
__global__ void k_fine_blocks(const NodeRec *__restrict__ nodes, const uint32_t *__restrict__ block_node,
                              TopCell *__restrict__ fine, uint32_t n_blocks, int TG, int FB, int order)
{
    const size_t total = (size_t)n_blocks << (3 * FB);
    const uint32_t mask = (1u << FB) - 1u;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (size_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)(j >> (3 * FB)), local = (uint32_t)j & ((1u << (3 * FB)) - 1u);
        const uint32_t cx = local & mask, cy = (local >> FB) & mask, cz = local >> (2 * FB);
        const size_t i = ((size_t)b << (3 * FB)) + fine_cell_index(cx, cy, cz, FB, order);    // where the cell is stored
        NodeRec r = nodes[block_node[b]];
        uint32_t level = (uint32_t)TG, down = 0;
        while (down < (uint32_t)FB && (int32_t)r.y >= 0) {
            const uint32_t sb = (uint32_t)FB - 1u - down;
            r = nodes[r.y + ((cx >> sb & 1u) | ((cy >> sb & 1u) << 1) | ((cz >> sb & 1u) << 2))];
            down++; level++;
        }
        TopCell t;
        t.level = (uint32_t)LM - level; t.v0 = r.z; t.v1 = r.w; t.children = -1;
        flat_cell(t);
        fine[i] = t;
    }
}

// sdfhip_octdata_validate (asdf_io.cpp) on the device, for the upload: the arrays are on their way to HBM anyway (8 ms for
// 451 MB) and the host's single-threaded pass over 28 M nodes takes 110 ms.  The same verdicts:
//   bad          a link out of range, or a parent chain of more than 64 links / a cycle (the host function then names the node)
//   inconsistent the root has a parent, a node's children block is block 0 or does not point back at it
//   depth        the most links from a node to the root among the nodes the root reaches (every link of the chain is mirrored by
//                its parent's children block): the host function's walk down from the root
// Every node's chain to the root by POINTER JUMPING (round 4; until then every thread walked its own chain link by link: <= 9
// dependent loads per node, 5.0 ms for 28 M nodes): k_validate_init checks a node's own links and writes {where my chain has got
// to, how many links that is, are they all mirrored}; a round of k_validate_jump doubles every chain's reach by appending the
// chain of the node it has got to (A -> B -> A: two buffers, no race).  Four rounds resolve chains of up to 16 links -- every
// tree the shader can descend -- and the last of them reduces the verdicts; chains still unresolved then get three more rounds
// (128 links), and what is unresolved after those is longer than 64 links or a cycle: bad.  Reads nothing through a link that
// has not been range-checked.
struct Jump { int32_t at; uint32_t info; };              // at: the node the chain has reached (-1: its end); info: links | ok << 8
constexpr uint32_t JUMP_OK = 0x100u, JUMP_LINKS = 0xFFu;

__global__ __launch_bounds__(256) void k_validate_init(const int2 *__restrict__ structs, uint32_t n, uint32_t *__restrict__ verdict,
                                                       Jump *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    const int2 me = valid ? structs[i] : make_int2(-1, -1);
    bool bad = false, inconsistent = false;
    Jump jmp{-1, 0u};
    if (valid) {
        if (me.x >= 0 && (uint32_t)me.x >= n) bad = true;
        if (me.y >= 0 && (uint64_t)(uint32_t)me.y + 8u > (uint64_t)n) bad = true;
        if (i == 0 && me.x >= 0) inconsistent = true;
        if (!bad && me.y >= 0) {
            if (me.y == 0) inconsistent = true;
            for (int k = 0; k < 8; k++)
                if (structs[(uint32_t)me.y + (uint32_t)k].x != (int32_t)i) inconsistent = true;
        }
        if (me.x < 0) {
            jmp.info = i == 0 ? JUMP_OK : 0u;            // a chain must end at node 0
        } else if ((uint32_t)me.x < n) {
            const int2 up = structs[(uint32_t)me.x];
            const bool mirrored = up.y >= 0 && i >= (uint32_t)up.y && i - (uint32_t)up.y < 8u;
            jmp.at = me.x; jmp.info = 1u | (mirrored && !bad ? JUMP_OK : 0u);
        }                                                // (a parent out of range: this thread has said `bad`; the chain is not attached)
        out[i] = jmp;
    }
    const uint32_t flags = (__ballot(bad) ? 1u : 0u) | (__ballot(inconsistent) ? 2u : 0u);
    if ((threadIdx.x & 63u) == 0 && flags) atomicOr(&verdict[0], flags);
}

// LAST: also the verdicts of the resolved chains (more than 64 links: bad; the deepest attached node) and, in verdict[0] bit 2,
// whether any chain is still on its way
template <bool LAST>
__global__ __launch_bounds__(256) void k_validate_jump(const Jump *__restrict__ in, Jump *__restrict__ out, uint32_t n, uint32_t *__restrict__ verdict)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Jump a{-1, 0u};
    if (i < n) {
        a = in[i];
        if (a.at >= 0) {
            const Jump t = in[(uint32_t)a.at];
            const uint32_t links = min((a.info & JUMP_LINKS) + (t.info & JUMP_LINKS), JUMP_LINKS);
            a.info = links | (a.info & t.info & JUMP_OK);
            a.at = t.at;
        }
        out[i] = a;
    }
    if (LAST) {
        const bool open = i < n && a.at >= 0;
        const uint32_t links = a.info & JUMP_LINKS;
        const bool bad = i < n && !open && links > 64u;
        uint32_t m = (i < n && !open && !bad && (a.info & JUMP_OK)) ? links : 0u;
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
        const uint32_t flags = (__ballot(bad) ? 1u : 0u) | (__ballot(open) ? 4u : 0u);
        // (a wave adds its maximum only when it would raise the word: 440 000 atomic maxima on one address serialise at ~90 per
        // microsecond -- that, not the chain walks, was most of the old kernel's 5 ms)
        if ((threadIdx.x & 63u) == 0) {
            if (flags) atomicOr(&verdict[0], flags);
            if (m > *reinterpret_cast<volatile uint32_t *>(&verdict[1])) atomicMax(&verdict[1], m);
        }
    }
}

// Numbering the internal cells of a split grid's coarse level in cell order (an exclusive prefix sum of "is internal"),
// on the device: chunks of 256 cells; count per chunk, scan of the chunk counts by one workgroup, then every internal cell
// gets its block -- children = the block's first fine cell -- and the block its node.
__global__ __launch_bounds__(256) void k_split_count(const TopCell *__restrict__ coarse, uint32_t ncell, uint32_t n_chunks,
                                                     uint32_t *__restrict__ chunk_sums)
{
    __shared__ uint32_t part[4];
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const uint32_t i = c * 256u + threadIdx.x;
        const bool internal = i < ncell && coarse[i].level == 15u;
        const unsigned long long m = __ballot(internal);
        if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) chunk_sums[c] = part[0] + part[1] + part[2] + part[3];
        __syncthreads();
    }
}
// exclusive scan of n values in place by ONE workgroup of 1024 threads; the total goes to v[n]
__global__ __launch_bounds__(1024) void k_split_scan(uint32_t *__restrict__ v, uint32_t n)
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
__global__ __launch_bounds__(256) void k_split_assign(TopCell *__restrict__ coarse, uint32_t ncell, uint32_t n_chunks,
                                                      const uint32_t *__restrict__ chunk_offsets, uint32_t *__restrict__ block_node, int FB)
{
    __shared__ uint32_t part[4];
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const uint32_t i = c * 256u + threadIdx.x, w = threadIdx.x >> 6;
        const bool internal = i < ncell && coarse[i].level == 15u;
        const unsigned long long m = __ballot(internal);
        if ((threadIdx.x & 63u) == 0) part[w] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = chunk_offsets[c];
        for (uint32_t k = 0; k < w; k++) before += part[k];
        if (internal) {
            const uint32_t id = before + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            block_node[id] = (uint32_t)coarse[i].children;              // the node the block hangs under
            coarse[i].children = (int32_t)(id << (3 * FB));             // the block's first fine cell
        }
        __syncthreads();
    }
}
}  // namespace sdfhip
