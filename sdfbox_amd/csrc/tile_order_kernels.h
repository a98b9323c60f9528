// SDFHIP_FLAG_TILE_ORDER: the next frame's launch order from this frame's per-tile cost (gfx950).  Non-template kernels:
// included by render.hip ONLY.
#pragma once
#include "raymarch_device.h"

namespace sdfhip {

// SDFHIP_FLAG_TILE_ORDER: the launch order of the next frame's tiles from the wave-iterations of this frame's (k_march's per-tile
// cost output).  A frame alone ends when its longest wave does -- a 100-step tile launched in the last round of workgroups adds
// its whole chain of dependent steps to the frame time -- so the expensive tiles go first.  Workgroup b runs on XCD b & 7 and the
// default order gives XCD x the tile rows x, x + 8, ...: every XCD keeps its own tiles (its L2 keeps seeing whole rows) and
// launches them in 8 classes of descending cost, in their old order within a class (a stable partition: neighbouring tiles of
// equal cost -- the sky -- still run together).  One workgroup per XCD label; thread t owns a contiguous span of the label's tiles.
// A tile's expected cost is the largest cost among the tiles up to ORDER_REACH away from it: the camera moves between frames,
// and a silhouette that crosses into a neighbouring tile must find that tile at the front too (with the camera turning round
// the scene by one degree per frame this takes a frame alone from 0.168 to 0.144 ms, the same as knowing the frame's own costs).
constexpr int ORDER_CLASSES = 8;
constexpr int ORDER_REACH = 2;
constexpr int ORDER_SPAN = 12;                                         // tiles per thread at most: 8 x 1024 x 12 order slots
// one thread per tile: its class, 0 = the longest (all CUs take part: the 25 reads per tile are the bulk of the work)
__global__ __launch_bounds__(256) void k_tile_class(const uint16_t *__restrict__ cost, uint8_t *__restrict__ cls, uint32_t tiles_x, uint32_t tiles_y)
{
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= tiles_x * tiles_y) return;
    const int tx = (int)(tile % tiles_x), ty = (int)(tile / tiles_x);
    uint32_t c = 0;                                                    // primary + shadow loop iterations, <= 140
    for (int y = max(ty - ORDER_REACH, 0); y <= min(ty + ORDER_REACH, (int)tiles_y - 1); y++)
        for (int x = max(tx - ORDER_REACH, 0); x <= min(tx + ORDER_REACH, (int)tiles_x - 1); x++) {
            const uint32_t v = cost[(size_t)y * tiles_x + x];
            c = max(c, (v & 0xFFu) + (v >> 8));
        }
    const uint32_t k = c / 18u;                                        // 0 .. 7
    cls[tile] = (uint8_t)((uint32_t)(ORDER_CLASSES - 1) - (k < (uint32_t)ORDER_CLASSES ? k : (uint32_t)(ORDER_CLASSES - 1)));
}
// one workgroup per XCD label: the stable partition of its tiles by class
__global__ __launch_bounds__(1024) void k_tile_order(const uint8_t *__restrict__ cls, uint32_t *__restrict__ perm,
                                                     uint32_t tiles_x, uint32_t tiles_y)
{
    __shared__ uint32_t wsum[ORDER_CLASSES][16];
    __shared__ uint32_t total[ORDER_CLASSES], base[ORDER_CLASSES + 1];
    const uint32_t x = blockIdx.x, t = threadIdx.x;
    const uint32_t per_label = ((tiles_y + 7u) >> 3) * tiles_x;       // workgroups (and order slots) per XCD label
    const uint32_t span = (per_label + 1023u) / 1024u, lo = min(per_label, t * span), hi = min(per_label, lo + span);
    auto tile_of = [&](uint32_t j) { const uint32_t r = j / tiles_x, row = r * 8u + x; return row < tiles_y ? row * tiles_x + (j - r * tiles_x) : 0xFFFFFFFFu; };
    auto coords = [&](uint32_t tile) { const uint32_t row = tile / tiles_x; return row << 16 | (tile - row * tiles_x); };     // what k_march reads
    // the classes of this thread's tiles, 4 bits each (15 = no tile), kept for the second pass
    unsigned long long packed = ~0ull;
    uint32_t mine[ORDER_CLASSES];
    for (int k = 0; k < ORDER_CLASSES; k++) mine[k] = 0;
#pragma unroll
    for (int i = 0; i < ORDER_SPAN; i++) {
        const uint32_t j = lo + (uint32_t)i;
        const uint32_t tile = j < hi ? tile_of(j) : 0xFFFFFFFFu;
        if (tile != 0xFFFFFFFFu) {
            const uint32_t c = cls[tile];
            packed = (packed & ~(0xFull << (4 * i))) | ((unsigned long long)c << (4 * i));
            for (int k = 0; k < ORDER_CLASSES; k++) mine[k] += c == (uint32_t)k ? 1u : 0u;      // (no indexed register array)
        }
    }
    // exclusive prefix of every class over the 1024 threads: within the wave by shuffles, across the 16 waves through LDS
    const uint32_t lane = t & 63u, wave = t >> 6;
    uint32_t before[ORDER_CLASSES];
    for (int k = 0; k < ORDER_CLASSES; k++) {
        uint32_t v = mine[k];
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(v, o); if ((int)lane >= o) v += y; }
        before[k] = v - mine[k];
        if (lane == 63u) wsum[k][wave] = v;
    }
    __syncthreads();
    if (t < (uint32_t)ORDER_CLASSES) {                                 // thread k: class k's waves, then the class bases
        uint32_t run = 0;
        for (int w = 0; w < 16; w++) { const uint32_t v = wsum[t][w]; wsum[t][w] = run; run += v; }
        total[t] = run;
    }
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (int k = 0; k < ORDER_CLASSES; k++) { base[k] = run; run += total[k]; }
        base[ORDER_CLASSES] = run;
    }
    __syncthreads();
    uint32_t at[ORDER_CLASSES];
    for (int k = 0; k < ORDER_CLASSES; k++) at[k] = base[k] + wsum[k][wave] + before[k];
#pragma unroll
    for (int i = 0; i < ORDER_SPAN; i++) {
        const uint32_t c = (uint32_t)(packed >> (4 * i)) & 0xFu;
        if (c == 0xFu) continue;
        uint32_t slot = 0;
        for (int k = 0; k < ORDER_CLASSES; k++) { const bool m = c == (uint32_t)k; slot = m ? at[k] : slot; at[k] += m ? 1u : 0u; }
        perm[(size_t)x * per_label + slot] = coords(tile_of(lo + (uint32_t)i));
    }
    // the label's idle workgroups (rows past the frame's last tile row) behind its tiles
    const uint32_t real = base[ORDER_CLASSES];
    for (uint32_t j = real + t; j < per_label; j += 1024u) perm[(size_t)x * per_label + j] = 0xFFFFFFFFu;
}
}  // namespace sdfhip
