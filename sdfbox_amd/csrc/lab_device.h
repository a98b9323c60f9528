// LABORATORY (-DSDFHIP_EXPERIMENTS only; included at the end of raymarch_device.h): the default kernel's loop through the grid's
// second form.  Measured slower than the 16-byte cells; kept as a bit-identical A/B
// (SDFHIP_SAMPLE_RECORDS=1 at upload builds the arrays, SDFHIP_TUNE_BYTE_CELLS switches back to the cells).
#pragma once
#ifndef SDFHIP_EXPERIMENTS
#error "lab_device.h belongs to the experiments build (-DSDFHIP_EXPERIMENTS)"
#endif

namespace sdfhip {

// ---- the default kernel's cursor: a dense grid of 4-byte words + sample records with pre-decoded corners -----------------
// What bounds the march are the VALU pipes and the vector-memory address path (TA) together: the frame
// takes as long as its VALU instructions take to issue, and a load instruction costs the TA one tag lookup per cache line its
// lanes touch.  A third of the wave-iterations pay the 44 instructions of a non-flat sample: the exact R8_UNorm decode of 8
// bytes (8 v_cvt_f32_ubyte + 4 v_pk_mul + 4 v_pk_fma), the differences t1 - t0 of the x-lerps, and the cell's local
// coordinates from its level (masks, conversions, 2^-k) -- all functions of the LEAF alone; and 83 % of the lane-steps are in
// flat leaves, which need 4 bytes: the distance.  So the grid exists a second time, in the form the loop wants:
//   d4     one 32-bit word per cell of the tree's deepest level (dense: 8^F words, 0.54 GB at F = 9): the distance a flat leaf
//          returns, or -- a word that is no distance: as a float it is >= 2 -- TAG + the index, in 16-byte units, of the
//          leaf's sample record.  find() is ONE 4-byte load for every flat step (no coarse / fine levels, no shifts);
//   recs   per non-flat leaf (one per LEAF: the cells of a large leaf share it) 64 bytes: q0 = {-ax 2^-k, -ay 2^-k, -az 2^-k, 2^-k}
//          (a = the leaf's lower corner in the grid's units, 2^k its width), q1 = {t0, t4, t2, t6}, q2 = {t1 - t0, t5 - t4,
//          t3 - t2, t7 - t6} (t_i = unorm8(byte i), the differences rounded as lerp() rounds them).
// The local coordinates are d = sat(fma(u, 2^-k, -a 2^-k)): (u - a) 2^-k without a rounding of its own wherever its value
// matters (inside the cell u - a is exact; outside the cube both saturate the same way), so sat((u - a) * inv) bit for bit.
// The sample: three fmas, 2 + 1 packed fmas and a packed subtraction for the x- and y-lerps, the z-lerp, and
// (v - 0.25) * (2 scale) with 2 scale = 2^(1 - F) / 2^-k from the exponent of q0.w: 14 instructions for every non-flat
// leaf of any level, ONE block of code.  (Two earlier forms, both measured slower than the byte cells: 32-byte cells for the
// full-depth leaves only -- the waves near the surface hold full-depth leaves AND the non-flat level-8 leaves next to them, so
// both the new and the old sample ran; and 64-byte cells for every cell of a split grid's blocks -- 19 % fewer VALU
// instructions, but three 16-byte loads per lane that is inside a block: 1.7 x the tag lookups, and the waves waited.)
// The cursor shrinks to the coordinates of the last lookup: the loop keeps no bytes and no s; the shading step's gradient --
// once per wave -- and the on-a-face rule (rare) look the leaf up again in the 16-byte cells every other kernel reads.
constexpr uint32_t D4_TAG = 0x40000000u;       // 2.0f: no leaf returns a distance of 2 or more (at most 1.5 cell widths <= 1.5)
struct CursorFF {
    struct Pos { float prox; };
    static constexpr int32_t ROOT_MARK = 0x40000000;
    int32_t ax, ay, az;      // the cell coordinates of the last lookup in units of 2^-(LM - sh), or the root mark
    uint32_t sh;             // wave-uniform
    uint32_t loads;
    __device__ __forceinline__ void reset(const NodeRec &) { ax = ay = az = ROOT_MARK; sh = 0; }
};
// the 16-byte cell {LM - level | FLAT_BIT, bytes / distance, .} the cursor sits in (coordinates in the grid's own units)
__device__ __forceinline__ uint4 cell_at(const GridRef &g, int32_t Dx, int32_t Dy, int32_t Dz)
{
    const int FB = g.fine_bits;                                     // 0: a dense grid as deep as the tree
    uint4 e = reinterpret_cast<const uint4 *>(g.top)[top_index((uint32_t)Dx >> FB, (uint32_t)Dy >> FB, (uint32_t)Dz >> FB, g.level)];
    if (FB && e.x == 15u) {
        const uint32_t m = (1u << FB) - 1u;
        e = reinterpret_cast<const uint4 *>(g.fine)[e.w + fine_cell_index((uint32_t)Dx & m, (uint32_t)Dy & m, (uint32_t)Dz & m, FB, 0)];
    }
    return e;
}
__device__ __forceinline__ Cell cell_of(const CursorFF &c, const GridRef &g, const NodeRec *__restrict__ nodes)
{
    Cell k;
    const bool fresh = c.ax == CursorFF::ROOT_MARK;                 // never looked up: the root box and the root's values
    const uint4 e = fresh ? make_uint4((uint32_t)LM, nodes[0].z, nodes[0].w, 0u) : cell_at(g, c.ax, c.ay, c.az);
    const uint32_t s = e.x & 15u;
    const float q = 1.0f / 4096.0f;
    const int32_t keep = (int32_t)(0xFFFFFFFFu << s);
    const int32_t x = (int32_t)((uint32_t)c.ax << c.sh) & keep, y = (int32_t)((uint32_t)c.ay << c.sh) & keep, z = (int32_t)((uint32_t)c.az << c.sh) & keep;
    k.lx = fresh ? 0.0f : (float)x * q; k.ly = fresh ? 0.0f : (float)y * q; k.lz = fresh ? 0.0f : (float)z * q;
    k.scale = __uint_as_float((127u - LM + s) << 23);
    k.inv = __uint_as_float((127u + LM - s) << 23);
    k.v0 = e.y; k.v1 = (e.x & FLAT_BIT) ? e.y : e.z;
    return k;
}
// find() and the sample that follows it, in one piece (nothing but the distance leaves it)
template <bool FRESH>
__device__ __forceinline__ float find_sample_ff(CursorFF &c, const GridRef &g, float px, float py, float pz)
{
    const int F = g.level + g.fine_bits, sh = LM - F;
    const float unit = __uint_as_float((uint32_t)(127 + F) << 23), top = unit - 1.0f;      // 2^F, and the last cell
    const float ux = px * unit, uy = py * unit, uz = pz * unit;
    const float fm = __builtin_fminf(__builtin_fminf(__builtin_amdgcn_fractf(ux), __builtin_amdgcn_fractf(uy)), __builtin_amdgcn_fractf(uz));
    int32_t Dx, Dy, Dz;
    if (FRESH || __ballot(fm == 0.0f) == 0ull) {
        Dx = cvt_floor(__builtin_amdgcn_fmed3f(ux, 0.0f, top)); Dy = cvt_floor(__builtin_amdgcn_fmed3f(uy, 0.0f, top));
        Dz = cvt_floor(__builtin_amdgcn_fmed3f(uz, 0.0f, top));
    } else {                                                        // a lane on a cell face: the reference's rule (find_units)
        float tx, ty, tz, qx, qy, qz;
        Dx = axis_a(px, tx, qx); Dy = axis_a(py, ty, qy); Dz = axis_a(pz, tz, qz);
        const bool fresh = c.ax == CursorFF::ROOT_MARK;
        int s = LM;
        if (!fresh) s = (int)(cell_at(g, c.ax, c.ay, c.az).x & 15u);
        bool moved;
        on_face_choice(fresh ? c.ax : (int32_t)((uint32_t)c.ax << c.sh), fresh ? c.ay : (int32_t)((uint32_t)c.ay << c.sh),
                       fresh ? c.az : (int32_t)((uint32_t)c.az << c.sh), s, Dx, Dy, Dz, tx == qx, ty == qy, tz == qz, moved);
        Dx = min(max(Dx, 0), 4095) >> sh; Dy = min(max(Dy, 0), 4095) >> sh; Dz = min(max(Dz, 0), 4095) >> sh;
    }
    c.sh = (uint32_t)sh; c.ax = Dx; c.ay = Dy; c.az = Dz;
    c.loads++;
    const uint32_t w = g.d4[top_index((uint32_t)Dx, (uint32_t)Dy, (uint32_t)Dz, F)];
    if (!(__uint_as_float(w) >= 2.0f)) return __uint_as_float(w);     // a flat leaf: its distance
    c.loads++;
    // (whole 16-byte loads, each kept as ONE register tuple: constrained word by word the compiler shuffles the words back
    // into pairs for the packed fmas)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *cellp = reinterpret_cast<const u32x4 *>(g.recs) + (size_t)w;   // (g.recs is biased by the tag: no subtraction here)
    u32x4 q0 = cellp[0], q1 = cellp[1], q2 = cellp[2];
    asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2));                   // all three in flight together
    const float inv = __uint_as_float(q0.w);
    const float dx = sat(__builtin_fmaf(ux, inv, __uint_as_float(q0.x)));
    const float dy = sat(__builtin_fmaf(uy, inv, __uint_as_float(q0.y)));
    const float dz = sat(__builtin_fmaf(uz, inv, __uint_as_float(q0.z)));
    const float topL = __builtin_fmaf(dx, __uint_as_float(q2.x), __uint_as_float(q1.x));
    const float topH = __builtin_fmaf(dx, __uint_as_float(q2.y), __uint_as_float(q1.y));
    const float botL = __builtin_fmaf(dx, __uint_as_float(q2.z), __uint_as_float(q1.z));
    const float botH = __builtin_fmaf(dx, __uint_as_float(q2.w), __uint_as_float(q1.w));
    const float loadL = __builtin_fmaf(dy, botL - topL, topL);
    const float loadH = __builtin_fmaf(dy, botH - topH, topH);
    // 2 * scale = 2^(1 - level) = 2^(1 - F) / 2^-k: exponents subtract, (127 + 1 - F) - (127 - k) + 127 (multiplying by scale and then
    // by 2 rounds nowhere)
    const float scale2 = __uint_as_float(((uint32_t)(255 - F) << 23) - q0.w);
    return (lerp(loadL, loadH, dz) - 0.25f) * scale2;
}
__device__ __forceinline__ uint32_t find(CursorFF &c, const NodeRec *__restrict__, const GridRef &g, uint32_t, int32_t *__restrict__,
                                         uint32_t, float px, float py, float pz, CursorFF::Pos &u)
{
    u.prox = find_sample_ff<false>(c, g, px, py, pz);
    return 0;
}
__device__ __forceinline__ uint32_t find_fresh(CursorFF &c, const NodeRec *__restrict__, const GridRef &g, uint32_t, int32_t *__restrict__,
                                               uint32_t, float px, float py, float pz, CursorFF::Pos &u)
{
    u.prox = find_sample_ff<true>(c, g, px, py, pz);
    return 0;
}
__device__ __forceinline__ float sample_after_find(const CursorFF &, const CursorFF::Pos &u, float, float, float) { return u.prox; }

}  // namespace sdfhip
