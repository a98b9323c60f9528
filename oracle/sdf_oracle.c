/*
 * sdf_oracle.c -- CPU restatement of SdfBox's ray-march compute pass.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sdfbox_amd/ may include, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker / the timed CPU
 * baseline -- never as the thing shipped.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference has no CPU implementation
 * of this path, no tests, no golden vectors, and its
 * only implementation is HLSL run through D3D11 texture hardware.  This file
 * is a scalar fp32 restatement of that HLSL, statement by statement, and is
 * the specification the HIP kernels are held to bit-for-bit.  It is anchored
 * by closed-form cases in tests/test_oracle.py, nothing more.
 *
 * Follows (all paths relative to the reference checkout):
 *   SdfBox/Shaders/Compute.hlsl:15-29    sam / sample_at      -> o_sample_at
 *   SdfBox/Shaders/Compute.hlsl:31-58    Cube                 -> o_cube, o_*
 *   SdfBox/Shaders/Compute.hlsl:88-108   find                 -> o_find
 *   SdfBox/Shaders/Compute.hlsl:112-130  gradient             -> o_gradient
 *   SdfBox/Shaders/Compute.hlsl:163-168  ray                  -> o_ray
 *   SdfBox/Shaders/Compute.hlsl:180-231  main                 -> o_pixel
 *   SdfBox/Logic.cs:407-463              Info / Float3x3      -> o_info
 *   SdfBox/Program.cs:514-538            value texture layout -> o_sample_at
 *   SdfBox/Shaders/DisplayFrag.hlsl:16-24 display pass         -> oracle_display
 *
 * Arithmetic contract (shared with the HIP kernels, DESIGN.md "Numerics"):
 *   - every operation is an individually rounded IEEE fp32 operation in the
 *     order written here; build with -ffp-contract=off, no fast-math: the
 *     compiler fuses nothing on its own;
 *   - the multiply-adds that HLSL compiles to `mad` / `dp3` are written as
 *     explicit fused fmaf (D3D11 leaves mad's fusing to the implementation):
 *       lerp(a,b,t)   = fmaf(t, b - a, a)
 *       dot(a,b)      = fmaf(a.z, b.z, fmaf(a.y, b.y, a.x*b.x))
 *       pos += dir*s  = fmaf(dir, s, pos)
 *   - saturate(x)   = fminf(fmaxf(x, 0), 1)           (NaN -> 0, as HLSL)
 *   - normalize(v)  = v * (1.0f / sqrtf(dot(v,v)))    (HLSL: v * rsqrt(dot(v,v)))
 *   - length(v)     = sqrtf(dot(v,v))
 *   - R8_UNorm texel = (float)byte / 255.0f, bilinear = x-lerp then y-lerp
 *   - exp2(strength) is evaluated once per frame on the host (exp2f).
 *
 * CONTRACT VARIANTS (measurement only; DESIGN.md section 2 "how far is the contract from other
 * readings of the shader").  The contract above is one legal reading of D3D11's `mad` and of its
 * texture sampler; nothing in the reference pins it.  Four build-time variants restate the other
 * readings so that tests/test_oracle_variants.py can put a number on the distance between them (the
 * parity statement of SURVEY.md 8d: RGB within 1e-5 and equal step count, fraction of pixels):
 *   -DO_UNFUSED       SURVEY.md 8c's literal text: lerp = a + t*(b-a), dot and pos += dir*s with
 *                     separately rounded multiplies and adds (no fused operation anywhere);
 *   -DO_LERP_MATHCS   lerp as the reference's own (uncalled) CPU helper writes it, SdfBox/Math.cs:25-28:
 *                     a*(1-p) + b*p, unfused;
 *   -DO_SAMPLER8      the bilinear weights of the two texture taps (Compute.hlsl:15-29, sampler
 *                     Program.cs:147) reduced to 8 fractional bits, as D3D11 hardware filtering
 *                     does at its minimum precision (the z-lerp stays in shader arithmetic);
 *   -DO_RSQRT1ULP     HLSL's normalize is v * rsqrt(dot(v,v)), and a GPU's rsqrt is an approximation (D3D11 allows
 *                     it one ulp): every reciprocal square root one ulp off the correctly rounded 1/sqrtf(x), up or
 *                     down by the lowest bit of x -- ray directions, light directions and normals all move.
 * The default build (none of them) is the oracle; the variants are never compared with the kernels.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE            /* sched_getaffinity / pthread_setaffinity_np: oracle_bench_rows only */
#endif
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- Info, 112 bytes, Logic.cs:407-420 / Compute.hlsl:70-81 ------------- */
typedef struct {
    float heading[3][4];   /* offset 0: three float4 rows, Logic.cs:427-463 */
    float position[3];     /* 48 */
    float margin;          /* 60 */
    float screen_size[2];  /* 64 */
    uint32_t buffer_size;  /* 72 (ignored: the scene's own length is used) */
    float limit;           /* 76 */
    float light[3];        /* 80 */
    float strength;        /* 92 */
    float fov;             /* 96 */
    int32_t hidef;         /* 100 (unused by the shader) */
    uint32_t pad[2];       /* -> 112 */
} o_info;

typedef struct {
    const int32_t *structs;  /* N x {parent, children}, Program.cs:339-350 */
    const uint8_t *values;   /* N x 8 corner bytes, k = x + 2y + 4z */
    uint32_t n;
} o_scene;

/* Cube, Compute.hlsl:31-61 */
typedef struct { float lx, ly, lz, scale; } o_cube;

/* per-pixel state: the shader's static `index` and `box` (Compute.hlsl:12,61) */
typedef struct {
    const o_scene *sc;
    uint32_t index;
    o_cube box;
    uint64_t n_nodes;    /* node records read by find(): SURVEY 8d */
    uint64_t n_samples;  /* interpol_world calls */
} o_ctx;

/* Every helper is forced inline so that the two clones of the exported entry
 * points (plain x86-64 and +fma, where fmaf is one vfmadd instruction;
 * elsewhere it is libm's correctly rounded fmaf: same bits) carry their own copy. */
#define O_INLINE static inline __attribute__((always_inline))
#define O_CLONES __attribute__((target_clones("default", "fma")))

O_INLINE float o_sat(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
#if defined(O_UNFUSED) || defined(O_LERP_MATHCS)
/* contract variants (see the header): every multiply and add rounded on its own */
O_INLINE float o_mad(float a, float b, float c) { return a * b + c; }      /* -ffp-contract=off: two roundings */
#if defined(O_LERP_MATHCS)
O_INLINE float o_lerp(float a, float b, float t) { return a * (1.0f - t) + b * t; }    /* Math.cs:25-28 */
#else
O_INLINE float o_lerp(float a, float b, float t) { return a + t * (b - a); }
#endif
O_INLINE float o_dot(float ax, float ay, float az, float bx, float by, float bz)
{
    return (ax * bx + ay * by) + az * bz;
}
#else
O_INLINE float o_mad(float a, float b, float c) { return fmaf(a, b, c); }
O_INLINE float o_lerp(float a, float b, float t) { return fmaf(t, b - a, a); }
O_INLINE float o_dot(float ax, float ay, float az, float bx, float by, float bz)
{
    return fmaf(az, bz, fmaf(ay, by, ax * bx));
}
#endif
#if defined(O_RSQRT1ULP)
O_INLINE float o_rsqrt(float x)
{
    union { float f; uint32_t u; } in = { x }, r = { 1.0f / sqrtf(x) };
    if (r.f == r.f && r.u != 0 && (r.u & 0x7F800000u) != 0x7F800000u) r.u += (in.u & 1u) ? 1u : 0xFFFFFFFFu;     /* finite, non-zero: +- 1 ulp */
    return r.f;
}
#else
O_INLINE float o_rsqrt(float x) { return 1.0f / sqrtf(x); }
#endif
#if defined(O_SAMPLER8)
/* a bilinear weight as the texture unit sees it at D3D11's minimum filtering precision: 8 fractional bits */
O_INLINE float o_weight(float w) { return floorf(w * 256.0f + 0.5f) / 256.0f; }
#else
O_INLINE float o_weight(float w) { return w; }
#endif

/* Cube::scale_up, Compute.hlsl:36-40 */
O_INLINE void o_scale_up(o_cube *b)
{
    b->scale *= 2.0f;
    b->lx = floorf(b->lx / b->scale) * b->scale;
    b->ly = floorf(b->ly / b->scale) * b->scale;
    b->lz = floorf(b->lz / b->scale) * b->scale;
}
/* Cube::scale_down, Compute.hlsl:41-45 */
O_INLINE void o_scale_down(o_cube *b, int dx, int dy, int dz)
{
    b->scale /= 2.0f;
    b->lx += (float)dx * b->scale;
    b->ly += (float)dy * b->scale;
    b->lz += (float)dz * b->scale;
}
/* Cube::inside, Compute.hlsl:50-53 */
O_INLINE int o_inside(const o_cube *b, float px, float py, float pz)
{
    float hx = b->lx + b->scale, hy = b->ly + b->scale, hz = b->lz + b->scale;
    return (b->lx <= px && b->ly <= py && b->lz <= pz) &&
           (px <= hx && py <= hy && pz <= hz);
}

/* Analysis hook (scripts/descent_levels.py): when set, every node record a descent loads is
 * counted by the octree level of that node (1..12).  NULL in every other use. */
static uint64_t *g_level_hist;
void oracle_level_histogram(uint64_t *hist16) { g_level_hist = hist16; }

/* find, Compute.hlsl:88-108 */
O_INLINE void o_find(o_ctx *t, float px, float py, float pz)
{
    const int32_t *S = t->sc->structs;
    int iterations = 0;
    int32_t c_parent = S[2 * (size_t)t->index], c_children = S[2 * (size_t)t->index + 1];
    t->n_nodes++;

    while (!o_inside(&t->box, px, py, pz) && c_parent >= 0) {
        t->index = (uint32_t)c_parent;
        c_parent = S[2 * (size_t)t->index];
        c_children = S[2 * (size_t)t->index + 1];
        t->n_nodes++;
        o_scale_up(&t->box);
    }
    while (t->index < t->sc->n && iterations < 12 && c_children >= 0) {
        /* (int3) saturate((pos - box.lower) / box.scale * subdiv) */
        int dx = (int)o_sat((px - t->box.lx) / t->box.scale * 2.0f);
        int dy = (int)o_sat((py - t->box.ly) / t->box.scale * 2.0f);
        int dz = (int)o_sat((pz - t->box.lz) / t->box.scale * 2.0f);
        int p = dx + 2 * dy + 4 * dz;
        t->index = (uint32_t)(c_children + p);
        c_parent = S[2 * (size_t)t->index];
        c_children = S[2 * (size_t)t->index + 1];
        t->n_nodes++;
        o_scale_down(&t->box, dx, dy, dz);
        iterations++;
        if (__builtin_expect(g_level_hist != NULL, 0))
            __atomic_fetch_add(&g_level_hist[(-ilogbf(t->box.scale)) & 15], 1, __ATOMIC_RELAXED);
    }
}

/* The 8 texels of the current node as R8_UNorm floats.  Texture layout
 * (Program.cs:514-538): row0 = corners {0,1,4,5}, row1 = {2,3,6,7}; one
 * bilinear tap at (d.xy + p) blends corners 0..3, the tap at +(2,0) blends
 * corners 4..7 (Compute.hlsl:19-29). */
O_INLINE void o_texels(const o_ctx *t, float v[8])
{
    const uint8_t *b = t->sc->values + 8 * (size_t)t->index;
    for (int k = 0; k < 8; k++) v[k] = (float)b[k] / 255.0f;
}
O_INLINE float o_bilerp(float t00, float t10, float t01, float t11, float wx, float wy)
{
    wx = o_weight(wx); wy = o_weight(wy);          /* identity in the oracle; see O_SAMPLER8 */
    float top = o_lerp(t00, t10, wx);
    float bot = o_lerp(t01, t11, wx);
    return o_lerp(top, bot, wy);
}

/* sample_at, Compute.hlsl:19-29 */
O_INLINE float o_sample_at(o_ctx *t, float dx, float dy, float dz, float scale)
{
    float v[8];
    o_texels(t, v);
    float loadL = o_bilerp(v[0], v[1], v[2], v[3], dx, dy);
    float loadH = o_bilerp(v[4], v[5], v[6], v[7], dx, dy);
    float result = (o_lerp(loadL, loadH, dz) - 0.25f) * scale * 2.0f;
    return result;
}
/* Cube::interpol_world, Compute.hlsl:54-58 */
O_INLINE float o_interpol_world(o_ctx *t, float px, float py, float pz)
{
    float dx = o_sat((px - t->box.lx) / t->box.scale);
    float dy = o_sat((py - t->box.ly) / t->box.scale);
    float dz = o_sat((pz - t->box.lz) / t->box.scale);
    t->n_samples++;
    return o_sample_at(t, dx, dy, dz, t->box.scale);
}

/* gradient, Compute.hlsl:112-130.  A tap at integer x (or y) has bilinear
 * weight 0 towards its neighbour, i.e. lerp(a, b, 0) = a exactly. */
O_INLINE void o_gradient(const o_ctx *t, float px, float py, float pz, float g[3])
{
    float dx = o_sat((px - t->box.lx) / t->box.scale);
    float dy = o_sat((py - t->box.ly) / t->box.scale);
    float dz = o_sat((pz - t->box.lz) / t->box.scale);
    float v[8];
    o_texels(t, v);

    const float wx = o_weight(dx), wy = o_weight(dy);   /* texture weights (identity in the oracle) */
    float xl = o_lerp(o_lerp(v[0], v[2], wy), o_lerp(v[4], v[6], wy), dz);
    float xh = o_lerp(o_lerp(v[1], v[3], wy), o_lerp(v[5], v[7], wy), dz);

    float yl = o_lerp(o_lerp(v[0], v[1], wx), o_lerp(v[4], v[5], wx), dz);
    float yh = o_lerp(o_lerp(v[2], v[3], wx), o_lerp(v[6], v[7], wx), dz);

    float zl = o_bilerp(v[0], v[1], v[2], v[3], dx, dy);
    float zh = o_bilerp(v[4], v[5], v[6], v[7], dx, dy);

    g[0] = xh - xl;
    g[1] = yh - yl;
    g[2] = zh - zl;
}

/* ray, Compute.hlsl:163-168 */
O_INLINE void o_ray(const o_info *inf, uint32_t cx, uint32_t cy, float dir[3])
{
    float sx = (float)cx / inf->screen_size[1] - inf->screen_size[0] / inf->screen_size[1] * 0.5f;
    float sy = (float)cy / inf->screen_size[1] - 0.5f;
    float vx = sx * inf->fov, vy = sy * inf->fov, vz = 0.5f;
    /* mul(v, heading): component j = dot(v, j-th float4 row of Info) */
    float d0 = o_dot(vx, vy, vz, inf->heading[0][0], inf->heading[0][1], inf->heading[0][2]);
    float d1 = o_dot(vx, vy, vz, inf->heading[1][0], inf->heading[1][1], inf->heading[1][2]);
    float d2 = o_dot(vx, vy, vz, inf->heading[2][0], inf->heading[2][1], inf->heading[2][2]);
    /* normalize(v) = v * rsqrt(dot(v, v)) in HLSL; here rsqrt(x) = 1 / sqrtf(x) */
    float rl = o_rsqrt(o_dot(d0, d1, d2, d0, d1, d2));
    dir[0] = d0 * rl;
    dir[1] = d1 * rl;
    dir[2] = d2 * rl;
}

/* main, Compute.hlsl:180-231.  out = rgba; counters accumulate in t. */
O_INLINE void o_pixel(const o_scene *sc, const o_info *inf, float exp2_strength_m1,
                    uint32_t cx, uint32_t cy, float out[4], uint64_t cnt[4])
{
    o_ctx t;
    t.sc = sc;
    t.index = 0;
    t.box.lx = t.box.ly = t.box.lz = 0.0f;
    t.box.scale = 1.0f;
    t.n_nodes = t.n_samples = 0;

    float px = inf->position[0], py = inf->position[1], pz = inf->position[2];
    float dir[3];
    o_ray(inf, cx, cy, dir);
    float prox = 1.0f;
    const float margin = inf->margin;

    int i, j = 0;
    for (i = 0; (prox > margin * 2.0f || prox < 0.0f) && i < 100; i++) {
        if (o_dot(px, py, pz, px, py, pz) > inf->limit) {
            out[0] = 0.005f; out[1] = 0.01f; out[2] = 0.2f; out[3] = (float)i;
            goto done;
        }
        o_find(&t, px, py, pz);
        prox = o_interpol_world(&t, px, py, pz);
        px = o_mad(dir[0], prox, px);
        py = o_mad(dir[1], prox, py);
        pz = o_mad(dir[2], prox, pz);
    }
    {
        /* dir = normalize(inf.light - pos); pos += dir * inf.margin; */
        float lx = inf->light[0] - px, ly = inf->light[1] - py, lz = inf->light[2] - pz;
        float rl = o_rsqrt(o_dot(lx, ly, lz, lx, ly, lz));
        dir[0] = lx * rl; dir[1] = ly * rl; dir[2] = lz * rl;
        px = o_mad(dir[0], margin, px);
        py = o_mad(dir[1], margin, py);
        pz = o_mad(dir[2], margin, pz);
        /* angle = dot(dir, normalize(gradient(pos))) */
        float g[3];
        o_gradient(&t, px, py, pz, g);
        float rg = o_rsqrt(o_dot(g[0], g[1], g[2], g[0], g[1], g[2]));
        float angle = o_dot(dir[0], dir[1], dir[2], g[0] * rg, g[1] * rg, g[2] * rg);
        if (angle < 0.0f) {
            out[0] = out[1] = out[2] = 0.0f; out[3] = (float)i;
            goto done;
        }
        /* dist = length(inf.light - pos) / 2 */
        lx = inf->light[0] - px; ly = inf->light[1] - py; lz = inf->light[2] - pz;
        float dist = sqrtf(o_dot(lx, ly, lz, lx, ly, lz)) / 2.0f;
        cnt[3]++;                          /* a shadow ray is cast */
        for (j = 0; j < 40 && prox > -margin; j++) {
            if (prox > dist || (px < 0.0f || py < 0.0f || pz < 0.0f) ||
                (px > 1.0f || py > 1.0f || pz > 1.0f)) {
                float attenuation = angle / (dist * dist) * exp2_strength_m1;
                out[0] = out[1] = out[2] = attenuation; out[3] = (float)(i + j);
                goto done;
            }
            if (prox < margin) {
                o_gradient(&t, px, py, pz, g);
                if (o_dot(g[0], g[1], g[2], dir[0], dir[1], dir[2]) < 0.0f)
                    break;
            }
            o_find(&t, px, py, pz);
            prox = o_interpol_world(&t, px, py, pz);
            float step = prox + margin;
            px = o_mad(dir[0], step, px);
            py = o_mad(dir[1], step, py);
            pz = o_mad(dir[2], step, pz);
        }
        out[0] = out[1] = out[2] = 0.0f; out[3] = (float)(i + j);
    }
done:
    cnt[0] += t.n_nodes;
    cnt[1] += t.n_samples;
    cnt[2] += (uint64_t)(i + j);
}

/* ---- path-traced mode (BASELINE config 5) ---------------
 * NOT in the reference: its README lists "path tracing" under plans only.  This
 * function DEFINES the mode; the HIP kernel k_path is held to it bit for bit.  It is
 * built from the reference's own pieces -- the primary march (Compute.hlsl:194-203),
 * the shading step and the shadow march towards the point light (:205-230) -- chained
 * by cosine-weighted diffuse bounces:
 *   per pixel, spp samples; per sample a jittered camera ray, then up to
 *   1 + max_bounces segments.  A segment marches like the primary march; on escape it
 *   adds throughput * sky (0.005, 0.01, 0.2) and the sample ends; on a hit it shades
 *   exactly like main() (offset towards the light, gradient normal, shadow march) and
 *   adds throughput * albedo * angle / dist^2 * (exp2(strength) - 1) when lit; then it
 *   bounces: dir = normalize(n' + u), n' = the normal facing the incoming ray, u uniform
 *   on the unit sphere (rejection in the cube, at most 8 tries), pos += n' * 4 * margin,
 *   throughput *= albedo.  The cursor (index, box) carries over between segments, as it
 *   does between the primary and the shadow march in the shader.
 * RNG: PCG hash chained over (seed + pixel, sample, bounce, draw); uniform = top 24 bits
 * * 2^-24.  No transcendental function anywhere, so parity stays bit-exact.
 * out = mean radiance (r, g, b), alpha = march steps of all segments and samples. */
O_INLINE uint32_t o_pcg(uint32_t v)
{
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
O_INLINE float o_rnd(uint32_t seed, uint32_t p, uint32_t s, uint32_t b, uint32_t d)
{
    uint32_t h = o_pcg(o_pcg(o_pcg(o_pcg(seed + p) + s) + b) + d);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
/* ray() of Compute.hlsl:163-168 through a fractional pixel coordinate */
O_INLINE void o_ray_f(const o_info *inf, float fx, float fy, float dir[3])
{
    float sx = fx / inf->screen_size[1] - inf->screen_size[0] / inf->screen_size[1] * 0.5f;
    float sy = fy / inf->screen_size[1] - 0.5f;
    float vx = sx * inf->fov, vy = sy * inf->fov, vz = 0.5f;
    float d0 = o_dot(vx, vy, vz, inf->heading[0][0], inf->heading[0][1], inf->heading[0][2]);
    float d1 = o_dot(vx, vy, vz, inf->heading[1][0], inf->heading[1][1], inf->heading[1][2]);
    float d2 = o_dot(vx, vy, vz, inf->heading[2][0], inf->heading[2][1], inf->heading[2][2]);
    float rl = o_rsqrt(o_dot(d0, d1, d2, d0, d1, d2));
    dir[0] = d0 * rl; dir[1] = d1 * rl; dir[2] = d2 * rl;
}

O_INLINE void o_pixel_pt(const o_scene *sc, const o_info *inf, float k, uint32_t spp,
                         uint32_t max_bounces, uint32_t seed, float albedo, uint32_t frame_w,
                         uint32_t cx, uint32_t cy, float out[4], uint64_t cnt[4])
{
    o_ctx t;
    t.sc = sc;
    t.n_nodes = t.n_samples = 0;
    const float margin = inf->margin;
    const uint32_t p = cy * frame_w + cx;
    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
    uint64_t steps = 0;
    for (uint32_t s = 0; s < spp; s++) {
        t.index = 0;
        t.box.lx = t.box.ly = t.box.lz = 0.0f;
        t.box.scale = 1.0f;
        float px = inf->position[0], py = inf->position[1], pz = inf->position[2];
        float dir[3];
        o_ray_f(inf, (float)cx + o_rnd(seed, p, s, 0, 0), (float)cy + o_rnd(seed, p, s, 0, 1), dir);
        float T = 1.0f;
        for (uint32_t b = 0;; b++) {
            float prox = 1.0f;
            int i, escaped = 0;
            for (i = 0; (prox > margin * 2.0f || prox < 0.0f) && i < 100; i++) {
                if (o_dot(px, py, pz, px, py, pz) > inf->limit) { escaped = 1; break; }
                o_find(&t, px, py, pz);
                prox = o_interpol_world(&t, px, py, pz);
                px = fmaf(dir[0], prox, px);
                py = fmaf(dir[1], prox, py);
                pz = fmaf(dir[2], prox, pz);
            }
            steps += (uint64_t)i;
            if (escaped) {
                acc0 = fmaf(T, 0.005f, acc0); acc1 = fmaf(T, 0.01f, acc1); acc2 = fmaf(T, 0.2f, acc2);
                break;
            }
            /* shade, Compute.hlsl:205-213 */
            float lx = inf->light[0] - px, ly = inf->light[1] - py, lz = inf->light[2] - pz;
            float rl = o_rsqrt(o_dot(lx, ly, lz, lx, ly, lz));
            float L0 = lx * rl, L1 = ly * rl, L2 = lz * rl;
            px = fmaf(L0, margin, px); py = fmaf(L1, margin, py); pz = fmaf(L2, margin, pz);
            float g[3];
            o_gradient(&t, px, py, pz, g);
            float rg = o_rsqrt(o_dot(g[0], g[1], g[2], g[0], g[1], g[2]));
            float n0 = g[0] * rg, n1 = g[1] * rg, n2 = g[2] * rg;
            float angle = o_dot(L0, L1, L2, n0, n1, n2);
            if (!(angle < 0.0f)) {
                /* shadow march, Compute.hlsl:213-230, on a copy of pos */
                float sx = px, sy = py, sz = pz, sprox = prox;
                lx = inf->light[0] - px; ly = inf->light[1] - py; lz = inf->light[2] - pz;
                float dist = sqrtf(o_dot(lx, ly, lz, lx, ly, lz)) / 2.0f;
                int j, lit = 0;
                cnt[3]++;                  /* a shadow ray is cast */
                for (j = 0; j < 40 && sprox > -margin; j++) {
                    if (sprox > dist || (sx < 0.0f || sy < 0.0f || sz < 0.0f) ||
                        (sx > 1.0f || sy > 1.0f || sz > 1.0f)) { lit = 1; break; }
                    if (sprox < margin) {
                        float gg[3];
                        o_gradient(&t, sx, sy, sz, gg);
                        if (o_dot(gg[0], gg[1], gg[2], L0, L1, L2) < 0.0f) break;
                    }
                    o_find(&t, sx, sy, sz);
                    sprox = o_interpol_world(&t, sx, sy, sz);
                    float st = sprox + margin;
                    sx = fmaf(L0, st, sx); sy = fmaf(L1, st, sy); sz = fmaf(L2, st, sz);
                }
                steps += (uint64_t)j;
                if (lit) {
                    float e = T * (albedo * (angle / (dist * dist) * k));
                    acc0 += e; acc1 += e; acc2 += e;
                }
            }
            if (b == max_bounces) break;
            /* diffuse bounce */
            if (o_dot(n0, n1, n2, dir[0], dir[1], dir[2]) > 0.0f) { n0 = -n0; n1 = -n1; n2 = -n2; }
            float u0 = n0, u1 = n1, u2 = n2, q = 1.0f;
            for (uint32_t a = 0; a < 8; a++) {
                float c0 = o_rnd(seed, p, s, b + 1, 3 * a) * 2.0f - 1.0f;
                float c1 = o_rnd(seed, p, s, b + 1, 3 * a + 1) * 2.0f - 1.0f;
                float c2 = o_rnd(seed, p, s, b + 1, 3 * a + 2) * 2.0f - 1.0f;
                float qq = o_dot(c0, c1, c2, c0, c1, c2);
                if (qq <= 1.0f && qq > 1e-12f) { u0 = c0; u1 = c1; u2 = c2; q = qq; break; }
            }
            float ru = o_rsqrt(q);
            float d0 = fmaf(u0, ru, n0), d1 = fmaf(u1, ru, n1), d2 = fmaf(u2, ru, n2);
            float qd = o_dot(d0, d1, d2, d0, d1, d2);
            if (!(qd > 1e-12f)) { d0 = n0; d1 = n1; d2 = n2; qd = o_dot(n0, n1, n2, n0, n1, n2); }
            float rd = o_rsqrt(qd);
            dir[0] = d0 * rd; dir[1] = d1 * rd; dir[2] = d2 * rd;
            float off = margin * 4.0f;
            px = fmaf(n0, off, px); py = fmaf(n1, off, py); pz = fmaf(n2, off, pz);
            T *= albedo;
        }
    }
    float inv = (float)spp;
    out[0] = acc0 / inv; out[1] = acc1 / inv; out[2] = acc2 / inv;
    out[3] = (float)steps;
    cnt[0] += t.n_nodes;
    cnt[1] += t.n_samples;
    cnt[2] += steps;
}

/* ---- exported entry points --------------------------------------------- */

typedef struct {
    o_scene sc;
    o_info inf;
    float k;
    uint32_t W, row0, nrows, row_step;
    int tid, nthreads;
    float *rgba;          /* nrows x W x 4, row-major, row 0 = global row0 */
    uint32_t *pix_nodes;  /* optional: nrows x W algorithmic node reads per pixel */
    uint64_t cnt[4];
    uint32_t pt_spp, pt_bounces, pt_seed, frame_w;   /* path-traced mode when pt_spp > 0 */
    float pt_albedo;
} o_job;

O_CLONES static void *o_worker_impl(void *arg)
{
    o_job *jb = (o_job *)arg;
    for (uint32_t r = (uint32_t)jb->tid; r < jb->nrows; r += (uint32_t)jb->nthreads) {
        for (uint32_t x = 0; x < jb->W; x++) {
            uint64_t before = jb->cnt[0];
            if (jb->pt_spp)
                o_pixel_pt(&jb->sc, &jb->inf, jb->k, jb->pt_spp, jb->pt_bounces, jb->pt_seed, jb->pt_albedo,
                           jb->frame_w, x, jb->row0 + r * jb->row_step,
                           jb->rgba + 4 * ((size_t)r * jb->W + x), jb->cnt);
            else
                o_pixel(&jb->sc, &jb->inf, jb->k, x, jb->row0 + r * jb->row_step,
                        jb->rgba + 4 * ((size_t)r * jb->W + x), jb->cnt);
            if (jb->pix_nodes)
                jb->pix_nodes[(size_t)r * jb->W + x] = (uint32_t)(jb->cnt[0] - before);
        }
    }
    return NULL;
}

/* Render the nrows rows y = row0 + r*row_step (r = 0..nrows-1) of a W-wide
 * frame into compact rows of `rgba`.  Pixel (x, y) uses the dispatch-thread id
 * (x, y) exactly as Compute.hlsl:180 does.  counters[0..3]
 * receive node reads, samples, march steps and shadow rays cast, summed over the
 * rendered pixels (may be NULL).  Rows are interleaved over `nthreads` pthreads. */
static void *o_worker(void *arg) { return o_worker_impl(arg); }

static int o_render_rows(const int32_t *structs, const uint8_t *values, uint32_t n,
                         const void *info112, uint32_t W, uint32_t row0, uint32_t nrows,
                         uint32_t row_step, float *rgba, uint64_t *counters, uint32_t *pix_nodes, int nthreads,
                         uint32_t pt_spp, uint32_t pt_bounces, uint32_t pt_seed, float pt_albedo)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    o_job *jobs = (o_job *)calloc((size_t)nthreads, sizeof(o_job));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    if (!jobs || !th) { free(jobs); free(th); return -1; }
    o_info inf;
    memcpy(&inf, info112, sizeof inf);
    float k = exp2f(inf.strength) - 1.0f;
    for (int t = 0; t < nthreads; t++) {
        jobs[t].sc.structs = structs; jobs[t].sc.values = values; jobs[t].sc.n = n;
        jobs[t].inf = inf; jobs[t].k = k;
        jobs[t].W = W; jobs[t].row0 = row0; jobs[t].nrows = nrows; jobs[t].row_step = row_step ? row_step : 1;
        jobs[t].tid = t; jobs[t].nthreads = nthreads;
        jobs[t].rgba = rgba; jobs[t].pix_nodes = pix_nodes;
        jobs[t].pt_spp = pt_spp; jobs[t].pt_bounces = pt_bounces; jobs[t].pt_seed = pt_seed;
        jobs[t].pt_albedo = pt_albedo; jobs[t].frame_w = W;
    }
    if (nthreads == 1) {
        o_worker(&jobs[0]);
    } else {
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, o_worker, &jobs[t]);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    }
    if (counters) {
        counters[0] = counters[1] = counters[2] = counters[3] = 0;
        for (int t = 0; t < nthreads; t++)
            for (int c = 0; c < 4; c++) counters[c] += jobs[t].cnt[c];
    }
    free(jobs); free(th);
    return 0;
}

int oracle_render_rows(const int32_t *structs, const uint8_t *values, uint32_t n,
                       const void *info112, uint32_t W, uint32_t row0, uint32_t nrows,
                       uint32_t row_step, float *rgba, uint64_t *counters, uint32_t *pix_nodes, int nthreads)
{
    return o_render_rows(structs, values, n, info112, W, row0, nrows, row_step, rgba, counters, pix_nodes,
                         nthreads, 0, 0, 0, 0.0f);
}

/* Path-traced mode (see o_pixel_pt).  The pixel index that seeds the RNG is y * W + x. */
int oracle_render_rows_pt(const int32_t *structs, const uint8_t *values, uint32_t n,
                          const void *info112, uint32_t W, uint32_t row0, uint32_t nrows,
                          uint32_t row_step, uint32_t spp, uint32_t max_bounces, uint32_t seed,
                          float albedo, float *rgba, uint64_t *counters, int nthreads)
{
    if (spp == 0) return -1;
    return o_render_rows(structs, values, n, info112, W, row0, nrows, row_step, rgba, counters, NULL,
                         nthreads, spp, max_bounces, seed, albedo);
}

/* ---- the timed CPU baseline (bench.py's cpu_baseline leg) ------------------------------------------------
 * The same o_pixel over the same rows as oracle_render_rows, arranged so that what is timed is the march and nothing
 * else: the worker threads exist, are pinned and have their data in place BEFORE the clock starts, the clock is read
 * inside this function between two barriers, and the work is dealt dynamically.
 *   - threads are pinned one per CPU of the caller's affinity mask, dealt round robin over the NUMA nodes those CPUs
 *     belong to (/sys/devices/system/node/node<k>/cpulist; one node when that cannot be read), within a node in the
 *     order of the list -- on the usual enumeration a core's first hardware thread comes before any second one;
 *   - flags & 1: every node's first thread makes that node's own copy of the two scene arrays (first touch by a thread
 *     that runs there), and the node's threads read that copy: no thread walks the tree through the socket interconnect;
 *   - pixels are dealt in chunks of 64 consecutive pixels of a row from one atomic counter (rows differ tenfold in
 *     cost: sky against grazing rays), so no thread waits for a neighbour's expensive rows;
 *   - `rgba` (nrows x W x 4, may be NULL: then nothing is stored) is written by the workers themselves -- fresh pages
 *     are first touched by the thread that fills them;
 *   - `repeat` > 1 renders the sample that many times over (the counters count every pass): a host whose cgroup grants CPU time
 *     in 100 ms periods needs a timed region of seconds to show its steady rate, and a frame has only so many rows.
 * -> 0, *seconds = wall time between the barriers, counters[4] as oracle_render_rows, topo[4] = {threads started,
 * NUMA nodes used, CPUs in the affinity mask, scene copies made}. */
typedef struct {
    o_scene sc;
    o_info inf;
    float k;
    uint32_t W, row0, nrows, row_step;
    float *rgba;
    uint64_t cnt[4];
    int cpu, node, leader;
    struct o_bench_shared *sh;
    char pad[64];
} o_bench_job;

struct o_bench_shared {
    pthread_barrier_t bar;                            /* initialised for the threads that really started, before `go` is set */
    pthread_mutex_t mu; pthread_cond_t cv; int go;
    uint64_t next __attribute__((aligned(128)));      /* next chunk */
    uint64_t nchunks, chunks_per_row, chunks_per_pass;     /* nchunks = repeat x chunks_per_pass */
    const int32_t *node_structs[64];
    const uint8_t *node_values[64];
    int copy, failed;
    const int32_t *structs; const uint8_t *values; uint32_t n;
    struct timespec t0, t1;
};

static int o_parse_cpulist(const char *path, cpu_set_t *set)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char buf[4096];
    size_t len = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[len] = 0;
    CPU_ZERO(set);
    for (char *p = buf; *p && *p != '\n';) {
        char *e;
        long a = strtol(p, &e, 10), b = a;
        if (e == p) break;
        if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) CPU_SET((int)c, set);
        p = (*e == ',') ? e + 1 : e;
    }
    return 0;
}

O_CLONES static void *o_bench_worker_impl(void *arg)
{
    o_bench_job *jb = (o_bench_job *)arg;
    struct o_bench_shared *sh = jb->sh;
    pthread_mutex_lock(&sh->mu);
    while (!sh->go) pthread_cond_wait(&sh->cv, &sh->mu);
    pthread_mutex_unlock(&sh->mu);
    if (jb->cpu >= 0) {
        cpu_set_t one;
        CPU_ZERO(&one); CPU_SET(jb->cpu, &one);
        (void)pthread_setaffinity_np(pthread_self(), sizeof one, &one);
    }
    if (sh->copy && jb->leader) {                       /* this node's copy of the scene, first touched here */
        size_t bs = (size_t)sh->n * 8;
        int32_t *s = (int32_t *)malloc(bs);
        uint8_t *v = (uint8_t *)malloc(bs);
        if (s && v) { memcpy(s, sh->structs, bs); memcpy(v, sh->values, bs); sh->node_structs[jb->node] = s; sh->node_values[jb->node] = v; }
        else { free(s); free(v); sh->failed = 1; }
    }
    pthread_barrier_wait(&sh->bar);                     /* every copy is in place */
    if (sh->copy && sh->node_structs[jb->node]) { jb->sc.structs = sh->node_structs[jb->node]; jb->sc.values = sh->node_values[jb->node]; }
    if (pthread_barrier_wait(&sh->bar) == PTHREAD_BARRIER_SERIAL_THREAD) clock_gettime(CLOCK_MONOTONIC, &sh->t0);
    pthread_barrier_wait(&sh->bar);                     /* ---- the clock runs from here ---- */
    float scratch[4];
    for (;;) {
        uint64_t c = __atomic_fetch_add(&sh->next, 1, __ATOMIC_RELAXED);
        if (c >= sh->nchunks) break;
        c %= sh->chunks_per_pass;                       /* (a repeated pass renders the same pixels again) */
        uint32_t r = (uint32_t)(c / sh->chunks_per_row), x0 = (uint32_t)(c % sh->chunks_per_row) * 64u;
        uint32_t x1 = x0 + 64u < jb->W ? x0 + 64u : jb->W;
        for (uint32_t x = x0; x < x1; x++)
            o_pixel(&jb->sc, &jb->inf, jb->k, x, jb->row0 + r * jb->row_step,
                    jb->rgba ? jb->rgba + 4 * ((size_t)r * jb->W + x) : scratch, jb->cnt);
    }
    if (pthread_barrier_wait(&sh->bar) == PTHREAD_BARRIER_SERIAL_THREAD) clock_gettime(CLOCK_MONOTONIC, &sh->t1);
    pthread_barrier_wait(&sh->bar);
    return NULL;
}
static void *o_bench_worker(void *arg) { return o_bench_worker_impl(arg); }

int oracle_bench_rows(const int32_t *structs, const uint8_t *values, uint32_t n, const void *info112, uint32_t W,
                      uint32_t row0, uint32_t nrows, uint32_t row_step, float *rgba, uint64_t *counters, int nthreads,
                      int flags, int repeat, double *seconds, int *topo)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 1024) nthreads = 1024;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return -2;
    /* the allowed CPUs node by node */
    static int cpus[64][CPU_SETSIZE];
    int ncpu[64], nnodes = 0, nallowed = CPU_COUNT(&allowed);
    cpu_set_t seen;
    CPU_ZERO(&seen);
    for (int k = 0; k < 64; k++) {
        char path[96];
        cpu_set_t ns;
        snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", k);
        if (o_parse_cpulist(path, &ns) != 0) break;
        int m = 0;
        for (int c = 0; c < CPU_SETSIZE; c++)
            if (CPU_ISSET(c, &ns) && CPU_ISSET(c, &allowed) && !CPU_ISSET(c, &seen)) { cpus[nnodes][m++] = c; CPU_SET(c, &seen); }
        if (m) ncpu[nnodes++] = m;
    }
    if (CPU_COUNT(&seen) != nallowed) {                 /* no (or an incomplete) node map: one node, every allowed CPU */
        nnodes = 1; ncpu[0] = 0;
        for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &allowed)) cpus[0][ncpu[0]++] = c;
    }
    o_bench_job *jobs = NULL;
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    struct o_bench_shared *sh = NULL;
    if (posix_memalign((void **)&jobs, 128, (size_t)nthreads * sizeof(o_bench_job)) != 0) jobs = NULL;
    if (posix_memalign((void **)&sh, 128, sizeof *sh) != 0) sh = NULL;
    if (!jobs || !th || !sh) { free(jobs); free(th); free(sh); return -1; }
    memset(jobs, 0, (size_t)nthreads * sizeof(o_bench_job));
    memset(sh, 0, sizeof *sh);
    o_info inf;
    memcpy(&inf, info112, sizeof inf);
    const float k = exp2f(inf.strength) - 1.0f;
    sh->copy = (flags & 1) && nnodes > 1;
    sh->structs = structs; sh->values = values; sh->n = n;
    sh->chunks_per_row = (W + 63u) / 64u;
    sh->chunks_per_pass = (uint64_t)nrows * sh->chunks_per_row;
    sh->nchunks = sh->chunks_per_pass * (uint64_t)(repeat < 1 ? 1 : repeat);
    pthread_mutex_init(&sh->mu, NULL); pthread_cond_init(&sh->cv, NULL);
    int used_nodes = 0, node_has_leader[64] = {0};
    for (int t = 0; t < nthreads; t++) {
        const int node = t % nnodes, slot = t / nnodes;
        jobs[t].sc.structs = structs; jobs[t].sc.values = values; jobs[t].sc.n = n;
        jobs[t].inf = inf; jobs[t].k = k;
        jobs[t].W = W; jobs[t].row0 = row0; jobs[t].nrows = nrows; jobs[t].row_step = row_step ? row_step : 1;
        jobs[t].rgba = rgba; jobs[t].sh = sh;
        jobs[t].node = node; jobs[t].cpu = cpus[node][slot % ncpu[node]];
        jobs[t].leader = !node_has_leader[node];
        if (jobs[t].leader) { node_has_leader[node] = 1; used_nodes++; }
    }
    int started = 0;
    for (; started < nthreads; started++)
        if (pthread_create(&th[started], NULL, o_bench_worker, &jobs[started]) != 0) break;
    if (started == 0) { free(jobs); free(th); free(sh); return -3; }
    nthreads = started;                                 /* (a thread limit: the run goes on with the threads there are) */
    pthread_barrier_init(&sh->bar, NULL, (unsigned)nthreads);
    pthread_mutex_lock(&sh->mu); sh->go = 1; pthread_cond_broadcast(&sh->cv); pthread_mutex_unlock(&sh->mu);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    if (seconds) *seconds = (double)(sh->t1.tv_sec - sh->t0.tv_sec) + 1e-9 * (double)(sh->t1.tv_nsec - sh->t0.tv_nsec);
    if (counters) {
        counters[0] = counters[1] = counters[2] = counters[3] = 0;
        for (int t = 0; t < nthreads; t++)
            for (int c = 0; c < 4; c++) counters[c] += jobs[t].cnt[c];
    }
    int copies = 0;
    for (int k2 = 0; k2 < 64; k2++)
        if (sh->node_structs[k2]) { free((void *)sh->node_structs[k2]); free((void *)sh->node_values[k2]); copies++; }
    if (topo) { topo[0] = nthreads; topo[1] = used_nodes; topo[2] = nallowed; topo[3] = copies; }
    const int failed = sh->failed;
    pthread_barrier_destroy(&sh->bar); pthread_mutex_destroy(&sh->mu); pthread_cond_destroy(&sh->cv);
    free(jobs); free(th); free(sh);
    return failed ? -1 : 0;
}

/* One pixel, for unit tests.  out[4] = rgba, cnt[4] = nodes, samples, steps, shadow rays. */
O_CLONES void oracle_pixel(const int32_t *structs, const uint8_t *values, uint32_t n,
                  const void *info112, uint32_t x, uint32_t y, float *out, uint64_t *cnt)
{
    o_scene sc = { structs, values, n };
    o_info inf;
    memcpy(&inf, info112, sizeof inf);
    cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
    o_pixel(&sc, &inf, exp2f(inf.strength) - 1.0f, x, y, out, cnt);
}

/* Trilinear distance at a world position starting from the root cursor:
 * find() + interpol_world().  For the analytic anchors in the tests. */
O_CLONES float oracle_distance_at(const int32_t *structs, const uint8_t *values, uint32_t n,
                         float x, float y, float z, uint32_t *leaf_index, float *leaf_scale)
{
    o_scene sc = { structs, values, n };
    o_ctx t;
    t.sc = &sc; t.index = 0; t.box.lx = t.box.ly = t.box.lz = 0.0f; t.box.scale = 1.0f;
    t.n_nodes = t.n_samples = 0;
    o_find(&t, x, y, z);
    if (leaf_index) *leaf_index = t.index;
    if (leaf_scale) *leaf_scale = t.box.scale;
    return o_interpol_world(&t, x, y, z);
}

/* The 256 R8_UNorm decode values, so the GPU's reciprocal-and-fixup decode
 * can be checked exhaustively against the division it stands for. */
void oracle_unorm_table(float *out256)
{
    for (int b = 0; b < 256; b++) out256[b] = (float)b / 255.0f;
}

/* ---- display pass: SdfBox/Shaders/DisplayFrag.hlsl:16-24 ------------------------
 * The fragment shader samples the result texture at the fragment's own texel and
 * returns pow(val, 1/2.2) (all four channels), or with `debug` the heat map
 * float4(1,1,1,0) * val.w / 140; the swap chain is R8G8B8A8_UNorm, so the output is
 * converted as D3D11 converts render-target output: NaN -> 0, clamp to [0,1], scale by
 * 255, round to nearest.  out = 4 bytes per pixel, R,G,B,A. */
static uint8_t o_unorm8(float c)
{
    return (uint8_t)(fminf(fmaxf(c, 0.0f), 1.0f) * 255.0f + 0.5f);
}
void oracle_display(const float *rgba, uint64_t npix, int debug, uint8_t *out)
{
    for (uint64_t i = 0; i < npix; i++) {
        const float *v = rgba + 4 * i;
        uint8_t *o = out + 4 * i;
        if (debug) {
            float h = 1.0f * v[3] / 140.0f;
            o[0] = o[1] = o[2] = o_unorm8(h);
            o[3] = o_unorm8(0.0f * v[3] / 140.0f);
        } else {
            for (int c = 0; c < 4; c++) o[c] = o_unorm8(powf(v[c], 1.0f / 2.2f));
        }
    }
}
