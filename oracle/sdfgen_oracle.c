/*
 * sdfgen_oracle.c -- CPU restatement of SdfGen's point-cloud -> ASDF builder.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as sdf_oracle.c): the checker for the HIP
 * builder in sdfbox_amd/csrc/sdfgen_device.hip, never shipped, never called by it.
 *
 * PARITY UNPINNED BY THE REFERENCE: SdfGen itself cannot be built here (it includes
 * <gsl/gsl> and <windows.h>, absent from the image), it has no tests, and no mesh data
 * ships.  This file restates it function by function; the only facts of a real SdfGen
 * run available (SURVEY.md / BASELINE.md: sphere cloud r = 0.5, depth 4 -> 4 529 nodes,
 * 72 472 B, levels 1/8/64/512/3944) are checked in tests/test_sdfgen.py.
 *
 * Follows (reference paths):
 *   SdfGen/dllmain.cpp:67-80    FindDimensions     -> g_find_dimensions
 *   SdfGen/dllmain.cpp:82-87    Transform          -> g_transform
 *   SdfGen/dllmain.cpp:88-91    Inside             -> inline in g_distance_at
 *   SdfGen/dllmain.cpp:99-118   TrueDistanceAt     -> g_true_distance_at
 *   SdfGen/dllmain.cpp:119-149  DistanceAt         -> g_distance_at
 *   SdfGen/dllmain.cpp:151-162  GetPossible        -> g_get_possible
 *   SdfGen/dllmain.cpp:163-190  construct          -> g_construct
 *   SdfGen/dllmain.cpp:192-207  FromFloat/WriteBytes -> g_from_float / g_write_bytes
 *   SdfGen/dllmain.cpp:295-319  SdfGen             -> oracle_sdfgen
 *   SdfGen/math.h, math.cpp     Vector3 arithmetic (x*x + y*y + z*z, no fusing)
 *
 * Semantics worth knowing (all reproduced, none "fixed"):
 *   - the candidate list handed down the recursion is pruned with a radius that does NOT
 *     guarantee the nearest point of every corner survives, so deeper corner values come
 *     from "the nearest *surviving* point";
 *   - child i inherits its corner i from the parent instead of recomputing it;
 *   - ties in the nearest-point search go to the first point in list order;
 *   - `minDistance < 0.015` compares the float with a double literal;
 *   - an empty candidate list makes the reference throw ("Did not find"): here it is
 *     error code 2.
 * Build with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;
typedef struct { v3 pos, normal; } vertex;               /* math.h:47-51 */
typedef struct { float v[8]; } octverts;

static inline v3 v3_(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 add(v3 a, v3 b) { return v3_(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return v3_(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mulf(v3 a, float b) { return v3_(a.x * b, a.y * b, a.z * b); }
static inline v3 divf(v3 a, float b) { return v3_(a.x / b, a.y / b, a.z / b); }
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float lensq(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static inline v3 split(int i) { return v3_((float)(i % 2), (float)((i / 2) % 2), (float)((i / 2 / 2) % 2)); }

typedef struct {
    const vertex *verts;
    float global_scale;
    v3 global_offset;
    int max_depth;
    /* growing node arrays (octs / vals of the reference) */
    int32_t *parent, *children;
    octverts *vals;
    size_t n, cap;
    int error;
} gen;

static const float HalfSqrt3 = 0.866025404f;
static const float padding = 1.1f;

static size_t g_push(gen *g)
{
    if (g->n == g->cap) {
        size_t nc = g->cap ? g->cap * 2 : 1024;
        g->parent = (int32_t *)realloc(g->parent, nc * sizeof(int32_t));
        g->children = (int32_t *)realloc(g->children, nc * sizeof(int32_t));
        g->vals = (octverts *)realloc(g->vals, nc * sizeof(octverts));
        if (!g->parent || !g->children || !g->vals) { g->error = 1; return 0; }
        g->cap = nc;
    }
    g->parent[g->n] = 0; g->children[g->n] = 0;
    for (int i = 0; i < 8; i++) g->vals[g->n].v[i] = INFINITY;     /* EmptyVerts */
    return g->n++;
}

static void g_find_dimensions(gen *g, const vertex *v, uint32_t n)
{
    v3 lower = v3_(INFINITY, INFINITY, INFINITY), higher = v3_(-INFINITY, -INFINITY, -INFINITY);
    for (uint32_t i = 0; i < n; i++) {
        lower = v3_(fminf(lower.x, v[i].pos.x), fminf(lower.y, v[i].pos.y), fminf(lower.z, v[i].pos.z));
        higher = v3_(fmaxf(higher.x, v[i].pos.x), fmaxf(higher.y, v[i].pos.y), fmaxf(higher.z, v[i].pos.z));
    }
    g->global_offset = add(mulf(add(lower, higher), 0.5f), v3_(0.003f, 0.003f, 0.003f));
    float lowest = fminf(lower.x, fminf(lower.y, lower.z));
    float highest = fmaxf(higher.x, fmaxf(higher.y, higher.z));
    g->global_scale = (highest - lowest) * padding;
}

static v3 g_transform(const gen *g, v3 w)
{
    w.y = 1 - w.y;
    return add(mulf(sub(w, v3_(.5f, .5f, .5f)), g->global_scale), g->global_offset);
}

static float g_true_distance_at(gen *g, v3 p, const int32_t *list, size_t count)
{
    p = g_transform(g, p);
    float min_d = INFINITY;
    int found = 0;
    for (size_t i = 0; i < count; i++) {
        float d = lensq(sub(g->verts[list[i]].pos, p));
        if (d < min_d) { min_d = d; found = 1; }
    }
    if (!found || isinf(min_d) || isnan(min_d)) { g->error = 2; return 0.0f; }
    return sqrtf(min_d) / g->global_scale;
}

static float g_distance_at(gen *g, v3 p, const int32_t *list, size_t count)
{
    p = g_transform(g, p);
    const vertex *closest = NULL;
    float min_d = INFINITY;
    for (size_t i = 0; i < count; i++) {
        float d = lensq(sub(g->verts[list[i]].pos, p));
        if (d < min_d) { min_d = d; closest = &g->verts[list[i]]; }
    }
    if (!closest || isinf(min_d) || isnan(min_d)) { g->error = 2; return 0.0f; }
    min_d = sqrtf(min_d);
    if ((double)min_d < 0.015)
        min_d = dot(divf(closest->normal, sqrtf(lensq(closest->normal))), sub(p, closest->pos));
    else if (dot(closest->normal, sub(closest->pos, p)) > 0)      /* Inside */
        min_d *= -1;
    return min_d / g->global_scale;
}

/* -> malloc'ed list of the survivors, *out_count of them */
static int32_t *g_get_possible(gen *g, v3 pos, float min_distance, const int32_t *list, size_t count,
                               size_t *out_count)
{
    pos = g_transform(g, pos);
    min_distance *= g->global_scale;
    min_distance *= min_distance;
    int32_t *next = (int32_t *)malloc((count ? count : 1) * sizeof(int32_t));
    size_t m = 0;
    if (!next) { g->error = 1; *out_count = 0; return NULL; }
    for (size_t i = 0; i < count; i++)
        if (lensq(sub(g->verts[list[i]].pos, pos)) < min_distance) next[m++] = list[i];
    *out_count = m;
    return next;
}

static void g_construct(gen *g, const int32_t *list, size_t count, int depth, v3 pos, int parent, size_t insert)
{
    if (g->error) return;
    float scale = powf(0.5f, (float)depth);
    v3 center = add(pos, mulf(mulf(v3_(1, 1, 1), 0.5f), scale));
    float center_value = g_true_distance_at(g, center, list, count);
    if (g->error) return;
    size_t pcount = 0;
    int32_t *possible = g_get_possible(g, center, center_value + HalfSqrt3 * scale, list, count, &pcount);
    if (g->error) { free(possible); return; }
    int32_t cur_children = -1;
    for (int i = 0; i < 8; i++) {
        if (g->vals[insert].v[i] == INFINITY) {
            float d = g_distance_at(g, add(pos, mulf(split(i), scale)), possible, pcount);
            if (g->error) { free(possible); return; }
            g->vals[insert].v[i] = d;
        }
    }
    if (center_value < scale * 2 && depth < g->max_depth) {
        cur_children = (int32_t)g->n;
        for (int i = 0; i < 8; i++) {
            size_t k = g_push(g);
            if (g->error) { free(possible); return; }
            g->vals[k].v[i] = g->vals[insert].v[i];
        }
        for (int i = 0; i < 8; i++)
            g_construct(g, possible, pcount, depth + 1, add(pos, mulf(split(i), scale / 2)), (int)insert,
                        (size_t)cur_children + (size_t)i);
    }
    g->parent[insert] = parent;
    g->children[insert] = cur_children;
    free(possible);
}

static float g_saturate(float x) { return x > 1 ? 1 : (x < 0 ? 0 : x); }
static uint8_t g_from_float(float f, float scale)
{
    float normd = f / 2 / scale;
    return (uint8_t)floorf(g_saturate(normd + 0.25f) * 255);
}
static void g_write_bytes(const gen *g, uint8_t *dest, size_t p, float scale)
{
    for (int j = 0; j < 8; j++) dest[p * 8 + j] = g_from_float(g->vals[p].v[j], scale);
    if (g->children[p] != -1)
        for (int i = 0; i < 8; i++) g_write_bytes(g, dest, (size_t)g->children[p] + (size_t)i, scale / 2);
}

/* verts6: n x {position xyz, normal xyz}.  Outputs are malloc'ed: structs (count x 2 int32),
 * values (count x 8 bytes), and optionally the unquantised corner values (count x 8 floats).
 * Returns 0, 1 (out of memory) or 2 (the reference would have thrown "Did not find"). */
int oracle_sdfgen(const float *verts6, uint32_t n, int32_t depth, int32_t **structs_out,
                  uint8_t **values_out, float **float_values_out, uint32_t *count_out,
                  float *global_scale_out, float *global_offset_out)
{
    gen g;
    memset(&g, 0, sizeof g);
    g.verts = (const vertex *)verts6;
    g.max_depth = depth;
    g_find_dimensions(&g, g.verts, n);
    int32_t *all = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
    if (!all) return 1;
    for (uint32_t i = 0; i < n; i++) all[i] = (int32_t)i;
    g_push(&g);
    g_construct(&g, all, n, 0, v3_(0, 0, 0), -1, 0);
    free(all);
    if (g.error) { free(g.parent); free(g.children); free(g.vals); return g.error; }
    int32_t *s = (int32_t *)malloc(g.n * 8);
    uint8_t *v = (uint8_t *)malloc(g.n * 8);
    float *fv = float_values_out ? (float *)malloc(g.n * 8 * sizeof(float)) : NULL;
    if (!s || !v || (float_values_out && !fv)) { free(s); free(v); free(fv); free(g.parent); free(g.children); free(g.vals); return 1; }
    for (size_t i = 0; i < g.n; i++) { s[2 * i] = g.parent[i]; s[2 * i + 1] = g.children[i]; }
    g_write_bytes(&g, v, 0, 1.0f);
    if (fv) memcpy(fv, g.vals, g.n * 8 * sizeof(float));
    *structs_out = s; *values_out = v; *count_out = (uint32_t)g.n;
    if (float_values_out) *float_values_out = fv;
    if (global_scale_out) *global_scale_out = g.global_scale;
    if (global_offset_out) { global_offset_out[0] = g.global_offset.x; global_offset_out[1] = g.global_offset.y; global_offset_out[2] = g.global_offset.z; }
    free(g.parent); free(g.children); free(g.vals);
    return 0;
}

void oracle_sdfgen_free(void *p) { free(p); }
