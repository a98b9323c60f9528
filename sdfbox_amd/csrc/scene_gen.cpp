// Analytic ASDF builder: SdfGen's octree construction applied to a
// closed-form distance function.
//
// Follows SdfGen/dllmain.cpp:
//   construct  :163-190  split iff centerValue < 2*scale && depth < MaxDepth,
//                        centerValue = unsigned distance at the cell centre;
//                        on a split the 8 children are appended contiguously,
//                        then each child's subtree is built depth-first;
//                        corner i sits at pos + split(i)*scale, split(i) =
//                        (i%2, i/2%2, i/4%2) (math.h:40-43)
//   FromFloat  :192-196  byte = floor(saturate(f/2/scale + 0.25) * 255)
//   WriteBytes :197-207  per-node scale halves per level
// What differs: distances come from a formula, not from the nearest point of
// a cloud (DistanceAt :119-149), so depth-9/10 scenes build in seconds, and
// subtrees are built by several threads and spliced in the same order the
// serial recursion would have produced (the output bytes do not depend on the
// thread count).
#include "abi_guard.h"
#include <memory>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <new>
#include <thread>
#include <vector>

namespace sdfhip {

// ---- deterministic sin / cos --------------------------------------------
// Cody-Waite reduction by pi/2 in double + Taylor kernels; |x| stays < ~1e3
// here.  Built with -ffp-contract=off, so the result is a pure function of x
// on every IEEE-754 host.  Accuracy ~1e-15, far below the 8-bit quantiser.
static inline void reduce(double x, double &r, int &q)
{
    const double two_over_pi = 0.63661977236758134308;
    const double p1 = 1.57079632673412561417e+00;   // pi/2 split in three
    const double p2 = 6.07710050650619224932e-11;
    const double p3 = 2.02226624879595063154e-21;
    // round-to-nearest-even via the 2^52+2^51 trick (|x| < 2^31 here)
    const double magic = 6755399441055744.0;
    double k = (x * two_over_pi + magic) - magic;
    q = (int)((long long)k & 3);
    r = ((x - k * p1) - k * p2) - k * p3;
}
static inline double ksin(double r)
{
    double r2 = r * r;
    double p = -7.6471637318198164759e-13;        // -1/15!
    p = p * r2 + 1.6059043836821614599e-10;       //  1/13!
    p = p * r2 - 2.5052108385441718775e-08;       // -1/11!
    p = p * r2 + 2.7557319223985890653e-06;       //  1/9!
    p = p * r2 - 1.9841269841269841270e-04;       // -1/7!
    p = p * r2 + 8.3333333333333333333e-03;       //  1/5!
    p = p * r2 - 1.6666666666666666667e-01;       // -1/3!
    return r + r * r2 * p;
}
static inline double kcos(double r)
{
    double r2 = r * r;
    double p = 4.7794773323873852974e-14;         //  1/16!
    p = p * r2 - 1.1470745597729724714e-11;       // -1/14!
    p = p * r2 + 2.0876756987868098979e-09;       //  1/12!
    p = p * r2 - 2.7557319223985890653e-07;       // -1/10!
    p = p * r2 + 2.4801587301587301587e-05;       //  1/8!
    p = p * r2 - 1.3888888888888888889e-03;       // -1/6!
    p = p * r2 + 4.1666666666666666667e-02;       //  1/4!
    p = p * r2 - 0.5;
    return 1.0 + r2 * p;
}
static inline void det_sincos(double x, double &sn, double &cs)
{
    double r; int q; reduce(x, r, q);
    double s = ksin(r), c = kcos(r);
    switch (q) {
    case 0: sn = s; cs = c; break;
    case 1: sn = c; cs = -s; break;
    case 2: sn = -s; cs = -c; break;
    default: sn = -c; cs = s; break;
    }
}
double det_sin(double x) { double s, c; det_sincos(x, s, c); return s; }
double det_cos(double x) { double s, c; det_sincos(x, s, c); return c; }

// ---- shapes: signed distance (or a 1-Lipschitz lower bound) in unit-cube
// coordinates, evaluated in double, handed to the builder as float ---------
struct Shape {
    int kind;
    double p[8];
    float eval(double x, double y, double z) const
    {
        double dx = x - p[0], dy = y - p[1], dz = z - p[2];
        switch (kind) {
        case SDFHIP_SHAPE_SPHERE:
            return (float)(std::sqrt(dx * dx + dy * dy + dz * dz) - p[3]);
        case SDFHIP_SHAPE_TORUS: {
            double q = std::sqrt(dx * dx + dz * dz) - p[3];
            return (float)(std::sqrt(q * q + dy * dy) - p[4]);
        }
        default: {  // gyroid shell clipped to a sphere ("dragon stand-in")
            double f = p[4];
            double sx, cx, sy, cy, sz, cz;
            det_sincos(f * dx, sx, cx);
            det_sincos(f * dy, sy, cy);
            det_sincos(f * dz, sz, cz);
            double g = sx * cy + sy * cz + sz * cx;            // |grad g| <= f*sqrt(3)
            double shell = std::fabs(g) / (f * 1.7320508075688772) - p[5];
            double ball = std::sqrt(dx * dx + dy * dy + dz * dz) - p[3];
            return (float)(shell > ball ? shell : ball);
        }
        }
    }
};

static inline float saturate(float x) { return x > 1 ? 1 : (x < 0 ? 0 : x); }  // math.cpp:43-46
static inline uint8_t from_float(float f, float scale)                         // dllmain.cpp:192-196
{
    float normd = f / 2 / scale;
    return (uint8_t)floorf(saturate(normd + 0.25f) * 255);
}

struct Tree {
    std::vector<int32_t> structs;  // 2 per node
    std::vector<uint8_t> values;   // 8 per node
    size_t size() const { return structs.size() / 2; }
    size_t push(int32_t parent)
    {
        structs.push_back(parent); structs.push_back(-1);
        values.resize(values.size() + 8);
        return size() - 1;
    }
};

// A node at depth `seed_depth` whose subtree another thread builds.
struct Seed { size_t node; int depth; float x, y, z; Tree sub; };

// construct(), dllmain.cpp:163-190.  `insert` already exists in `t`; this
// fills its values, decides the split and recurses.  Positions are exact
// dyadic floats, as in the reference (pos + split(i) * (scale / 2)).
// With seeds != nullptr, nodes at depth == seed_depth are recorded instead of
// built (their values and subtree come from the seed's own tree later).
static void construct(const Shape &sh, Tree &t, int max_depth, int depth, float px, float py,
                      float pz, size_t insert, int seed_depth, std::vector<Seed> *seeds)
{
    if (seeds && depth == seed_depth) {
        seeds->push_back(Seed{ insert, depth, px, py, pz, Tree() });
        return;
    }
    float scale = powf(0.5f, (float)depth);
    float h = 0.5f * scale;
    float center_value = fabsf(sh.eval(px + h, py + h, pz + h));
    for (int i = 0; i < 8; i++) {
        float cx = px + (float)(i % 2) * scale, cy = py + (float)(i / 2 % 2) * scale,
              cz = pz + (float)(i / 4 % 2) * scale;
        t.values[insert * 8 + i] = from_float(sh.eval(cx, cy, cz), scale);
    }
    if (center_value < scale * 2 && depth < max_depth) {
        size_t children = t.size();
        t.structs[insert * 2 + 1] = (int32_t)children;
        for (int i = 0; i < 8; i++) t.push((int32_t)insert);
        for (int i = 0; i < 8; i++)
            construct(sh, t, max_depth, depth + 1, px + (float)(i % 2) * (scale / 2),
                      py + (float)(i / 2 % 2) * (scale / 2), pz + (float)(i / 4 % 2) * (scale / 2),
                      children + i, seed_depth, seeds);
    }
}

// Assign final indices to the top tree's nodes by replaying the serial
// recursion: a seed's descendants land where the serial build would have
// appended them, between the blocks of the top tree.
static void replay(const Tree &top, size_t node, std::vector<int32_t> &remap,
                   const std::vector<int32_t> &seed_of, const std::vector<Seed> &seeds,
                   std::vector<size_t> &seed_base, size_t &cur)
{
    int32_t s = seed_of[node];
    if (s >= 0) { seed_base[(size_t)s] = cur; cur += seeds[(size_t)s].sub.size() - 1; return; }
    int32_t c = top.structs[node * 2 + 1];
    if (c < 0) return;
    for (int i = 0; i < 8; i++) remap[(size_t)c + i] = (int32_t)cur++;
    for (int i = 0; i < 8; i++) replay(top, (size_t)c + i, remap, seed_of, seeds, seed_base, cur);
}

}  // namespace sdfhip

using namespace sdfhip;

extern "C" int sdfhip_generate(int shape, const float *params, int nparams, int max_depth,
                               int nthreads, sdfhip_octdata *out)
try {
    if (!out || !params) return fail(SDFHIP_ERR_ARG, "generate: null argument");
    out->length = 0; out->structs = nullptr; out->values = nullptr;
    static const int need[] = { 4, 5, 6 };
    if (shape < 0 || shape > 2) return fail(SDFHIP_ERR_ARG, "generate: unknown shape %d", shape);
    if (nparams != need[shape]) return fail(SDFHIP_ERR_ARG, "generate: shape %d takes %d params, got %d", shape, need[shape], nparams);
    if (max_depth < 0 || max_depth > 12) return fail(SDFHIP_ERR_ARG, "generate: max_depth %d outside 0..12", max_depth);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    Shape sh; sh.kind = shape;
    for (int i = 0; i < 8; i++) sh.p[i] = i < nparams ? (double)params[i] : 0.0;

    try {
        // Serial recursion down to seed_depth, leaving seed nodes whose
        // subtrees are built by the pool and spliced in serial DFS order.
        const int seed_depth = nthreads > 1 ? (max_depth >= 6 ? 3 : (max_depth >= 4 ? 2 : -1)) : -1;
        Tree top;
        top.push(-1);
        std::vector<Seed> seeds;
        construct(sh, top, max_depth, 0, 0, 0, 0, 0, seed_depth, seed_depth >= 0 ? &seeds : nullptr);
        if (!seeds.empty()) {
            std::vector<std::thread> pool;
            std::vector<int> err(seeds.size(), 0);
            std::atomic<size_t> cursor{0};
            auto work = [&]() {
                for (;;) {
                    size_t s = cursor.fetch_add(1);
                    if (s >= seeds.size()) return;
                    try {
                        Seed &sd = seeds[s];
                        sd.sub.push(-1);  // node 0 of the private tree is the seed itself
                        construct(sh, sd.sub, max_depth, sd.depth, sd.x, sd.y, sd.z, 0, -1, nullptr);
                    } catch (...) { err[s] = 1; }
                }
            };
            // a thread the system refuses (std::system_error) or a pool that cannot grow is not an error: the threads that did
            // start, and this one, build the seeds (which thread builds a seed changes nothing: every seed has its own tree)
            try {
                pool.reserve((size_t)nthreads);
                for (int i = 0; i < nthreads - 1; i++) pool.emplace_back(work);
            } catch (...) { }
            work();
            for (auto &th : pool) th.join();
            for (int e : err) if (e) throw std::bad_alloc();
        }

        // Splice.  Serial order: nodes of levels <= seed_depth were appended while
        // recursing; in the true serial recursion a seed's descendants are
        // appended *between* blocks of the top tree.  Reproduce that order by
        // replaying the top-level recursion and emitting blocks in DFS order.
        size_t total = top.size();
        for (auto &sd : seeds) total += sd.sub.size() - 1;
        if (total > 0xFFFFFFF0ull / 2) return fail(SDFHIP_ERR_NOMEM, "generate: %zu nodes exceed the 32-bit index space", total);
        // (owned until the hand-over at the end: the vectors of the splice below may throw)
        std::unique_ptr<int32_t, void (*)(void *)> S_own((int32_t *)malloc(total * 8), free);
        std::unique_ptr<uint8_t, void (*)(void *)> V_own((uint8_t *)malloc(total * 8), free);
        int32_t *S = S_own.get();
        uint8_t *V = V_own.get();
        if (!S || !V) return fail(SDFHIP_ERR_NOMEM, "generate: out of memory for %zu nodes", total);

        if (seeds.empty()) {
            memcpy(S, top.structs.data(), total * 8);
            memcpy(V, top.values.data(), total * 8);
        } else {
            // new index of every top-tree node, assigned by replaying DFS.
            std::vector<int32_t> remap(top.size(), -1);
            std::vector<int32_t> seed_of(top.size(), -1);
            for (size_t s = 0; s < seeds.size(); s++) seed_of[seeds[s].node] = (int32_t)s;
            std::vector<size_t> seed_base(seeds.size(), 0);  // where sub nodes 1.. land
            size_t cur = 0;
            remap[0] = (int32_t)cur++;
            replay(top, 0, remap, seed_of, seeds, seed_base, cur);
            // top nodes
            for (size_t i = 0; i < top.size(); i++) {
                size_t ni = (size_t)remap[i];
                int32_t p = top.structs[i * 2], c = top.structs[i * 2 + 1];
                S[ni * 2] = p < 0 ? -1 : remap[(size_t)p];
                S[ni * 2 + 1] = c < 0 ? -1 : remap[(size_t)c];
                memcpy(V + ni * 8, &top.values[i * 8], 8);
            }
            // seeds: sub node 0 is the seed (already placed, take its values
            // and children from the sub tree); sub node k>=1 -> seed_base+k-1.
            for (size_t s = 0; s < seeds.size(); s++) {
                const Tree &sub = seeds[s].sub;
                size_t root_new = (size_t)remap[seeds[s].node];
                size_t base = seed_base[s];
                auto map = [&](int32_t k) -> int32_t { return k == 0 ? (int32_t)root_new : (int32_t)(base + (size_t)k - 1); };
                memcpy(V + root_new * 8, &sub.values[0], 8);
                S[root_new * 2 + 1] = sub.structs[1] < 0 ? -1 : map(sub.structs[1]);
                for (size_t k = 1; k < sub.size(); k++) {
                    size_t nk = base + k - 1;
                    S[nk * 2] = map(sub.structs[k * 2]);
                    S[nk * 2 + 1] = sub.structs[k * 2 + 1] < 0 ? -1 : map(sub.structs[k * 2 + 1]);
                    memcpy(V + nk * 8, &sub.values[k * 8], 8);
                }
            }
        }
        out->length = (uint32_t)total; out->structs = S_own.release(); out->values = V_own.release();
        return SDFHIP_OK;
    } catch (const std::bad_alloc &) {
        return fail(SDFHIP_ERR_NOMEM, "generate: out of memory");
    }
}
SDFHIP_ABI_CATCH(sdfhip_generate)
