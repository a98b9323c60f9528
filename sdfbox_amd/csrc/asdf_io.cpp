// .asdf reader / writer and structural validation.
//
// File format (SdfGen/dllmain.cpp:250-292): 'a','s','d','f', uint32 N
// (little-endian), N x {int32 Parent, int32 Children}, N x uint8[8]; no
// version field (dllmain.cpp:285-286 is commented out).  8 + 16*N bytes.
#include "abi_guard.h"
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace sdfhip;

extern "C" int sdfhip_asdf_load(const char *path, sdfhip_octdata *out)
try {
    if (!path || !out) return fail(SDFHIP_ERR_ARG, "asdf_load: null argument");
    out->length = 0; out->structs = nullptr; out->values = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return fail(SDFHIP_ERR_IO, "asdf_load: could not open %s", path);
    unsigned char hdr[8];
    if (fread(hdr, 1, 8, f) != 8 || memcmp(hdr, "asdf", 4) != 0) {
        fclose(f);
        return fail(SDFHIP_ERR_IO, "asdf_load: %s is not a supported ASDF file", path);
    }
    uint32_t n = (uint32_t)hdr[4] | ((uint32_t)hdr[5] << 8) | ((uint32_t)hdr[6] << 16) |
                 ((uint32_t)hdr[7] << 24);
    if (n == 0) { fclose(f); return fail(SDFHIP_ERR_IO, "asdf_load: %s holds zero nodes", path); }
    int32_t *s = (int32_t *)malloc((size_t)n * 8);
    uint8_t *v = (uint8_t *)malloc((size_t)n * 8);
    if (!s || !v) { free(s); free(v); fclose(f); return fail(SDFHIP_ERR_NOMEM, "asdf_load: out of memory for %u nodes", n); }
    bool ok = fread(s, 8, n, f) == n && fread(v, 8, n, f) == n;
    fclose(f);
    if (!ok) { free(s); free(v); return fail(SDFHIP_ERR_IO, "asdf_load: %s is shorter than its header says (%u nodes)", path, n); }
    out->length = n; out->structs = s; out->values = v;
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_asdf_load)

extern "C" int sdfhip_asdf_save(const sdfhip_octdata *d, const char *path)
try {
    if (!d || !path || !d->structs || !d->values || d->length == 0)
        return fail(SDFHIP_ERR_ARG, "asdf_save: null or empty octdata");
    FILE *f = fopen(path, "wb");
    if (!f) return fail(SDFHIP_ERR_IO, "asdf_save: could not open %s for saving", path);
    unsigned char hdr[8] = { 'a', 's', 'd', 'f', (unsigned char)(d->length & 0xFF),
                             (unsigned char)((d->length >> 8) & 0xFF),
                             (unsigned char)((d->length >> 16) & 0xFF),
                             (unsigned char)((d->length >> 24) & 0xFF) };
    bool ok = fwrite(hdr, 1, 8, f) == 8 && fwrite(d->structs, 8, d->length, f) == d->length &&
              fwrite(d->values, 8, d->length, f) == d->length;
    ok = (fclose(f) == 0) && ok;
    if (!ok) return fail(SDFHIP_ERR_IO, "asdf_save: short write to %s", path);
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_asdf_save)

extern "C" void sdfhip_octdata_free(sdfhip_octdata *d)
try {
    if (!d) return;
    free(d->structs); free(d->values);
    d->structs = nullptr; d->values = nullptr; d->length = 0;
}
SDFHIP_ABI_CATCH_VOID(sdfhip_octdata_free)

// Bounds check (mandatory for any kernel: no out-of-range node index is ever
// dereferenced on the GPU) + depth + parent/child consistency (gates the
// cursor-stack kernel, which derives parents from its own descent path).
extern "C" int sdfhip_octdata_validate(const int32_t *structs, uint32_t n, uint32_t *depth_out,
                                       int *consistent_out)
try {
    if (!structs || n == 0) return fail(SDFHIP_ERR_ARG, "validate: null or empty structs");
    bool consistent = structs[0] < 0;
    for (uint32_t i = 0; i < n; i++) {
        int32_t p = structs[2 * (size_t)i], c = structs[2 * (size_t)i + 1];
        if (p >= 0 && (uint32_t)p >= n)
            return fail(SDFHIP_ERR_BAD_TREE, "node %u: parent %d out of range (N=%u)", i, p, n);
        if (c >= 0 && ((uint64_t)c + 8 > n))
            return fail(SDFHIP_ERR_BAD_TREE, "node %u: children %d..%d out of range (N=%u)", i, c, c + 7, n);
        if (c >= 0) {
            if (c == 0) consistent = false;
            for (int k = 0; k < 8; k++)
                if (structs[2 * ((size_t)c + k)] != (int32_t)i) consistent = false;
        }
    }
    // Parent chains must end at a node without parent after a bounded number of links.
    // The shader's ascend loop (Compute.hlsl:93-97) follows `parent` until the box contains
    // the position or parent < 0; for a position outside the root cube it walks the whole
    // chain, so a cycle would never terminate (a GPU hang) and a chain of millions of links
    // would take seconds per march step.  No octree the shader can descend (12 levels) needs
    // more than a few links: chains longer than MAX_CHAIN are rejected.
    {
        const uint32_t MAX_CHAIN = 64;
        std::vector<uint8_t> links;             // links to the chain's end, 0xFF = unknown
        try { links.assign(n, 0xFF); } catch (const std::bad_alloc &) {
            return fail(SDFHIP_ERR_NOMEM, "validate: out of memory");
        }
        std::vector<uint32_t> path;
        for (uint32_t i = 0; i < n; i++) {
            if (links[i] != 0xFF) continue;
            path.clear();
            uint32_t j = i;
            uint32_t base = 0;
            for (;;) {
                if (links[j] != 0xFF) { base = links[j]; break; }
                int32_t p = structs[2 * (size_t)j];
                path.push_back(j);
                if (p < 0) { base = 0; path.pop_back(); links[j] = 0; break; }
                if (path.size() > MAX_CHAIN + 1)
                    return fail(SDFHIP_ERR_BAD_TREE, "node %u: parent chain longer than %u links or cyclic", i, MAX_CHAIN);
                j = (uint32_t)p;
            }
            // path holds i .. (node before j), nearest-to-end last
            for (size_t k = path.size(); k-- > 0;) {
                base += 1;
                if (base > MAX_CHAIN)
                    return fail(SDFHIP_ERR_BAD_TREE, "node %u: parent chain longer than %u links", path[k], MAX_CHAIN);
                links[path[k]] = (uint8_t)base;
            }
        }
    }
    // Depth by following parent links needs a consistent tree; for an
    // inconsistent one report "unknown" as 0xFFFFFFFF.
    uint32_t depth = 0xFFFFFFFFu;
    if (consistent) {
        std::vector<uint8_t> lvl;
        try { lvl.assign(n, 0xFF); } catch (const std::bad_alloc &) {
            return fail(SDFHIP_ERR_NOMEM, "validate: out of memory");
        }
        // children always sit after their parent in a consistent tree built
        // by append; do not rely on it: iterate until every node is levelled.
        lvl[0] = 0; depth = 0;
        std::vector<uint32_t> stack; stack.push_back(0);
        size_t seen = 0;
        while (!stack.empty()) {
            uint32_t i = stack.back(); stack.pop_back(); seen++;
            int32_t c = structs[2 * (size_t)i + 1];
            if (c < 0) continue;
            if (lvl[i] >= 250) { consistent = false; break; }
            for (int k = 0; k < 8; k++) {
                if (lvl[(size_t)c + k] != 0xFF) { consistent = false; break; }  // shared block / cycle
                lvl[(size_t)c + k] = (uint8_t)(lvl[i] + 1);
                if ((uint32_t)lvl[i] + 1 > depth) depth = (uint32_t)lvl[i] + 1;
                stack.push_back((uint32_t)c + k);
            }
            if (!consistent) break;
        }
        if (!consistent) depth = 0xFFFFFFFFu;
        (void)seen;  // unreachable nodes are harmless: no cursor can reach them
    }
    if (depth_out) *depth_out = depth;
    if (consistent_out) *consistent_out = consistent ? 1 : 0;
    return SDFHIP_OK;
}
SDFHIP_ABI_CATCH(sdfhip_octdata_validate)
