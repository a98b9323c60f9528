// Point-cloud readers for the ASDF builder: what SdfGen's LoadPly / LoadObj accept.
//
//   sdfhip_load_ply   SdfGen/dllmain.cpp:244-248 -> ply_reader.cpp:35-71: binary
//                     little-endian PLY whose vertex element comes first and holds
//                     exactly 6 floats (position, normal) per vertex (README.md:18-19).
//                     ASCII and big-endian files are refused, as there.
//   sdfhip_load_obj   SdfGen/dllmain.cpp:237-242 -> obj_reader.cpp:45-96: `v`, `vn`,
//                     `f v/t/n` or `v//n` (a face assigns its normals to its vertices;
//                     vertices no face mentions keep the normal (0,0,0)); `#`, `o`, `s`,
//                     `vt` lines are skipped; anything else is an error.
// Out-of-range face indices are errors here (the reference indexes without a check).
#include "abi_guard.h"
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <string>
#include <vector>

using namespace sdfhip;

static int hand_over(const std::vector<float> &v, sdfhip_points *out)
{
    out->count = (uint32_t)(v.size() / 6);
    out->data = (float *)malloc(v.empty() ? 4 : v.size() * sizeof(float));
    if (!out->data) return fail(SDFHIP_ERR_NOMEM, "point reader: out of memory");
    if (!v.empty()) memcpy(out->data, v.data(), v.size() * sizeof(float));
    return SDFHIP_OK;
}

extern "C" void sdfhip_points_free(sdfhip_points *p)
try {
    if (!p) return;
    free(p->data);
    p->data = nullptr; p->count = 0;
}
SDFHIP_ABI_CATCH_VOID(sdfhip_points_free)

extern "C" int sdfhip_load_ply(const char *path, sdfhip_points *out)
try {
    if (!path || !out) return fail(SDFHIP_ERR_ARG, "load_ply: null argument");
    out->count = 0; out->data = nullptr;
    std::ifstream file(path, std::ios::binary);
    if (!file.is_open()) return fail(SDFHIP_ERR_IO, "load_ply: could not open %s", path);
    file.exceptions(std::ios::badbit);            // (an extraction that runs out of memory must not read as "end of file": see load_obj)
    std::string word;
    auto bad = [&]() { return fail(SDFHIP_ERR_IO, "load_ply: %s: file format is unsupported or invalid", path); };
    if (!(file >> word) || word != "ply") return bad();
    if (!(file >> word) || word != "format") return bad();
    if (!(file >> word)) return bad();
    if (word == "ascii") return fail(SDFHIP_ERR_IO, "load_ply: %s: ASCII PLY is not supported (binary_little_endian only)", path);
    if (word == "binary_big_endian") return fail(SDFHIP_ERR_IO, "load_ply: %s: big-endian PLY is not supported", path);
    if (word != "binary_little_endian") return bad();
    auto search = [&](const char *w) { while (file >> word) if (word == w) return true; return false; };
    if (!search("vertex")) return bad();
    long long count = -1;
    if (!(file >> count) || count < 0 || count > 0x7FFFFFFFll / 24) return bad();
    if (!search("end_header")) return bad();
    file.get();                                   // the newline that ends the header
    try {
        std::vector<float> v((size_t)count * 6);
        file.read((char *)v.data(), (std::streamsize)(v.size() * sizeof(float)));
        if ((size_t)file.gcount() != v.size() * sizeof(float))
            return fail(SDFHIP_ERR_IO, "load_ply: %s holds fewer than %lld vertices of 6 floats", path, count);
        return hand_over(v, out);
    } catch (const std::bad_alloc &) {
        return fail(SDFHIP_ERR_NOMEM, "load_ply: out of memory for %lld vertices", count);
    }
}
SDFHIP_ABI_CATCH(sdfhip_load_ply)

extern "C" int sdfhip_load_obj(const char *path, sdfhip_points *out)
try {
    if (!path || !out) return fail(SDFHIP_ERR_ARG, "load_obj: null argument");
    out->count = 0; out->data = nullptr;
    std::ifstream file(path);
    if (!file.is_open()) return fail(SDFHIP_ERR_IO, "load_obj: could not open %s", path);
    // iostreams swallow an exception inside an extraction and set badbit: a std::getline that cannot grow its string would end the
    // loop below like the end of the file, and a truncated cloud would be handed over as a success (found by
    // tests/host_fault_injection.cpp).  With badbit in the mask the stream throws the original exception again.
    file.exceptions(std::ios::badbit);
    try {
        std::vector<float> verts;      // 6 per vertex
        std::vector<float> normals;    // 3 per normal
        std::string line;
        long lineno = 0;
        while (std::getline(file, line)) {
            lineno++;
            size_t i = 0;
            while (i < line.size() && isspace((unsigned char)line[i])) i++;
            if (i == line.size()) continue;
            const char *s = line.c_str() + i;
            auto bad = [&]() { return fail(SDFHIP_ERR_IO, "load_obj: %s:%ld: file format is unsupported or invalid", path, lineno); };
            if (s[0] == '#' || s[0] == 'o' || s[0] == 's') continue;
            if (s[0] == 'v' && s[1] == 't') continue;
            if (s[0] == 'v' && (s[1] == ' ' || s[1] == 'n')) {
                char *end = nullptr;
                const char *p = s + 2;
                float f[3];
                for (int k = 0; k < 3; k++) { f[k] = strtof(p, &end); if (end == p) return bad(); p = end; }
                if (s[1] == ' ') { verts.insert(verts.end(), { f[0], f[1], f[2], 0.0f, 0.0f, 0.0f }); }
                else normals.insert(normals.end(), { f[0], f[1], f[2] });
                continue;
            }
            if (s[0] == 'f') {
                const char *p = s + 1;
                for (;;) {
                    while (*p == ' ' || *p == '\t' || *p == '\r') p++;
                    if (!*p) break;
                    char *end = nullptr;
                    long vi = strtol(p, &end, 10);
                    if (end == p || *end != '/') return bad();
                    p = end + 1;
                    if (*p != '/') { (void)strtol(p, &end, 10); if (end == p) return bad(); p = end; }
                    if (*p != '/') return bad();
                    p++;
                    long ni = strtol(p, &end, 10);
                    if (end == p) return bad();
                    p = end;
                    if (vi < 1 || (size_t)vi > verts.size() / 6 || ni < 1 || (size_t)ni > normals.size() / 3)
                        return fail(SDFHIP_ERR_IO, "load_obj: %s:%ld: face index out of range", path, lineno);
                    memcpy(&verts[(size_t)(vi - 1) * 6 + 3], &normals[(size_t)(ni - 1) * 3], 3 * sizeof(float));
                }
                continue;
            }
            return bad();
        }
        return hand_over(verts, out);
    } catch (const std::bad_alloc &) {
        return fail(SDFHIP_ERR_NOMEM, "load_obj: out of memory");
    } catch (const std::ios_base::failure &) {
        return fail(SDFHIP_ERR_IO, "load_obj: %s: read error", path);
    }
}
SDFHIP_ABI_CATCH(sdfhip_load_obj)
