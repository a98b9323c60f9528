// Host half of libsdfhip.so under AddressSanitizer + UBSan (CPU build only: GPU ASan is
// not available on the pool).  Built and run by tests/test_host_sanitizers.py.
#include "sdfhip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d %s (%s)\n", __FILE__, __LINE__, #c, sdfhip_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    const char *tmp = argc > 1 ? argv[1] : "/tmp/sdfhip_sanitize.asdf";
    // builders, serial and threaded, every shape
    const float sphere[] = {0.5f, 0.5f, 0.5f, 0.3f}, torus[] = {0.5f, 0.5f, 0.5f, 0.25f, 0.09f},
                gyroid[] = {0.5f, 0.5f, 0.5f, 0.42f, 37.699112f, 0.004f};
    sdfhip_octdata a{}, b{}, c{};
    REQUIRE(sdfhip_generate(SDFHIP_SHAPE_SPHERE, sphere, 4, 4, 1, &a) == SDFHIP_OK);
    REQUIRE(sdfhip_generate(SDFHIP_SHAPE_TORUS, torus, 5, 6, 3, &b) == SDFHIP_OK);
    REQUIRE(sdfhip_generate(SDFHIP_SHAPE_GYROID, gyroid, 6, 6, 4, &c) == SDFHIP_OK);
    REQUIRE(a.length == 3465 && b.length > 1000 && c.length > 100000);
    uint32_t depth = 0; int cons = 0;
    REQUIRE(sdfhip_octdata_validate(c.structs, c.length, &depth, &cons) == SDFHIP_OK && depth == 6 && cons == 1);
    // file round trip
    REQUIRE(sdfhip_asdf_save(&c, tmp) == SDFHIP_OK);
    sdfhip_octdata d{};
    REQUIRE(sdfhip_asdf_load(tmp, &d) == SDFHIP_OK && d.length == c.length);
    REQUIRE(memcmp(d.structs, c.structs, (size_t)c.length * 8) == 0 && memcmp(d.values, c.values, (size_t)c.length * 8) == 0);
    // error paths must not leak or touch freed memory
    sdfhip_octdata e{};
    REQUIRE(sdfhip_asdf_load("/nonexistent/x.asdf", &e) == SDFHIP_ERR_IO && e.structs == nullptr);
    REQUIRE(sdfhip_generate(9, sphere, 4, 4, 1, &e) == SDFHIP_ERR_ARG);
    REQUIRE(sdfhip_generate(SDFHIP_SHAPE_SPHERE, sphere, 3, 4, 1, &e) == SDFHIP_ERR_ARG);
    std::vector<int32_t> bad(c.structs, c.structs + (size_t)c.length * 2);
    bad[2 * 7 + 1] = (int32_t)c.length - 2;
    REQUIRE(sdfhip_octdata_validate(bad.data(), c.length, &depth, &cons) == SDFHIP_ERR_BAD_TREE);
    bad = std::vector<int32_t>(c.structs, c.structs + (size_t)c.length * 2);
    bad[2 * 9] = 3;                                   // in range, inconsistent
    REQUIRE(sdfhip_octdata_validate(bad.data(), c.length, &depth, &cons) == SDFHIP_OK && cons == 0);
    // camera block
    sdfhip_info info;
    sdfhip_info_default(&info, 1920.0f, 1080.0f);
    sdfhip_info_set_heading(&info, -0.2f, 0.35f);
    sdfhip_info_set_position(&info, 0.5f, 0.5f, -0.35f);
    REQUIRE(info.limit > 2.3f && info.limit < 2.35f && info.heading[0][3] == 0.0f);
    sdfhip_octdata_free(&a); sdfhip_octdata_free(&b); sdfhip_octdata_free(&c); sdfhip_octdata_free(&d);
    sdfhip_octdata_free(&d);                          // idempotent
    remove(tmp);
    puts("host sanitizer run ok");
    return 0;
}
