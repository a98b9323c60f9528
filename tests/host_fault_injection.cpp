// The exception firewall of the C ABI under injected failures ("never throw/abort across the boundary"; what it
// guards against is the reference's own behaviour, SdfGen/pch.h:20-26: a throw that crosses P/Invoke ends the host).
// CPU build of the host half of the library; built and run by tests/test_host_sanitizers.py.
//
// Two injectors, both in this executable (so they win over libstdc++'s / libc's definitions at link time):
//   * operator new that throws std::bad_alloc at its k-th call from now (k counts down in THIS thread and in the threads the
//     library starts: one global atomic) -- every entry point that allocates is called with k = 0, 1, 2, ... until a call gets
//     through without meeting the countdown; each call must RETURN: SDFHIP_ERR_NOMEM with a message, or SDFHIP_OK;
//   * pthread_create that refuses (EAGAIN) while a flag is up -- std::thread then throws std::system_error; the threaded builder
//     must still return SDFHIP_OK with the same bytes (the calling thread builds the seeds).
// A std::terminate anywhere ends this program through the handler below with exit code 97 and the entry point's name.
#include "sdfhip.h"
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <exception>
#include <new>
#include <pthread.h>
#include <vector>

static std::atomic<long> g_countdown{-1};        // < 0: off; 0: the next allocation throws
static std::atomic<long> g_thrown{0};
static std::atomic<bool> g_no_threads{false};
static const char *g_entry = "(none)";

static void *counted_alloc(size_t n)
{
    long c = g_countdown.load(std::memory_order_relaxed);
    while (c >= 0) {
        if (g_countdown.compare_exchange_weak(c, c - 1)) {
            if (c == 0) { g_thrown.fetch_add(1); throw std::bad_alloc(); }
            break;
        }
    }
    void *p = malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void *operator new(size_t n) { return counted_alloc(n); }
void *operator new[](size_t n) { return counted_alloc(n); }
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, size_t) noexcept { free(p); }
void operator delete[](void *p, size_t) noexcept { free(p); }

extern "C" int pthread_create(pthread_t *t, const pthread_attr_t *a, void *(*fn)(void *), void *arg)
{
    using fn_t = int (*)(pthread_t *, const pthread_attr_t *, void *(*)(void *), void *);
    static fn_t real = reinterpret_cast<fn_t>(dlsym(RTLD_NEXT, "pthread_create"));
    if (g_no_threads.load()) return EAGAIN;
    return real(t, a, fn, arg);
}

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d %s (entry %s, last error: %s)\n", __FILE__, __LINE__, #c, g_entry, sdfhip_last_error()); exit(1); } } while (0)

// call(): one invocation of an entry point -> its status; clean(): free what a successful call handed over.
// -> how many of the calls met an injected failure
template <class Call, class Clean> static long sweep(const char *entry, Call call, Clean clean, long limit = 4000)
{
    g_entry = entry;
    long failures = 0;
    for (long k = 0; k < limit; k++) {
        g_thrown.store(0);
        g_countdown.store(k);
        const int rc = call();
        g_countdown.store(-1);
        if (g_thrown.load() == 0) {                       // the countdown outlived the call: every allocation of it has been failed once
            REQUIRE(rc == SDFHIP_OK);
            clean();
            printf("%-28s %ld of its allocations failed one by one: %ld x SDFHIP_ERR_NOMEM, then SDFHIP_OK\n", entry, k, failures);
            return failures;
        }
        // an allocation failed inside the call: a code and a message, or -- where the library copes (a thread pool that cannot
        // grow) -- success
        if (rc != SDFHIP_OK) {
            REQUIRE(rc == SDFHIP_ERR_NOMEM);
            REQUIRE(strlen(sdfhip_last_error()) > 0);
            failures++;
        } else {
            clean();
        }
    }
    fprintf(stderr, "FAILED: %s still allocating after %ld injected failures\n", entry, limit);
    exit(1);
}

int main(int argc, char **argv)
{
    std::set_terminate([] { fprintf(stderr, "FAILED: std::terminate inside %s\n", g_entry); _Exit(97); });
    const char *dir = argc > 1 ? argv[1] : "/tmp";
    char obj_path[512], ply_path[512], asdf_path[512];
    snprintf(obj_path, sizeof obj_path, "%s/fault.obj", dir);
    snprintf(ply_path, sizeof ply_path, "%s/fault.ply", dir);
    snprintf(asdf_path, sizeof asdf_path, "%s/fault.asdf", dir);

    // inputs, made with the injectors off
    const float gyroid[] = {0.5f, 0.5f, 0.5f, 0.42f, 37.699112f, 0.004f};
    sdfhip_octdata ref{};
    REQUIRE(sdfhip_generate(SDFHIP_SHAPE_GYROID, gyroid, 6, 4, 1, &ref) == SDFHIP_OK && ref.length > 500);
    REQUIRE(sdfhip_asdf_save(&ref, asdf_path) == SDFHIP_OK);
    {
        FILE *f = fopen(obj_path, "w");
        REQUIRE(f);
        fprintf(f, "# fault injection\no cloud\n");
        for (int i = 0; i < 300; i++) fprintf(f, "v %g %g %g\nvn 0 0 1\n", 0.001 * i, 0.002 * i, 0.5);
        for (int i = 1; i + 2 <= 300; i += 3) fprintf(f, "f %d//%d %d//%d %d//%d\n", i, i, i + 1, i + 1, i + 2, i + 2);
        fclose(f);
        f = fopen(ply_path, "wb");
        REQUIRE(f);
        fprintf(f, "ply\nformat binary_little_endian 1.0\nelement vertex 500\nproperty float x\nproperty float y\nproperty float z\n"
                   "property float nx\nproperty float ny\nproperty float nz\nend_header\n");
        std::vector<float> v(500 * 6, 0.25f);
        fwrite(v.data(), 4, v.size(), f);
        fclose(f);
    }

    long met = 0;
    // 1. operator new fails at every allocation in turn
    {
        uint32_t depth = 0; int cons = 0;
        met += sweep("sdfhip_octdata_validate", [&] { return sdfhip_octdata_validate(ref.structs, ref.length, &depth, &cons); }, [] {});
        REQUIRE(depth == 4 && cons == 1);
    }
    {
        sdfhip_points p{};
        met += sweep("sdfhip_load_obj", [&] { return sdfhip_load_obj(obj_path, &p); }, [&] { REQUIRE(p.count == 300); sdfhip_points_free(&p); });
        met += sweep("sdfhip_load_ply", [&] { return sdfhip_load_ply(ply_path, &p); }, [&] { REQUIRE(p.count == 500); sdfhip_points_free(&p); });
    }
    {
        sdfhip_octdata d{};
        auto same = [&] {
            REQUIRE(d.length == ref.length && memcmp(d.structs, ref.structs, (size_t)ref.length * 8) == 0 &&
                    memcmp(d.values, ref.values, (size_t)ref.length * 8) == 0);
            sdfhip_octdata_free(&d);
        };
        met += sweep("sdfhip_generate (1 thread)", [&] { return sdfhip_generate(SDFHIP_SHAPE_GYROID, gyroid, 6, 4, 1, &d); }, same);
        met += sweep("sdfhip_generate (4 threads)", [&] { return sdfhip_generate(SDFHIP_SHAPE_GYROID, gyroid, 6, 4, 4, &d); }, same);
        met += sweep("sdfhip_asdf_load", [&] { return sdfhip_asdf_load(asdf_path, &d); }, same);
        // 2. the system refuses every thread: the threaded builder builds the same tree on the calling thread
        g_entry = "sdfhip_generate (no threads to be had)";
        g_no_threads.store(true);
        REQUIRE(sdfhip_generate(SDFHIP_SHAPE_GYROID, gyroid, 6, 4, 4, &d) == SDFHIP_OK);
        g_no_threads.store(false);
        same();
        // ... and both at once
        g_no_threads.store(true);
        met += sweep("sdfhip_generate (no threads)", [&] { return sdfhip_generate(SDFHIP_SHAPE_GYROID, gyroid, 6, 4, 4, &d); }, same);
        g_no_threads.store(false);
    }
    REQUIRE(met > 20);                                // (the injector did meet the library's allocations)
    sdfhip_octdata_free(&ref);
    remove(obj_path); remove(ply_path); remove(asdf_path);
    printf("fault injection run ok: %ld injected failures, each one a status code\n", met);
    return 0;
}
