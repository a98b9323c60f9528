/* A plain-C host that keeps four frames in flight, as a throughput renderer would -- and as bench.py does -- through the C ABI of
 * include/sdfhip.h and four HIP streams of its own: no Python, no PyTorch, nothing that sets the environment for it.  What it
 * measures is what the HIP runtime's default of 4 hardware queues costs such a host against GPU_MAX_HW_QUEUES=8 (INTEGRATION.md
 * section 3, "Frames in flight and hardware queues"): the runtime reads that variable when it starts, so it is the HOST's to
 * export -- scripts/hw_queues_c_host.sh runs this program with the variable unset and set (-> profiles/r06_hw_queues_c_host.txt).
 *   c_frames_in_flight [frames = 2000] [streams = 4] [depth = 9]      cfg-2's frame: 1920x1080, the gyroid stand-in
 * exit 0 ok, 3 = no usable GPU. */
#define __HIP_PLATFORM_AMD__ 1
#define _GNU_SOURCE 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "sdfhip.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 2000, ns = argc > 2 ? atoi(argv[2]) : 4, depth = argc > 3 ? atoi(argv[3]) : 9;
    const unsigned W = 1920, H = 1080;
    if (frames < 1 || ns < 1 || ns > 8) { fprintf(stderr, "usage: %s [frames] [streams 1..8] [depth]\n", argv[0]); return 2; }
    /* a 4th argument "setenv": export the variable from main(), before the first call that touches the GPU (does the runtime still see it?) */
    if (argc > 4 && argv[4][0] == 's') setenv("GPU_MAX_HW_QUEUES", "8", 0);
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    const float gyroid[6] = { 0.5f, 0.5f, 0.5f, 0.42f, (float)(12.0 * M_PI), 0.004f };
    sdfhip_octdata od;
    if (sdfhip_generate(SDFHIP_SHAPE_GYROID, gyroid, 6, depth, 32, &od) != SDFHIP_OK) { fprintf(stderr, "generate: %s\n", sdfhip_last_error()); return 4; }
    sdfhip_scene *scene = NULL;
    int rc = sdfhip_scene_upload(0, od.structs, od.values, od.length, &scene);
    const unsigned n_nodes = od.length;
    sdfhip_octdata_free(&od);
    if (rc == SDFHIP_ERR_DEVICE) { fprintf(stderr, "upload: %s\n", sdfhip_last_error()); return 3; }
    if (rc != SDFHIP_OK) { fprintf(stderr, "upload: %s\n", sdfhip_last_error()); return 5; }
    sdfhip_info info;
    sdfhip_info_default(&info, (float)W, (float)H);
    sdfhip_info_set_heading(&info, -0.2f, 0.35f);
    sdfhip_info_set_position(&info, 0.5f, 0.5f, -0.35f);
    hipStream_t st[8];
    float *buf[8];
    for (int i = 0; i < ns; i++) {
        if (hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess || hipMalloc((void **)&buf[i], (size_t)W * H * 16) != hipSuccess) {
            fprintf(stderr, "stream / frame buffer %d failed\n", i); return 6;
        }
    }
    double best = 1e9;
    for (int pass = 0; pass < 4; pass++) {                      /* the first pass warms up (clocks, the streams' scratch) */
        if (hipDeviceSynchronize() != hipSuccess) return 7;
        const double t0 = now();
        for (int k = 0; k < frames; k++) {
            rc = sdfhip_render_device(scene, &info, W, H, H, 0, 1, H, SDFHIP_KERNEL_AUTO, buf[k % ns], st[k % ns], NULL);
            if (rc != SDFHIP_OK) { fprintf(stderr, "render: %s\n", sdfhip_last_error()); return 8; }
        }
        if (hipDeviceSynchronize() != hipSuccess) return 7;
        const double ms = (now() - t0) / frames * 1e3;
        if (pass > 0 && ms < best) best = ms;
        printf("  pass %d: %.4f ms per frame\n", pass, ms);
    }
    printf("GPU_MAX_HW_QUEUES=%s: %u nodes, %ux%u, %d frames on %d streams: best %.4f ms per frame = %.0f Mray/s\n", q ? q : "(unset: the runtime's 4)",
           n_nodes, W, H, frames, ns, best, (double)W * H / (best * 1e-3) / 1e6);
    for (int i = 0; i < ns; i++) { (void)hipFree(buf[i]); (void)hipStreamDestroy(st[i]); }
    sdfhip_scene_free(scene);
    return 0;
}
