/* A plain-C caller of the ABI in include/sdfhip.h: no Python, no PyTorch.
 * Loads an .asdf, builds the default camera block, renders W x H and writes the
 * raw RGBA32F frame.  Built and run by tests/test_c_harness.py.
 *   c_harness <scene.asdf> <W> <H> <out.raw>
 * exit 0 ok, 3 = no usable GPU (SDFHIP_ERR_DEVICE), anything else = failure. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sdfhip.h"

int main(int argc, char **argv)
{
    if (argc != 5) { fprintf(stderr, "usage: %s scene.asdf W H out.raw\n", argv[0]); return 2; }
    unsigned W = (unsigned)atoi(argv[2]), H = (unsigned)atoi(argv[3]);
    sdfhip_octdata od;
    if (sdfhip_asdf_load(argv[1], &od) != SDFHIP_OK) { fprintf(stderr, "load: %s\n", sdfhip_last_error()); return 4; }
    uint32_t depth = 0; int consistent = 0;
    if (sdfhip_octdata_validate(od.structs, od.length, &depth, &consistent) != SDFHIP_OK) {
        fprintf(stderr, "validate: %s\n", sdfhip_last_error()); return 5;
    }
    printf("nodes %u depth %u consistent %d\n", od.length, depth, consistent);
    sdfhip_info info;
    sdfhip_info_default(&info, (float)W, (float)H);
    sdfhip_scene *scene = NULL;
    int rc = sdfhip_scene_upload(0, od.structs, od.values, od.length, &scene);
    sdfhip_octdata_free(&od);
    if (rc == SDFHIP_ERR_DEVICE) { fprintf(stderr, "upload: %s\n", sdfhip_last_error()); return 3; }
    if (rc != SDFHIP_OK) { fprintf(stderr, "upload: %s\n", sdfhip_last_error()); return 6; }
    float *frame = (float *)malloc((size_t)W * H * 4 * sizeof(float));
    sdfhip_stats st;
    rc = sdfhip_render(scene, &info, W, H, SDFHIP_KERNEL_AUTO | SDFHIP_FLAG_COUNT, frame, &st);
    if (rc != SDFHIP_OK) { fprintf(stderr, "render: %s\n", sdfhip_last_error()); return 7; }
    printf("kernel %.3f ms, %llu node reads, %llu samples, %llu steps\n", st.kernel_ms,
           (unsigned long long)st.n_nodes, (unsigned long long)st.n_samples, (unsigned long long)st.n_steps);
    /* the viewer's frame array in page-locked memory: the same call, the march stores its pixels into the array itself */
    void *locked = NULL;
    const size_t bytes = (size_t)W * H * 4 * sizeof(float);
    if (sdfhip_host_alloc(bytes, &locked) != SDFHIP_OK) { fprintf(stderr, "host_alloc: %s\n", sdfhip_last_error()); return 9; }
    memset(locked, 0xA5, bytes);
    rc = sdfhip_render(scene, &info, W, H, SDFHIP_KERNEL_AUTO, (float *)locked, NULL);
    if (rc != SDFHIP_OK) { fprintf(stderr, "render (page-locked): %s\n", sdfhip_last_error()); return 10; }
    if (memcmp(locked, frame, bytes) != 0) { fprintf(stderr, "the page-locked frame differs\n"); return 11; }
    if (sdfhip_host_release(locked) != SDFHIP_OK) { fprintf(stderr, "host_release: %s\n", sdfhip_last_error()); return 12; }
    printf("page-locked frame identical\n");
    FILE *f = fopen(argv[4], "wb");
    if (!f || fwrite(frame, sizeof(float), (size_t)W * H * 4, f) != (size_t)W * H * 4) return 8;
    fclose(f);
    free(frame);
    sdfhip_scene_free(scene);
    return 0;
}
