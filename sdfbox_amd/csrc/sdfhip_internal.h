// Internal helpers shared by the host-side sources of libsdfhip.so.
#pragma once
#include "../../include/sdfhip.h"
#include <cstdarg>
#include <cstdio>

namespace sdfhip {

// Thread-local message behind sdfhip_last_error().  Returns `code` so call
// sites can `return fail(SDFHIP_ERR_IO, "...")`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

// sdfhip_scene_upload, or the same from arrays that are already in `device`'s memory (sdfhip_device.hip)
int scene_from_arrays(int device, const int32_t *structs, const uint8_t *values, uint32_t n, bool resident, sdfhip_scene **out);
// true when find() on this scene is a grid lookup (a dense grid as deep as the tree, or a split one): the default kernel
// k_march renders it and can write sparse wire shares (sdfhip_render_sparse_device)
bool scene_has_full_depth_grid(const sdfhip_scene *scene);

// Deterministic double-precision sin/cos (no libm): the same bytes come out of
// the scene generator on every x86-64 host, whatever glibc's ifunc picks.
double det_sin(double x);
double det_cos(double x);

}  // namespace sdfhip
