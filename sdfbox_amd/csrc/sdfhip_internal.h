// Internal helpers shared by the host-side sources of libsdfhip.so.
#pragma once
#include "../../include/sdfhip.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

namespace sdfhip {

// Thread-local message behind sdfhip_last_error().  Returns `code` so call
// sites can `return fail(SDFHIP_ERR_IO, "...")`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

// A measurement or test knob of the environment: read by the laboratory library only.  The product takes its choices through
// its ABI (sdfhip_upload_options, sdfhip_multi_configure) and reads five variables in all, each named in include/sdfhip.h:
// SDFHIP_MULTI_TRANSPORT, SDFHIP_RCCL_LIB, SDFHIP_MULTI_RCCL_SELF (the RCCL transport's self-test on one device), SDFHIP_GEN_POOL,
// SDFHIP_KEEP_ENV (errors.cpp: do not export GPU_MAX_HW_QUEUES at load).
#ifdef SDFHIP_EXPERIMENTS
inline const char *lab_env(const char *name) { return getenv(name); }
#else
inline const char *lab_env(const char *) { return nullptr; }
#endif

// sdfhip_scene_upload[_ex], or the same from arrays that are already in `device`'s memory (sdfhip_sdfgen_scene); opt may be null
// trusted_depth >= 0: the arrays are the GPU builder's own output -- consistent by construction, that deep -- and are not validated again
int scene_from_arrays(int device, const int32_t *structs, const uint8_t *values, uint32_t n, bool resident, const sdfhip_upload_options *opt,
                      sdfhip_scene **out, int trusted_depth = -1);
// true when find() on this scene is a grid lookup (a dense grid as deep as the tree, or a split one): the default kernel
// k_march renders it and can write sparse wire shares (sdfhip_render_sparse_device)
bool scene_has_full_depth_grid(const sdfhip_scene *scene);

// Deterministic double-precision sin/cos (no libm): the same bytes come out of
// the scene generator on every x86-64 host, whatever glibc's ifunc picks.
double det_sin(double x);
double det_cos(double x);

}  // namespace sdfhip

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
namespace sdfhip {
// hipMalloc -- and, when the device is out of memory, once more after the point-cloud builder's chunk pool (sdfgen_device.hip: up to
// 24 GB per device kept between builds) has given back what it holds on the current device.  Every allocation of the scene, the
// render scratch and the multi-device handle goes through it: a pool that sits on the memory must not make an upload fall back to a
// smaller grid, or a render fail.
hipError_t device_alloc_bytes(void **p, size_t bytes);
template <class T> inline hipError_t device_alloc(T **p, size_t bytes) { return device_alloc_bytes(reinterpret_cast<void **>(p), bytes); }
}  // namespace sdfhip
#endif
