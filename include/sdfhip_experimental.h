/*
 * sdfhip_experimental.h -- the LABORATORY of libsdfhip: entry points, flags and test hooks of kernel forms and gather formats
 * that were built, proved bit-identical to the oracle, and MEASURED SLOWER than (or equal to) the product's path
 * (DESIGN.md sections 4.2-4.7 and 5 hold the numbers).  They stay as A/B knobs for measurements and as regression tests of the
 * alternatives, in a library of their own:
 *
 *     libsdfhip.so       the product: exports exactly what include/sdfhip.h declares
 *     libsdfhip_lab.so   the same sources built with -DSDFHIP_EXPERIMENTS: the product's symbols + the ones below
 *
 * (make -C sdfbox_amd/csrc builds both.)  A host application never needs this header.
 */
#ifndef SDFHIP_EXPERIMENTAL_H
#define SDFHIP_EXPERIMENTAL_H

#include "sdfhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- A/B render flags (libsdfhip.so refuses them) --------------------------------------------------------------------- */
enum {
    SDFHIP_FLAG_WIRE = 0x10000,   /* device-resident entry points only: 5-byte wire pixels, round 1's gather format (see
                                     sdfhip_deinterleave_device with pixel_bytes = 5 below), lossless.  Every pixel Compute.hlsl
                                     writes is (a, a, a, n) with n <= 140 steps, or the sky constant (0.005, 0.01, 0.2, n) with
                                     n <= 100, so a frame of nrows_out x width pixels travels as nrows_out*width floats (the
                                     bits of a) followed by nrows_out*width bytes (n, or 255 - n for a sky pixel);
                                     nrows_out*width must be a multiple of 4 */
    /* bits 8..11: blockIdx -> tile order of the one-kernel form (1 row-major, 2 one slab per XCD); bits 12..15: its workgroup
     * size (1 = 64, 2 = 128, 3 = 256 threads).  Results never depend on them. */
    SDFHIP_TUNE_ORDER_SHIFT = 8,
    SDFHIP_TUNE_BLOCK_SHIFT = 12,
    /* round 1's one-kernel form (k_plain / k_path: a lane state machine that shades in place) on a scene where the product
     * runs k_march (primary march, shading, shadow march as three wave-converged loops) or the path-traced pipeline */
    SDFHIP_TUNE_ONE_KERNEL = 0x20000,
    /* the one-kernel form with the top grid (level <= 3: SDFHIP_TOP_GRID_LEVEL=3 at upload) staged in LDS per workgroup:
     * north_star's "LDS caching of the hot inner nodes"  */
    SDFHIP_TUNE_LDS_TOP = 0x40000,
    /* SDFHIP_FLAG_COMPACT's pair of kernels with EVERY shadow ray queued (round 2's form; SDFHIP_SHADOW_MIN_LANES=T in the
     * environment sets another threshold): k_march appends them to a queue (wavefront ballot + prefix compaction) that
     * k_shadow marches 64 to a wave.  The product queues the rays of waves that hold fewer than 32 */
    SDFHIP_TUNE_SHADOW_QUEUE = 0x80000,
    /* only meaningful on a scene uploaded with SDFHIP_SAMPLE_RECORDS=1 in the environment, which builds the grid's second
     * form (dense 4-byte words + 64-byte sample records of the non-flat leaves) and makes the default kernel read it: this
     * flag switches such a scene back to the 16-byte cells every other scene reads */
    SDFHIP_TUNE_BYTE_CELLS = 0x200000,
    /* with SDFHIP_FLAG_COMPACT on a scene behind a full-depth grid: the persistent-wave lane-refill kernel (k_compact), which carried
     * that flag until round 4 and still does on trees without such a grid: 2.9 x slower than the default at 4K */
    SDFHIP_TUNE_PERSISTENT_WAVES = 0x400000
};

/* ---- round 1 / round 2 gather formats (superseded by the sparse shares the march kernel writes, sdfhip_render_sparse_device) */
/* sdfhip_deinterleave_device / _bands_device of this library also accept pixel_bytes = 5: d_gathered then holds the wire
 * buffers SDFHIP_FLAG_WIRE renders make, and d_frame receives the RGBA32F frame, bit for bit what a render without the flag
 * writes. */

/* Sparse wire format: what the ranks put on xGMI when most of a frame is sky.  A frame-share in the
 * wire format of SDFHIP_FLAG_WIRE (rows x width pixels, rows a multiple of 8) is compacted on its own
 * GPU -- the code bytes stay; of the float plane only the values with any bit set, packed in tile
 * order behind a 64-bit mask and a slot index per 8x8 tile -- gathered, and expanded by rank 0 while it
 * restores row order.  `capacity` = float slots per frame-share; a share with more lit pixels sets the
 * overflow word (then the frame is not complete: choose the capacity from a measured maximum, or
 * rows * width to be safe).  Lossless within the capacity; 1.2 bytes per pixel + 4 per lit pixel.
 *   sdfhip_wire_sparse_bytes            bytes of one sparse frame-share
 *   sdfhip_wire_compact_device          d_wire [frames] dense wire shares -> d_sparse [frames] sparse shares
 *   sdfhip_deinterleave_sparse_device   like sdfhip_deinterleave[_bands]_device (owner may be NULL: round
 *                                       robin) for [world][frames] sparse shares; *d_overflow (device word,
 *                                       may be NULL) is OR-ed with 1 when a share overflowed */
SDFHIP_API uint64_t sdfhip_wire_sparse_bytes(uint32_t width, uint32_t rows, uint32_t capacity);
/* Recovery when a sparse share overflowed its capacity: byte offset, within a sparse share, of its 16-byte
 * header {uint32 lit pixels, uint32 overflowed, 0, 0} -- the sender reads word 1 of its own shares and rank 0
 * that of the gathered ones, and the rank concerned sends the share again in the dense wire format
 * (point to point: no other rank takes part), which rank 0 writes over that rank's rows with
 *   sdfhip_deinterleave_share_device   like sdfhip_deinterleave[_bands]_device, but d_share holds the
 *                                      [frames] buffers of ONE rank (`rank`) and only its rows are written
 * (owner may be NULL: round robin). */
SDFHIP_API uint64_t sdfhip_wire_sparse_head_offset(uint32_t width, uint32_t rows, uint32_t capacity);
SDFHIP_API int sdfhip_deinterleave_share_device(int device, const void *d_share, void *d_frame,
                                                uint32_t width, uint32_t height, uint32_t band_rows,
                                                uint32_t world, uint32_t rows_per_rank, const uint8_t *owner,
                                                uint32_t rank, uint32_t pixel_bytes, uint32_t frames, void *stream);
SDFHIP_API int sdfhip_wire_compact_device(int device, const void *d_wire, void *d_sparse, uint32_t width,
                                          uint32_t rows, uint32_t frames, uint32_t capacity, void *stream);
SDFHIP_API int sdfhip_deinterleave_sparse_device(int device, const void *d_gathered, void *d_frame,
                                                 uint32_t width, uint32_t height, uint32_t band_rows,
                                                 uint32_t world, uint32_t rows_per_rank, const uint8_t *owner,
                                                 uint32_t capacity, uint32_t frames, uint32_t *d_overflow,
                                                 void *stream);

/* ---- test hooks ----------------------------------------------------------------------------------------------------------- */
/* Test hook: how many packed floats the next share of every device carries (normally 1.25 x what its last share used; 0 = all
 * of them): a small value forces the float tail of the next shares to be sent again (sdfhip_multi_stats.resends). */
SDFHIP_API int sdfhip_multi_debug_floats_sent(sdfhip_multi *m, uint32_t floats);

/* Experiment hook (scripts/ab_tile_order.py): the default kernel of the following single-frame renders on this scene takes its
 * tiles from d_perm (device array of B = 8 * ceil(tiles_y / 8) * tiles_x entries, one per workgroup of a frame, 8x8 tiles), stored
 * LABEL-MAJOR: entry [label * (B / 8) + slot] is the tile of the slot-th workgroup that carries XCD label `label` (0..7: the label
 * is blockIdx.x of the launch, whose grid is (8, frames, B / 8)), PACKED as tile_row << 16 | tile_col; an entry whose row or
 * column lies outside the frame idles.  The kernel writes the march iterations of every tile's wave to d_cost[tile_row * tiles_x
 * + tile_col] (device array): the primary loop's in the low byte, the shadow loop's in the high byte.  NULL switches either off.
 * Refused (SDFHIP_ERR_ARG at the render) for frames with more than 65 535 tile rows or columns or more than 8 * 65 535 workgroups
 * (B / 8 is a grid dimension). */
SDFHIP_API int sdfhip_debug_tile_order(sdfhip_scene *scene, const uint32_t *d_perm, uint16_t *d_cost);

/* The compulsory bytes of this design: between _begin and _end every SDFHIP_FLAG_COUNT render on this scene
 * marks, per XCD, the 128-byte lines of the lookup grid its find() touches.  A "phase" is what is counted on its own: a whole
 * frame of the default / compact kernels; for sdfhip_render_path_device the camera segments and then each bounce level (each is a
 * kernel launch of its own, reading the bounce levels' second grid when the scene has one).  _end synchronises the device and
 * returns, per phase, out[8 * phase + ...] = {lines of the coarse array, lines of the fine array (distinct, chip-wide: one ideal
 * cache), the same two summed over the 8 XCDs (eight ideal L2s that share nothing), which grid (0: the scene's own, 2: the
 * bounce levels'), 0, 0, 0}; array_bytes4 (may be NULL) = the bytes of {coarse, fine, coarse2, fine2}.  Needs a scene whose grid
 * is as deep as its tree.  The hook costs the counting kernels an atomic per new line; the product has no such code. */
SDFHIP_API int sdfhip_debug_touch_begin(sdfhip_scene *scene);
SDFHIP_API int sdfhip_debug_touch_end(sdfhip_scene *scene, uint64_t *out, uint32_t max_phases, uint32_t *n_phases, uint64_t *array_bytes4);

/* Test hook (tests/test_gpu_fault_injection.py): the countdown-th allocation that THIS library's own host code makes from now on
 * (operator new: containers, handles, threads' state) throws std::bad_alloc; countdown < 0 switches the injector off.
 * *thrown_so_far (may be NULL) receives how many allocations have been failed since the library was loaded.  Whatever entry
 * point the failure lands in must return a status code (SDFHIP_ERR_NOMEM) with sdfhip_last_error() set -- the firewall of
 * csrc/abi_guard.h -- and leave the library usable.  The product has no such code. */
SDFHIP_API int sdfhip_debug_fail_host_allocations(int64_t countdown, uint64_t *thrown_so_far);

/* Diagnostics: after a SDFHIP_FLAG_COUNT render of the default kernel on `stream` (the stream argument of the
 * sdfhip_render_device call that made it; synchronises with it), how many lane-steps sampled which kind of cell: out6 = {flat leaf at or above the grid's coarse level, flat leaf below it, non-flat at or above the coarse level,
 * non-flat as deep as the grid, non-flat in between, non-flat with the position outside the cube (or NaN)}.  What the
 * layout of the grid's cells is tuned by. */
SDFHIP_API int sdfhip_debug_step_classes(sdfhip_scene *scene, void *stream, uint64_t *out6);

/* Test hook: the kernel's R8_UNorm decode of bytes 0..255 (256 floats to the
 * host), checked exhaustively against byte/255.0f. */
SDFHIP_API int sdfhip_debug_unorm_table(int device, float *out256);

#ifdef __cplusplus
}
#endif
#endif /* SDFHIP_EXPERIMENTAL_H */
