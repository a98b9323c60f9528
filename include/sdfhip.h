/*
 * sdfhip.h -- C ABI of libsdfhip.so: MI355X (gfx950) sphere tracing of
 * adaptively sampled distance fields behind SdfBox's frame boundary.
 *
 * Plain C, cdecl, PODs and pointers only; every function returns an int
 * status (SDFHIP_OK == 0) and never throws or aborts across the boundary;
 * sdfhip_last_error() returns the calling thread's last message.
 *
 * Each entry point names the reference interface it replaces (paths are
 * relative to the tau-dev/SdfBox checkout).  INTEGRATION.md shows the C#
 * [DllImport] stub a maintainer would add.
 *
 * Loading the library has ONE side effect on its host: unless GPU_MAX_HW_QUEUES is already in the environment (any value: the
 * host's word stands) or SDFHIP_KEEP_ENV is set, it exports GPU_MAX_HW_QUEUES=8 -- the HIP runtime reads that variable when the
 * process makes its first HIP call, and a host that keeps four frames in flight on four streams renders 20 % slower on the
 * runtime's default of four hardware queues (0.1014 against 0.0841 ms per 1080p frame: INTEGRATION.md section 3).  A process
 * that has initialised HIP before it loads the library keeps what it had.  (The export is a setenv() in the library's
 * constructor: like every setenv it is not ordered against a getenv() that another thread of the host makes at that very moment.
 * A host that loads the library while other threads of it run either exports the variable itself at start-up -- the export then
 * never happens -- or sets SDFHIP_KEEP_ENV.)
 */
#ifndef SDFHIP_H
#define SDFHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDFHIP_API __attribute__((visibility("default")))

/* ---- status codes ------------------------------------------------------ */
enum {
    SDFHIP_OK = 0,
    SDFHIP_ERR_ARG = 1,       /* null pointer, zero size, bad enum            */
    SDFHIP_ERR_IO = 2,        /* file missing / short / not an .asdf          */
    SDFHIP_ERR_BAD_TREE = 3,  /* parent/children index out of range           */
    SDFHIP_ERR_DEVICE = 4,    /* HIP call failed or no gfx950 device          */
    SDFHIP_ERR_NOMEM = 5
};

/* ---- data contract ----------------------------------------------------- */

/* The `Info` cbuffer, passed verbatim (112 bytes).
 * Replaces: struct Info, SdfBox/Logic.cs:407-420 == Compute.hlsl:70-81;
 * heading rows are Float3x3, SdfBox/Logic.cs:427-463.
 * buffer_size is ignored by the renderer (the scene handle knows its length);
 * hidef is unused by the shader and by us. */
typedef struct sdfhip_info {
    float heading[3][4];   /*   0 */
    float position[3];     /*  48 */
    float margin;          /*  60 */
    float screen_size[2];  /*  64 */
    uint32_t buffer_size;  /*  72 */
    float limit;           /*  76 */
    float light[3];        /*  80 */
    float strength;        /*  92 */
    float fov;             /*  96 */
    int32_t hidef;         /* 100 */
    uint32_t pad_[2];      /* 104 -> 112 */
} sdfhip_info;

/* Flattened octree as the reference's native loader hands it over.
 * Replaces: struct OctData {Length, Structs, Values}, SdfGen/dllmain.cpp:36-41
 * == NativeOctData, SdfBox/Program.cs:579-583.  structs = N x {int32 parent,
 * int32 children} (OctS, dllmain.cpp:18-29), values = N x 8 bytes, corner
 * k = x + 2y + 4z, *before* the texture swizzle of Program.cs:514-538. */
typedef struct sdfhip_octdata {
    uint32_t length;
    int32_t *structs;
    uint8_t *values;
} sdfhip_octdata;

/* Opaque scene handle: the octree resident in one GPU's HBM. */
typedef struct sdfhip_scene sdfhip_scene;

/* Render flags.  (A scene handle keeps scratch memory per stream that renders on it, 16 at a time; a 17th stream takes over the
 * least recently used scratch whose work has drained.)  Any other bit is refused with SDFHIP_ERR_ARG: the A/B knobs of the
 * measured-and-dropped kernel forms exist in the experiments build only (include/sdfhip_experimental.h, libsdfhip_lab.so). */
enum {
    SDFHIP_KERNEL_AUTO = 0,       /* the default: a grid lookup per find() wherever the tree allows, else the shader's own traversal */
    SDFHIP_KERNEL_GENERIC = 1,    /* one thread per pixel, follows parent / children links through memory as Compute.hlsl does */
    SDFHIP_KERNEL_STACK = 2,      /* integer cell coordinates and the lookup grids (needs a consistent tree of depth <= 12)   */
    SDFHIP_KERNEL_MASK = 0xF,
    SDFHIP_FLAG_COMPACT = 0x10,   /* BASELINE cfg-3's wavefront ray compaction: after the shading step a wave that holds fewer than 32
                                     shadow rays hands them to a queue (slots by ballot + prefix count) that a second kernel marches
                                     64 to a wave; fuller waves march theirs in place.  On a tree without a full-depth grid:
                                     persistent waves with ballot/prefix refill of finished lanes.  Bit-identical; 2-3 % slower than
                                     the default on the frames measured -- coherent primary rays leave little to compact */
    SDFHIP_FLAG_COUNT = 0x20,     /* also count algorithmic node/sample reads (slower)     */
    SDFHIP_FLAG_DISPLAY = 0x40,   /* fused display pass: output is RGBA8, gamma 1/2.2 (DisplayFrag.hlsl:24) */
    SDFHIP_FLAG_DISPLAY_DEBUG = 0x80, /* fused display pass, debug heat map w/140 (DisplayFrag.hlsl:21-22) */
    SDFHIP_FLAG_TILE_ORDER = 0x100000 /* for a viewer that renders one frame at a time: launch this frame's 8x8 tiles in
                                     descending order of the march iterations they (or a tile within two of them) took in the
                                     last frame rendered with the same geometry on the same stream; the order is made on the
                                     device behind every such frame whose camera block differs from the one the order in use
                                     came from (two small kernels, 13 us; a camera at rest pays once).  The first frame of a
                                     geometry, frames of a batch, frames of more than 65 536 tiles and frames rendered with
                                     SDFHIP_FLAG_COMPACT or in path-traced mode take the default order.
                                     A frame alone ends when its longest wave does, so the tiles that were expensive a moment
                                     ago go first: 0.174 -> 0.133 ms per 1080p frame with the camera at rest, 0.188 -> 0.148
                                     with one degree between frames.  Not for frames in flight on several streams (the ordering
                                     kernels sit between a stream's frames).  Never changes a pixel. */
};

/* Per-call statistics (all optional: pass NULL). */
typedef struct sdfhip_stats {
    float kernel_ms;        /* HIP-event time of the ray-march kernel(s)      */
    float total_ms;         /* kernel + device->host copy, host clock         */
    uint64_t n_nodes;       /* with SDFHIP_FLAG_COUNT: node records find()    */
                            /* reads in the reference algorithm   */
    uint64_t n_samples;     /* interpol_world calls                           */
    uint64_t n_steps;       /* sum of the alpha channel (march steps)         */
    uint32_t kernel_used;   /* SDFHIP_KERNEL_GENERIC or _STACK (| COMPACT)    */
    uint32_t pad_;
    uint64_t n_shadow_rays; /* with SDFHIP_FLAG_COUNT: pixels (path vertices) */
                            /* that cast a shadow ray, Compute.hlsl:213       */
    uint64_t n_loads;       /* with SDFHIP_FLAG_COUNT: 16-byte node records / */
                            /* grid cells the kernels themselves loaded (one  */
                            /* per lane and load): their own algorithmic reads */
    uint64_t n_hits;        /* with SDFHIP_FLAG_COUNT: entries that travelled     */
                            /* through a queue between two kernels -- the path-   */
                            /* traced pipeline's hits, summed over its levels (48 */
                            /* bytes each, written once and read once); shadow    */
                            /* rays queued by SDFHIP_FLAG_COMPACT; else 0         */
} sdfhip_stats;

/* ---- errors ------------------------------------------------------------ */
/* Replaces: CheckError -> throw across the FFI, SdfGen/pch.h:20-26. */
SDFHIP_API const char *sdfhip_last_error(void);

/* ---- scene data on the host (.asdf) ------------------------------------ */

/* Replaces: LoadAsdf, SdfGen/dllmain.cpp:250-276 (P/Invoke Program.cs:658).
 * Allocates out->structs / out->values; release with sdfhip_octdata_free. */
SDFHIP_API int sdfhip_asdf_load(const char *path, sdfhip_octdata *out);

/* Replaces: Save, SdfGen/dllmain.cpp:278-292 (P/Invoke Program.cs:661). */
SDFHIP_API int sdfhip_asdf_save(const sdfhip_octdata *data, const char *path);

/* Replaces: Free, SdfGen/dllmain.cpp:346-351 (P/Invoke Program.cs:666). */
SDFHIP_API void sdfhip_octdata_free(sdfhip_octdata *data);

/* Analytic scene builder: the split rule and quantiser of SdfGen's
 * construct / FromFloat / WriteBytes (SdfGen/dllmain.cpp:163-207) applied to
 * a closed-form distance function instead of a point cloud, same node order.
 * Stands in for SdfGen (dllmain.cpp:295-319), which needs mesh files that do
 * not ship.  shape: SDFHIP_SHAPE_*; params: see each shape. */
enum {
    SDFHIP_SHAPE_SPHERE = 0,   /* params: cx, cy, cz, r                        */
    SDFHIP_SHAPE_TORUS = 1,    /* params: cx, cy, cz, R, r  (axis = y)         */
    SDFHIP_SHAPE_GYROID = 2    /* params: cx, cy, cz, clip_r, freq, thickness  */
};
SDFHIP_API int sdfhip_generate(int shape, const float *params, int nparams,
                               int max_depth, int nthreads, sdfhip_octdata *out);

/* ---- point cloud -> ASDF --------------------------------------------- */

/* A point cloud with normals: count x {position xyz, normal xyz} floats.
 * Replaces: gsl::span<Vertex>* / struct Vertex, SdfGen/math.h:47-51. */
typedef struct sdfhip_points {
    uint32_t count;
    float *data;
} sdfhip_points;

/* Replaces: LoadPly, SdfGen/dllmain.cpp:244-248 -> ply_reader.cpp:35-71 (binary
 * little-endian, vertex element first, 6 floats per vertex). */
SDFHIP_API int sdfhip_load_ply(const char *path, sdfhip_points *out);
/* Replaces: LoadObj, SdfGen/dllmain.cpp:237-242 -> obj_reader.cpp:45-96. */
SDFHIP_API int sdfhip_load_obj(const char *path, sdfhip_points *out);
SDFHIP_API void sdfhip_points_free(sdfhip_points *points);

typedef struct sdfhip_sdfgen_stats {
    uint32_t nodes, levels;
    uint64_t candidate_entries;   /* sum of all candidate-list lengths: the work measure */
    float global_scale;           /* GlobalScale / GlobalOffset of dllmain.cpp:67-80 */
    float global_offset[3];
    float total_ms;
} sdfhip_sdfgen_stats;

/* Replaces: SdfGen(vertices, depth), SdfGen/dllmain.cpp:295-319 (P/Invoke Program.cs:
 * 662-663): builds the flattened octree of a point cloud on GPU `device`, level by
 * level, one wavefront per node -- the same nodes, order and bytes as the reference's
 * recursive construct (:163-207), including its lossy candidate pruning, corner
 * inheritance and first-point tie-breaking.  out: release with sdfhip_octdata_free.
 * SDFHIP_ERR_ARG when no point can be the nearest one (NaN input: the reference throws
 * "Did not find" / "NaN distance"). */
SDFHIP_API int sdfhip_sdfgen(int device, const float *verts6, uint32_t n, int32_t depth,
                             sdfhip_octdata *out, sdfhip_sdfgen_stats *stats);

/* Replaces: the viewer's generate -> upload flow, NativeOctData.Generate (SdfBox/Program.cs:613-650: LoadPly / LoadObj -> SdfGen) followed
 * by OctData.StructBuffer() / ValueTexture() and their binding (:543-572, :147-152): the point cloud goes in, a scene handle on `device`
 * comes out, and the tree never leaves HBM -- sdfhip_sdfgen's copy of the result to the host (12 ms for the 192 MB of a 12 M-node tree)
 * and sdfhip_scene_upload's copy back (8 ms) do not happen.  out (may be NULL): also the host arrays, e.g. for the .asdf cache the
 * reference writes next to the mesh (Program.cs:638-641).  The handle's renders are those of sdfhip_sdfgen + sdfhip_scene_upload. */
SDFHIP_API int sdfhip_sdfgen_scene(int device, const float *verts6, uint32_t n, int32_t depth, sdfhip_scene **scene,
                                   sdfhip_octdata *out, sdfhip_sdfgen_stats *stats);

/* The builder keeps the device memory of its work arrays (up to 24 GB per device: what a depth-10 tree of a million points takes) for
 * the next build of the process -- on this stack the first allocation after gigabytes have been freed takes a third of a second.
 * This gives it back (Replaces: nothing -- the reference builds on the host; cf. Free, SdfGen/dllmain.cpp:349-358).  A build that
 * runs out of device memory trims the pool itself and tries again.  SDFHIP_GEN_POOL=0 in the environment turns the pool off. */
SDFHIP_API int sdfhip_sdfgen_trim(void);

/* Structural check used by upload: 0 = ok; SDFHIP_ERR_BAD_TREE for an index out of
 * range, a cycle in the parent links or a parent chain of more than 64 links (either
 * would keep the shader's ascend loop from terminating).  depth_out = deepest level,
 * consistent_out = 1 when every child's parent field points back at it. */
SDFHIP_API int sdfhip_octdata_validate(const int32_t *structs, uint32_t n,
                                       uint32_t *depth_out, int *consistent_out);

/* ---- camera block ------------------------------------------------------ */

/* Replaces: Logic.State initialiser, SdfBox/Logic.cs:30-38 + Program.cs:54
 * (heading = identity via Logic.Heading = Zero) + limit via Logic.Position. */
SDFHIP_API void sdfhip_info_default(sdfhip_info *info, float width, float height);

/* Replaces: Logic.Heading setter, SdfBox/Logic.cs:46-55:
 * heading = Float3x3(Matrix4x4.CreateFromYawPitchRoll(heading_y, heading_x, 0)). */
SDFHIP_API void sdfhip_info_set_heading(sdfhip_info *info, float heading_x, float heading_y);

/* Replaces: Logic.Position setter, SdfBox/Logic.cs:60-78 (position + limit). */
SDFHIP_API void sdfhip_info_set_position(sdfhip_info *info, float x, float y, float z);

/* Movement keys of Logic.Update (SdfBox/Logic.cs:252-271), as a bit mask. */
enum {
    SDFHIP_KEY_RIGHT = 1, SDFHIP_KEY_LEFT = 2, SDFHIP_KEY_UP = 4, SDFHIP_KEY_DOWN = 8,   /* arrow keys: turn   */
    SDFHIP_KEY_FORWARD = 16,        /* W / numpad 8 */
    SDFHIP_KEY_BACK = 32,           /* S / numpad 2 */
    SDFHIP_KEY_STRAFE_RIGHT = 64,   /* D / numpad 6 */
    SDFHIP_KEY_STRAFE_LEFT = 128,   /* A / numpad 4 */
    SDFHIP_KEY_SHIFT = 256,         /* left shift / numpad 9: position.y -= step */
    SDFHIP_KEY_CONTROL = 512        /* left control / numpad 3: position.y += step */
};

/* Replaces: the camera part of Logic.Update, SdfBox/Logic.cs:239-272, for one time step of
 * `seconds`: the arrow keys turn the heading by tSpeed * seconds, the movement keys move the
 * position by mSpeed^2 * seconds in the yaw plane (yawMat, Logic.cs:79-83) or along y, in the
 * reference's order; heading_xy (in/out: X = pitch, Y = yaw) and `info` (heading, position,
 * limit) are updated.  m_speed: Logic.mSpeed, 0.5 at start (Logic.cs:28). */
SDFHIP_API void sdfhip_camera_update(sdfhip_info *info, float *heading_xy, float m_speed, uint32_t keys, float seconds);

/* Replaces: Logic.MouseMove, SdfBox/Logic.cs:290-293: heading += (-dy, dx) / 512 * 4. */
SDFHIP_API void sdfhip_camera_mouse_move(sdfhip_info *info, float *heading_xy, float dx, float dy);

/* Replaces: the MouseWheel handler, SdfBox/Logic.cs:202-205: the new mSpeed. */
SDFHIP_API float sdfhip_camera_mouse_wheel(float m_speed, float wheel_delta);

/* ---- device ------------------------------------------------------------ */

SDFHIP_API int sdfhip_device_count(int *count);
/* The PCI bus id of a device ("0000:c1:00.0"; out holds at least 16 bytes): what tells two ranks of a multi-GPU run apart. */
SDFHIP_API int sdfhip_device_pci_bus_id(int device, char *out, uint32_t len);
/* What this device's memory delivers to a streaming kernel, in GB/s (1e9 bytes): a float4 copy (reads + writes 2 x bytes),
 * STREAM's triad a = b + s c (3 x bytes) and a read-only sum (1 x bytes) over arrays of `bytes` each (>= 1 MiB; use >= 1 GiB:
 * the Infinity Cache holds 256 MiB), `reps` launches timed with HIP events.  Any result pointer may be null (no triad: one
 * array less; read only: one array).  SURVEY.md 8d's "measured device bandwidth on the box": the denominator bench.py quotes HBM
 * fractions against, beside the 8 TB/s nameplate.  Nothing in the reference corresponds to it (it displays FPS only,
 * SdfBox/Logic.cs:298-301). */
SDFHIP_API int sdfhip_device_bandwidth(int device, uint64_t bytes, uint32_t reps, double *copy_gbs, double *triad_gbs, double *read_gbs);

/* Replaces: OctData.StructBuffer() + OctData.ValueTexture(),
 * SdfBox/Program.cs:543-572, bound at Program.cs:147-152: copies the scene to
 * device `device` once.  Host arrays may be freed after return. */
SDFHIP_API int sdfhip_scene_upload(int device, const int32_t *structs, const uint8_t *values,
                                   uint32_t n, sdfhip_scene **out);
/* The same with the choices the upload otherwise makes by itself (sdfhip_scene_top_grid below describes them).  Every field:
 * -1 = choose.  A host uses this to bound the accelerators' memory or to take a particular grid; the pixels never depend on it. */
typedef struct sdfhip_upload_options {
    uint32_t size;            /* sizeof(sdfhip_upload_options), set by sdfhip_upload_options_default: lets the struct grow -- a smaller
                                 (older) struct is accepted from version 1's 20 bytes on, its missing fields mean "choose"; a larger
                                 (newer) one is accepted when the fields this library does not know are all -1 */
    int32_t top_grid_level;   /* 0 = no grid; 1..10 = a plain grid of that level, as deep as the tree at most -- the tree's depth asks
                                 for the dense full-depth grid, taken if it fits 1/64 of the device's memory (2.1 GB at depth 9) */
    int32_t top_grid_split;   /* 0 = never a split grid; 1..8 = a split grid with that coarse level (needs depth - level <= 6) */
    int32_t scatter_grid;     /* the path-traced mode's second grid: 0 = none, 1..4 = the levels of its blocks (8^n cells each) */
    int32_t scatter_order;    /* ... 0 = its blocks in x-y-z order (default: 2x2x2 sub-cubes, one cache line each) */
} sdfhip_upload_options;
SDFHIP_API void sdfhip_upload_options_default(sdfhip_upload_options *opt);
SDFHIP_API int sdfhip_scene_upload_ex(int device, const int32_t *structs, const uint8_t *values, uint32_t n,
                                      const sdfhip_upload_options *opt, sdfhip_scene **out);
SDFHIP_API int sdfhip_scene_free(sdfhip_scene *scene);
SDFHIP_API int sdfhip_scene_info(const sdfhip_scene *scene, uint32_t *n, uint32_t *depth,
                                 int *stack_kernel_ok, int *device);

/* The top grid the upload built for the cursor-stack kernels: for every cell of octree level
 * `level`, the record of the deepest node of that level or above that contains it, so that a
 * find() that restarts near the root takes one load instead of `level` dependent ones (results
 * and algorithmic counts are unchanged).  level = 0, bytes = 0: none (inconsistent or deeper than
 * 12 levels: generic kernel).  The level is the tree's depth for trees of depth <= 8 (16 bytes per cell,
 * 8^level cells, <= 268 MB) -- every leaf is then in the grid and a find is one load.
 * Deeper trees (up to 12 levels) get a split grid: a coarse dense level (reported as `level`; as deep as 8,
 * no larger than the tree's own records) whose internal cells point at dense blocks of the remaining <= 4
 * levels, built only where the tree is deep (`bytes` counts both; a find is one or two loads), if the blocks
 * fit 1/16 of the memory; else a plain grid of at most level 8, no larger than the tree's own records.
 * sdfhip_scene_upload_ex takes other choices (sdfhip_upload_options: a plain grid of a given level, none at all,
 * a split grid with a given coarse level).  The grid shrinks by itself when memory is short.  (The laboratory
 * library also reads them from the environment: SDFHIP_TOP_GRID_LEVEL, SDFHIP_TOP_GRID_SPLIT.) */
SDFHIP_API int sdfhip_scene_top_grid(const sdfhip_scene *scene, int32_t *level, uint64_t *bytes);

/* Replaces: Program.Draw's UpdateBuffer(info) + DispatchSized(W, H, 1),
 * SdfBox/Program.cs:81,94 (kernel: Compute.hlsl:180-231).  Renders the whole
 * W x H frame and copies it to `rgba_out` (host, W*H*4 floats, row-major,
 * top-left origin, tightly packed: no 12-pixel padding, unlike Program.cs:
 * 311-314).  Synchronous. */
SDFHIP_API int sdfhip_render(sdfhip_scene *scene, const sdfhip_info *info,
                             uint32_t width, uint32_t height, uint32_t flags,
                             float *rgba_out, sdfhip_stats *stats);

/* Device-resident variant for callers that keep the frame in HBM (multi-GPU
 * tile sharding, timing with inputs resident).  Renders the rows
 *     y = (band_first + (k / band_rows)*band_stride)*band_rows + k % band_rows,
 * k = 0 .. nrows_out-1, i.e. every band_stride-th band of band_rows rows
 * starting at band `band_first`, into d_rgba_out (device pointer, nrows_out x
 * width x 4 floats, compact).  band_stride = 1, band_first = 0, nrows_out =
 * height renders the whole frame.  Asynchronous on `stream` (a hipStream_t;
 * NULL = the HIP default stream, as for any HIP call); no host
 * synchronisation unless `stats` is given. */
SDFHIP_API int sdfhip_render_device(sdfhip_scene *scene, const sdfhip_info *info,
                                    uint32_t width, uint32_t height,
                                    uint32_t band_rows, uint32_t band_first,
                                    uint32_t band_stride, uint32_t nrows_out,
                                    uint32_t flags, float *d_rgba_out, void *stream,
                                    sdfhip_stats *stats);

/* Path-traced mode (BASELINE config 5: 16 spp diffuse path trace).  NOT in the
 * reference -- its README lists path tracing under "plans" only -- so there is no
 * interface to replace; the mode is defined by o_pixel_pt in oracle/sdf_oracle.c
 * (DESIGN.md section 8): per pixel `spp` jittered camera rays, each followed by up to
 * `max_bounces` cosine-weighted diffuse bounces, every segment marched, shaded and
 * shadow-tested with the reference's own rules (Compute.hlsl:194-230); RNG = PCG hash
 * of (seed + pixel, sample, bounce, draw).  Output RGBA32F: mean radiance, alpha =
 * march steps of all segments.  Same buffer conventions as sdfhip_render[_device]. */
typedef struct sdfhip_pathtrace {
    uint32_t spp;           /* samples per pixel, 1..4096 (config 5: 16)            */
    uint32_t max_bounces;   /* diffuse bounces after the camera ray (config 5: 3)   */
    uint32_t seed;          /* config 5: 0x5DFB0C5                                  */
    float albedo;           /* diffuse reflectance of the surface (0.8)             */
} sdfhip_pathtrace;
/* The bounce levels of the path-traced pipeline read a second split grid of the scene's cells, with larger blocks (+0.83 GB for
 * the depth-9 bench scene).  sdfhip_scene_prepare_path builds it at load time (allocations, kernels and two
 * stream synchronisations on the scene's own stream); without the call the first path-traced render builds it before its clock
 * starts.  Its blocks hold 16^3 cells by default (8^3 for trees of depth < 6) in the order of 2x2x2 sub-cubes, one cache line each;
 * sdfhip_upload_options.scatter_grid / scatter_order choose otherwise (0 = no second grid).  sdfhip_scene_top_grid counts its
 * bytes once it exists.  sdfhip_render_path returns SDFHIP_ERR_NOMEM, not a wrong image, if a hit ever found no room in its queue
 * (their capacity is the worst case of every sub-queue, so this is a check, not a limit). */
SDFHIP_API int sdfhip_scene_prepare_path(sdfhip_scene *scene);
SDFHIP_API int sdfhip_render_path(sdfhip_scene *scene, const sdfhip_info *info,
                                  const sdfhip_pathtrace *pt, uint32_t width, uint32_t height,
                                  uint32_t flags, float *rgba_out, sdfhip_stats *stats);
SDFHIP_API int sdfhip_render_path_device(sdfhip_scene *scene, const sdfhip_info *info,
                                         const sdfhip_pathtrace *pt, uint32_t width, uint32_t height,
                                         uint32_t band_rows, uint32_t band_first,
                                         uint32_t band_stride, uint32_t nrows_out, uint32_t flags,
                                         float *d_rgba_out, void *stream, sdfhip_stats *stats);

/* Replaces: the display pass, SdfBox/Shaders/DisplayFrag.hlsl:16-24 drawn by
 * Program.cs:96-99, fused into the ray-march epilogue: the frame comes back as
 * R8G8B8A8_UNorm bytes (W*H*4, row-major), `pow(val, 1/2.2)` per channel, or with
 * debug != 0 the step-count heat map `(1,1,1,0) * val.w / 140`.  4x fewer bytes to
 * store, gather and copy to the host.  (The same output is selected on the
 * sdfhip_render / sdfhip_render_device calls by SDFHIP_FLAG_DISPLAY[_DEBUG]; their
 * output pointer then addresses W*H uint32 pixels.) */
SDFHIP_API int sdfhip_render_display(sdfhip_scene *scene, const sdfhip_info *info,
                                     uint32_t width, uint32_t height, uint32_t flags, int debug,
                                     uint8_t *rgba8_out, sdfhip_stats *stats);

/* The host's frame array, page-locked: sdfhip_render / sdfhip_render_display into such memory run the frame in
 * row bands whose copies to the host go beside the march (into pageable memory every copy is staged inside the copy call,
 * with the host waiting in it), and below 4 M pixels the march kernel stores its pixels into the array itself -- no device
 * frame, no copy: 1080p 0.735 -> 0.676 ms per call, RGBA8 0.296 -> 0.260.  Replaces nothing in the reference
 * -- Program.cs:94-99 leaves the frame in a GPU texture -- it is what a host that wants the pixels does once, beside
 * its one frame array:
 *   sdfhip_host_alloc(bytes, &p)   page-locked memory of the library's (any device may copy into it);
 *   sdfhip_host_register(p, bytes) page-locks the caller's own array where it lies: it must not move or be freed while
 *                                  registered (C#: a GCHandleType.Pinned handle kept for as long);
 *   sdfhip_host_release(p)         frees / unregisters (p = the start of the range; NULL is a no-op).
 * A destination that is neither is rendered as before: nothing here is required. */
SDFHIP_API int sdfhip_host_alloc(uint64_t bytes, void **out);
SDFHIP_API int sdfhip_host_register(void *p, uint64_t bytes);
SDFHIP_API int sdfhip_host_release(void *p);

/* Several frames in one launch: infos[0..n_frames-1] (n_frames <= 8) are rendered into
 * d_rgba_out[f * nrows_out * width ...], each with its own camera block; same band
 * arguments as sdfhip_render_device.  For sharded rendering, where one rank's share of a
 * frame is little work behind the serial tail of its longest pixels: the frames of a
 * gather group share that tail.  Plain kernel only (no COMPACT / COUNT flags). */
SDFHIP_API int sdfhip_render_batch_device(sdfhip_scene *scene, const sdfhip_info *infos,
                                          uint32_t n_frames, uint32_t width, uint32_t height,
                                          uint32_t band_rows, uint32_t band_first,
                                          uint32_t band_stride, uint32_t nrows_out,
                                          uint32_t flags, float *d_rgba_out, void *stream,
                                          sdfhip_stats *stats);

/* Rank-0 helper for the tile gather of DENSE bands (the path-traced mode; scenes without a full-depth grid): scatter `world`
 * compact band buffers back into row order.  One gather may carry several frames (fewer, larger messages):
 * d_gathered is [world][frames][rows_per_rank][width] pixels, d_frame is [frames][height][width].
 * pixel_bytes = 16 (RGBA32F) or 4 (RGBA8): both sides hold such pixels.  Asynchronous on `stream`. */
SDFHIP_API int sdfhip_deinterleave_device(int device, const void *d_gathered, void *d_frame,
                                          uint32_t width, uint32_t height,
                                          uint32_t band_rows, uint32_t world,
                                          uint32_t rows_per_rank, uint32_t pixel_bytes,
                                          uint32_t frames, void *stream);

/* Layouts that give the ranks unequal shares of the frame (rank 0 also assembles the frame, so it
 * should render less): the bands a rank renders come as an explicit list instead of "every
 * band_stride-th".  Local band i of the compact output (rows i*band_rows .. of nrows_out) is band
 * bands[i] of the frame; n_bands <= 512.  n_frames consecutive Info blocks render in one launch as
 * in sdfhip_render_batch_device; pt != NULL selects the path-traced mode (one frame). */
SDFHIP_API int sdfhip_render_bands_device(sdfhip_scene *scene, const sdfhip_info *infos, uint32_t n_frames,
                                          const sdfhip_pathtrace *pt, uint32_t width, uint32_t height,
                                          uint32_t band_rows, const uint16_t *bands, uint32_t n_bands,
                                          uint32_t nrows_out, uint32_t flags, float *d_rgba_out, void *stream,
                                          sdfhip_stats *stats);

/* sdfhip_deinterleave_device for such a layout: owner[b] = the rank that rendered band b of the frame
 * (ceil(height / band_rows) entries, at most 512; world <= 64); a rank's bands sit in its buffer in
 * increasing band order. */
SDFHIP_API int sdfhip_deinterleave_bands_device(int device, const void *d_gathered, void *d_frame,
                                                uint32_t width, uint32_t height, uint32_t band_rows,
                                                uint32_t world, uint32_t rows_per_rank, const uint8_t *owner,
                                                uint32_t pixel_bytes, uint32_t frames, void *stream);

/* ---- sparse wire shares written by the march kernel itself ------------------------------------------------------------
 * A wave of the default kernel renders one 8x8 tile, which is the unit of the sparse wire format: at its end it holds the tile's
 * 64 wire pixels in registers and writes the tile's mask (one ballot), its code bytes (one 64-byte store) and its non-zero floats
 * (slots from one atomic add) straight into the share -- no dense wire buffer and no sdfhip_wire_compact_device behind the render.
 * One share holds all `frames` frames of a launch: 64-byte header (word 0 = float slots handed out; beyond `capacity` they are
 * dropped, and the count says so) | masks | slot bases | codes (tile order) | floats, so a gather copies the fixed part and as many
 * floats as were used.  The order of the tiles' floats is whatever order the waves finished in; the expanded frame does not
 * depend on it (bit for bit the frame sdfhip_render_device writes).
 *   sdfhip_sparse2_bytes / _floats_offset   size of a share; offset of its float array (= bytes of the fixed part)
 *   sdfhip_render_sparse_device             like sdfhip_render_bands_device (explicit band list, n_frames <= 8 in one launch),
 *                                           output = one share at d_share.  The share's counter (header word 0) is never zeroed
 *                                           by the library: a launch counts on from `count_base`, which must be the counter's
 *                                           value before the launch (0 for a buffer the caller zeroed; after a launch that used n
 *                                           slots, count_base + n, modulo 2^32) -- no memset between the frames of a viewer
 *   sdfhip_deinterleave_sparse2_device      d_shares[r] = rank r's share (device pointers valid on `device`; world <= 16) ->
 *                                           d_frame [frames][height][width] RGBA32F, or with SDFHIP_FLAG_DISPLAY[_DEBUG] in
 *                                           `flags` RGBA8 through the display pass; only_rank >= 0: write that rank's rows only;
 *                                           counts_out (may be NULL; device-accessible, e.g. pinned host memory): receives the
 *                                           `world` counters of the shares as they arrived */
SDFHIP_API uint64_t sdfhip_sparse2_bytes(uint32_t width, uint32_t rows, uint32_t frames, uint32_t capacity);
SDFHIP_API uint64_t sdfhip_sparse2_floats_offset(uint32_t width, uint32_t rows, uint32_t frames);
SDFHIP_API int sdfhip_render_sparse_device(sdfhip_scene *scene, const sdfhip_info *infos, uint32_t n_frames, uint32_t width,
                                           uint32_t height, uint32_t band_rows, const uint16_t *bands, uint32_t n_bands,
                                           uint32_t nrows_out, uint32_t capacity, uint32_t count_base, uint32_t flags, void *d_share,
                                           void *stream);
SDFHIP_API int sdfhip_deinterleave_sparse2_device(int device, const void *const *d_shares, void *d_frame, uint32_t width,
                                                  uint32_t height, uint32_t band_rows, uint32_t world, uint32_t rows_per_rank,
                                                  const uint8_t *owner, uint32_t capacity, uint32_t frames, uint32_t flags,
                                                  int only_rank, uint32_t *counts_out, void *stream);

/* ---- one frame over several GPUs, behind one call -----------------------------------------------------------
 * Replaces: Program.Draw's UpdateBuffer(info) + DispatchSized(W, H, 1) (SdfBox/Program.cs:81,94) when the frame is rendered by
 * the GPUs of a node: the host still makes ONE call per frame.  One process; the scene is replicated on every device at
 * create; the frame's 16-row bands are dealt to the devices; every device renders its bands with the default kernel, which
 * writes the sparse wire share itself; ranks > 0 push their shares into device devices[0]'s memory over their own xGMI links
 * (peer copies on the rank's stream; SDFHIP_MULTI_TRANSPORT=rccl in the environment at create: ncclSend / ncclRecv inside
 * ncclGroupStart/End instead, RCCL loaded with dlopen); devices[0] expands them into the frame in row order.  Inside the
 * library: one host thread per device, the band layout, the gather, the float tail of a share that needed more than was
 * sent, no Python.  The same device may appear several times (a rehearsal of the pipeline on one GPU; not with RCCL).
 * Environment, read at create: SDFHIP_MULTI_TRANSPORT (above), SDFHIP_RCCL_LIB (the RCCL library to dlopen, if not the system's),
 * SDFHIP_MULTI_RCCL_SELF=1 (with ONE device and the RCCL transport: its share travels through ncclSend / ncclRecv to itself -- all
 * of that transport a single GPU can run).  With SDFHIP_GEN_POOL and SDFHIP_KEEP_ENV (the note at the top of this file) these are
 * the only variables the product library reads; every other choice is an argument (sdfhip_upload_options, sdfhip_multi_configure).
 *   sdfhip_multi_render        one frame to a host array: the viewer's call (latency: every device works on this frame)
 *   sdfhip_multi_submit/_wait  groups of n_frames <= 8 frames (one camera block each, one launch per device), up to 4 groups
 *                              in flight (slot 0..3): throughput.  d_frames_out: device memory on devices[0] for
 *                              [n_frames][height][width] pixels, or NULL for the slot's own buffer (returned by _wait, valid
 *                              until the slot's next submit).  _wait blocks until the slot's frames are complete.
 *   ..._path                   the path-traced mode (one frame; gathers dense RGBA32F bands)
 * Flags: 0, SDFHIP_FLAG_DISPLAY[_DEBUG] (RGBA8 frames: the display pass runs where the frame is assembled),
 * SDFHIP_FLAG_TILE_ORDER.  Pixels are bit for bit those of sdfhip_render on one device. */
typedef struct sdfhip_multi sdfhip_multi;
typedef struct sdfhip_multi_stats {
    float total_ms;             /* host clock: submit -> frames complete (sdfhip_multi_render: -> frame in the host array) */
    uint32_t n_devices;
    uint32_t resends;           /* shares whose float tail had to be sent again */
    uint32_t pad_;
    uint64_t gathered_bytes;    /* bytes that crossed into devices[0] */
    float rank_ms[16];          /* per device: first launch -> share sent (HIP events on its stream) */
    uint32_t floats_used[16];   /* per device: float slots of its share */
} sdfhip_multi_stats;
SDFHIP_API int sdfhip_multi_create(const int *devices, uint32_t n_devices, const int32_t *structs, const uint8_t *values,
                                   uint32_t n, sdfhip_multi **out);
/* First contact with the node's links.  Per device r > 0: can it reach devices[0]'s memory (hipDeviceCanAccessPeer), and does a
 * 1 MB pattern pushed the way the gather pushes shares -- hipMemcpyPeerAsync on r's stream, or ncclSend / ncclRecv in a group with
 * the RCCL transport -- arrive in devices[0]'s memory intact (read back and compared on the host)?  sdfhip_multi_create runs it and
 * fails with SDFHIP_ERR_DEVICE naming the pair, instead of leaving a link problem to surface as a wrong frame; callable again at
 * any time no slot is in flight.  links (may be NULL): n_devices entries. */
typedef struct sdfhip_multi_link {
    int32_t device;             /* devices[r]                                                        */
    int32_t peer_access;        /* 1: devices[r] writes devices[0]'s memory directly (xGMI / PCIe P2P); 0: the copy is staged; -1: same device */
    uint32_t ok;                /* the pattern arrived                                               */
    float push_ms;              /* HIP-event time of the 1 MB push on the sender's stream            */
    char pci_bus_id[16];        /* "0000:c1:00.0"                                                    */
} sdfhip_multi_link;
SDFHIP_API int sdfhip_multi_selftest(sdfhip_multi *m, sdfhip_multi_link *links);
SDFHIP_API int sdfhip_multi_free(sdfhip_multi *m);
/* band height (a multiple of 8; default 16) and the share of devices[0], which also assembles the frame, as a fraction of a
 * peer's (default 1); no slot may be in flight */
SDFHIP_API int sdfhip_multi_configure(sdfhip_multi *m, uint32_t band_rows, float rank0_weight);
SDFHIP_API int sdfhip_multi_info(const sdfhip_multi *m, uint32_t *n_devices, int *devices, uint32_t *band_rows,
                                 float *rank0_weight, int *transport /* 0 peer copies, 1 RCCL */);
SDFHIP_API int sdfhip_multi_render(sdfhip_multi *m, const sdfhip_info *info, uint32_t width, uint32_t height, uint32_t flags,
                                   float *rgba_out, sdfhip_multi_stats *stats);
SDFHIP_API int sdfhip_multi_render_path(sdfhip_multi *m, const sdfhip_info *info, const sdfhip_pathtrace *pt, uint32_t width,
                                        uint32_t height, uint32_t flags, float *rgba_out, sdfhip_multi_stats *stats);
SDFHIP_API int sdfhip_multi_submit(sdfhip_multi *m, uint32_t slot, const sdfhip_info *infos, uint32_t n_frames, uint32_t width,
                                   uint32_t height, uint32_t flags, void *d_frames_out);
SDFHIP_API int sdfhip_multi_submit_path(sdfhip_multi *m, uint32_t slot, const sdfhip_info *info, const sdfhip_pathtrace *pt,
                                        uint32_t width, uint32_t height, uint32_t flags, void *d_frame_out);
SDFHIP_API int sdfhip_multi_wait(sdfhip_multi *m, uint32_t slot, void **d_frames, sdfhip_multi_stats *stats);
#ifdef __cplusplus
}
#endif
#endif /* SDFHIP_H */
