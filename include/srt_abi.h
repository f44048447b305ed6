/* srt_abi.h — the C ABI of the MI355X path-tracing library (libsrt_hip.so).
 *
 * It replaces exactly one thing in the reference: the boost.compute/OpenCL dispatch
 * inside `class Tracer` (/root/reference/include/tracer.hpp:26-88,
 * /root/reference/src/tracer.cpp:11-116). Every entry point names the reference
 * member it stands in for. Records are passed as raw bytes in the layouts of
 * srt_types.h. Plain pointers and sizes only: no C++ types, no torch types.
 *
 * Status codes: 0 = ok, anything else = failure; srt_last_error() returns the text
 * (the C++ wrapper simple-raytracer_amd/host/tracer.hpp turns it into an exception,
 * as boost.compute throws in the reference, src/tracer.cpp:20-26).
 *
 * Threading: a handle is NOT thread-safe (the reference drives its Tracer from the
 * SDL main thread only). All device work of a handle is ordered on one HIP stream.
 */
#ifndef SRT_ABI_H
#define SRT_ABI_H

#include "srt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct srt_tracer srt_tracer; /* opaque */

enum {
	SRT_OK = 0,
	SRT_ERR_INVALID = 1, /* bad argument */
	SRT_ERR_HIP = 2,     /* a HIP runtime call failed (no device, OOM, launch error) */
	SRT_ERR_STATE = 3    /* call order violated (e.g. render before a skybox exists) */
};

/* Deterministic work counters of the trace kernel, summed over all render calls since
 * the last srt_reset_counters(). Schedule independent; equal to the oracle's. */
typedef struct srt_counters {
	uint64_t paths;      /* (pixel, sample) pairs traced */
	uint64_t rays;       /* closest-hit queries = path segments (render.cl:404) */
	uint64_t sky;        /* segments that escaped to the sky (render.cl:463-467) */
	uint64_t tri_tests;  /* triangle tests executed (render.cl:331) */
	uint64_t tri_pass_u; /* triangle tests that passed the u-range check (render.cl:260) */
	uint64_t nan_pixels; /* pixels whose colour of a dispatch was NaN */
	uint64_t watchdog;   /* waves that left the work loop through its spin bound; must be 0 */
} srt_counters;

/* ---- life cycle ------------------------------------------------------------- */

/* Tracer::Tracer(width, height) — src/tracer.cpp:11-68. Selects HIP device
 * `device_index`, creates the stream, the float3 canvas (16 B/pixel) and the ARGB8
 * output buffer. width/height are fixed for the life of the handle (no resize). */
int srt_create(int width, int height, int device_index, srt_tracer **out);

/* ~Tracer (implicit in the reference). */
void srt_destroy(srt_tracer *t);

/* Text of the last failure on this handle (or of srt_create when t == NULL). */
const char *srt_last_error(const srt_tracer *t);

/* ---- inputs ------------------------------------------------------------------ */

/* The skybox upload of src/tracer.cpp:42-55, minus the PNG decode: `rgba` is the
 * RGBA32F image stb would have produced (row 0 = bottom after the vertical flip),
 * width*height*4 floats. Sampled like CL_ADDRESS_CLAMP_TO_EDGE | CL_FILTER_LINEAR with
 * normalized coordinates (src/tracer.cpp:47-48) in exact float arithmetic. */
int srt_set_skybox(srt_tracer *t, const float *rgba, int width, int height);

/* Tracer::update_scene — src/tracer.cpp:70-96. Copies the three arrays (any may be
 * empty: an empty scene renders pure sky) and latches *scene (SceneData reaches the
 * kernel only here, never in render). scene->num_shapes is overwritten by n_shapes,
 * as src/tracer.cpp:94 does. Also runs the world-space triangle pre-pass. */
int srt_update_scene(srt_tracer *t, const srt_shape *shapes, size_t n_shapes, const srt_triangle *triangles,
                     size_t n_triangles, const srt_material *materials, size_t n_materials,
                     const srt_scene_data *scene);

/* Acceleration structure for model shapes (new; the reference's first "future plan",
 * README.md:41). Takes effect at the NEXT srt_update_scene.
 *   SRT_ACCEL_NONE (default): every triangle of a model whose box the ray enters is tested
 *       in array order, as render.cl:329-345 -- the parity mode.
 *   SRT_ACCEL_BVH: srt_update_scene builds one bounding-volume hierarchy per model instance
 *       (host, binned SAH) and rays walk it instead of the array. Same triangle test, same
 *       first-in-array-order rule for hits of equal distance, conservative (padded) boxes:
 *       the canvas equals SRT_ACCEL_NONE's except where a grazing ray's rounding error puts
 *       an accepted hit outside its triangle's padded box (not observed on the test scenes;
 *       tests/test_gpu_bvh.py). Spheres, planes and each model's own AABB test are untouched. */
#define SRT_ACCEL_NONE 0
#define SRT_ACCEL_BVH 1
int srt_set_acceleration(srt_tracer *t, int mode);
/* out = {nodes, leaves, depth, host time spent on the hierarchies in microseconds, models built,
 * models re-used, models refitted} for the current scene (zeros without SRT_ACCEL_BVH or without
 * models). A model whose triangles are byte-identical to one of the previous srt_update_scene keeps
 * its hierarchy: as it is when its transform did not change either (camera, material, other-shape
 * edits cost no build), with new boxes around the same tree when it moved (a refit, ~10x cheaper
 * than a build). */
int srt_acceleration_info(const srt_tracer *t, uint64_t out[7]);

/* Host-only (no device needed): the hierarchy srt_update_scene builds under SRT_ACCEL_BVH for ONE
 * model shape, for inspection and tests. `model->type` must be SRT_SHAPE_MODEL and its triangle
 * range must lie inside `triangles[0 .. n_triangles)`. Writes at most nodes_cap nodes and
 * order_cap indices (order[r] = index inside the model of the triangle stored in record r;
 * leaves refer to records) and always sets *n_nodes to the number of nodes of the hierarchy;
 * either output may be NULL to only query that number. */
int srt_bvh_build_host(const srt_shape *model, const srt_triangle *triangles, size_t n_triangles, srt_bvh_node *nodes_out,
                       size_t nodes_cap, uint32_t *order_out, size_t order_cap, size_t *n_nodes);

/* Host-only: the WIDE form of that hierarchy, the one the kernel walks (layout: csrc/device_types.h; block indices
 * relative to the model's first block; an inner block: dwords 0-2 the origin of its grid, 3 the grid's exponents and the
 * number of children, 4-9 the children's boxes as bytes, 10 their tags, 11 the index of child 0 -- siblings lie side by side). blocks_out receives at most blocks_cap blocks
 * of 32 dwords (leaf blocks are zero here: the device writes their triangles), dest_out[r] = (leaf block << 2) | slot of record r (records as in
 * srt_bvh_build_host's order). *root = the root reference (0xffffffff for a model without triangles), *stack_need =
 * the most children a walk can have waiting at once (never above the kernel's stack of 64: a hierarchy that would
 * need more is rebuilt balanced, *balanced = 1; force_balanced != 0 asks for that form directly). */
int srt_bvh_wide_host(const srt_shape *model, const srt_triangle *triangles, size_t n_triangles, int force_balanced, uint32_t *blocks_out,
                      size_t blocks_cap, uint32_t *dest_out, size_t dest_cap, size_t *n_blocks, uint32_t *root, uint32_t *stack_need, int *balanced);

/* Tracer::clear_canvas — src/tracer.cpp:98-101. */
int srt_clear_canvas(srt_tracer *t);

/* ---- the hot path -------------------------------------------------------------- */

/* Tracer::render(ticks_stopped, output) — src/tracer.cpp:103-116: launches the trace
 * kernel (canvas += mean radiance of options->num_samples paths per pixel), the
 * resolve kernel (canvas/ticks_stopped -> ACES -> sqrt -> A,R,G,B bytes) and copies
 * width*height*4 bytes to argb_out; returns when they are in host memory.
 * With a row partition set (below) only the owned rows are written, packed. */
int srt_render(srt_tracer *t, const srt_render_data *options, uint32_t ticks_stopped, uint8_t *argb_out);

/* srt_render without the final wait (the front-end's step right after the path,
 * src/main.cpp:290-337: the host can prepare the next frame while the GPU finishes this
 * one). Everything is enqueued on the handle's stream, including the copy into argb_out,
 * which must stay valid — and is only valid to read — after srt_synchronize(). Use
 * hipHostMalloc'ed memory for a truly asynchronous copy. Frames still execute in order. */
int srt_render_async(srt_tracer *t, const srt_render_data *options, uint32_t ticks_stopped, uint8_t *argb_out);

/* The two halves of srt_render, asynchronous on the handle's stream, for callers
 * that keep results on the device (bench, multi-GPU gather). */
int srt_trace(srt_tracer *t, const srt_render_data *options);      /* `render` kernel, render.cl:483 */
int srt_resolve(srt_tracer *t, uint32_t ticks_stopped);             /* `average` kernel, render.cl:525 */
int srt_synchronize(srt_tracer *t);
/* The `average` kernel over caller-owned device buffers (num_pixels float4 in,
 * num_pixels*4 bytes out), e.g. the gathered full canvas on the root GPU. Async. */
int srt_resolve_external(srt_tracer *t, const void *device_canvas, uint32_t num_pixels, uint32_t ticks_stopped,
                         void *device_argb);

/* Upper bound in bytes for the per-path radiance buffer (12 B per (pixel, sample) of a
 * sample batch). A dispatch whose paths do not fit is run as several batches of samples
 * with the ordered per-pixel sum carried across them: results are identical, only
 * slower. 0 (default) = half of the free HBM at the first srt_trace, at most 96 GiB; the
 * environment variable SRT_RADIANCE_BUDGET_MB overrides that default. */
int srt_set_radiance_budget(srt_tracer *t, size_t bytes);

/* ---- results / introspection ---------------------------------------------------- */

/* Copies the accumulation canvas (owned rows, packed): n_owned_rows*width float4. */
int srt_read_canvas(srt_tracer *t, float *rgba_out);
/* Copies the resolved image (owned rows, packed): n_owned_rows*width*4 bytes. */
int srt_read_argb(srt_tracer *t, uint8_t *argb_out);
int srt_get_counters(srt_tracer *t, srt_counters *out);
/* Select the instrumented trace-kernel variant that also fills tri_tests/tri_pass_u
 * (one extra VALU op per triangle test); results are unchanged. Default off. */
int srt_set_count_triangles(srt_tracer *t, int enable);
int srt_reset_counters(srt_tracer *t);
/* Diagnostics of the trace kernel's scheduling (not part of any result): out[0..4] = rays, sky, paths,
 * tri_tests, tri_pass_u as above; out[5] = 0 (rounds 1-3: paths that outlived their LDS staging buffer; since round 4
 * every path stores its radiance itself); out[6] = iterations of the
 * waves' main loop; out[7] = SHADE phases executed (sums since the last srt_reset_counters); out[8] = persistent
 * waves per CU and out[9] = workgroups of the most recent trace launch; out[10..17] = per-phase wave cycles
 * (extend, sky ring, shade, park, deliver, refill, loop head, whole kernel) of a -DSRT_PHASE_CLOCK build; in the product
 * build of the array-scan kernels out[10..15] = triangle scans of big models, lanes in them, blocks of 64 rays taken out
 * of the launch-end ray pool, records handed in to it, sum over waves of (blocks taken)^2, blocks the last wave took; else 0. */
int srt_debug_counters(srt_tracer *t, uint64_t out[18]);
/* Development builds with -DSRT_REGION_COUNT only (the product build writes nothing and sets *written = 0): for each
 * region of the trace kernel, in the order of SRT_REGION_LIST (csrc/kernels.hip), {times a wave ran it, lanes that ran it}
 * summed since the last srt_reset_counters; scripts/isa_phase_mix.py multiplies them with the static instruction mix. */
int srt_debug_region_counters(srt_tracer *t, uint64_t *out, int capacity, int *written);
/* Device time of the most recent srt_trace (trace kernel(s) + ordered reduction) and of
 * the most recent resolve, from HIP events recorded on the handle's stream (milliseconds).
 * Synchronises the stream. The render calls (srt_render, srt_render_async, srt_render_pipelined) fuse the resolve into
 * the last reduction and record no timer events unless srt_set_kernel_timers(t, 1) was called (four event records are
 * 10-17 us of a 150 us interactive frame): after them both figures read 0 by default. */
int srt_last_kernel_ms(srt_tracer *t, float *trace_ms, float *resolve_ms);
int srt_set_kernel_timers(srt_tracer *t, int enable);
/* The same for srt_trace_kernel alone: one launch per sample batch of the dispatch. Batches that follow one another
 * (a dispatch of one batch, the usual case): the sum of each launch's own event pair, the ordered reductions between them
 * not counted. Overlapping batches (several batches: even and odd ones trace on two streams so that one batch's tail runs
 * under the next batch): the span from the first launch's start to the last one's end. srt_last_trace_launches says which:
 * the number of srt_trace_kernel launches of the last srt_trace and whether they overlapped. */
int srt_last_trace_kernel_ms(srt_tracer *t, float *kernel_ms);
int srt_last_trace_launches(const srt_tracer *t, int *launches, int *overlapped);
/* Device pointers of the handle's buffers, for zero-copy hand-off (e.g. to a
 * torch.distributed gather): canvas = owned_rows*width*16 B, argb = owned_rows*width*4 B. */
int srt_device_buffers(srt_tracer *t, void **canvas, size_t *canvas_bytes, void **argb, size_t *argb_bytes);
/* Use caller-owned device memory (>= owned_rows*width*16 B) as the canvas, and/or a
 * caller-owned hipStream_t for all launches. NULL restores the handle's own. */
int srt_bind_canvas(srt_tracer *t, void *device_canvas, size_t bytes);
int srt_bind_stream(srt_tracer *t, void *hip_stream);

/* ---- multi-GPU row partition (new; the reference is single-device) -------------- */

/* Rank `rank` of `world` owns the row blocks b with b % world == rank, a block being
 * `rows_per_block` scanlines; the canvas/ARGB buffers hold only those rows, packed in
 * increasing block order. Pixel seeds use the GLOBAL pixel index (render.cl:488,496),
 * so any partition reproduces the single-device image bit for bit. Clears the canvas.
 * world == 1 restores the full image. */
int srt_set_partition(srt_tracer *t, int rank, int world, int rows_per_block);

/* Pure host helpers (no GPU needed) describing that layout. */
int srt_partition_owned_rows(int height, int rank, int world, int rows_per_block);
/* padded_rows = rows every rank sends in an equal-count gather (max over ranks). */
int srt_partition_padded_rows(int height, int world, int rows_per_block);
/* Global y of packed local row `local_row` of `rank`, or -1 when it is padding. */
int srt_partition_global_row(int height, int rank, int world, int rows_per_block, int local_row);
/* Unpermute on the host: gathered = world * padded_rows rows of `row_bytes` each
 * (rank-major); image = height rows. */
int srt_partition_unpermute(const void *gathered, void *image, int height, int world, int rows_per_block,
                            size_t row_bytes);

/* ---- collecting a partitioned frame over RCCL / xGMI (new) ----------------------- */

/* One process per GPU: every rank owns a handle with srt_set_partition(rank, world, ...). Any one
 * rank calls srt_comm_unique_id and ships the 128 bytes to the others by whatever channel the host
 * program has (torch.distributed broadcast, MPI, a file); every rank then calls srt_comm_init
 * (collective: it returns when all ranks have joined). RCCL is loaded on first use. */
#define SRT_COMM_ID_BYTES 128
int srt_comm_unique_id(void *id_out);
int srt_comm_init(srt_tracer *t, const void *id, int rank, int world);
/* The ONE collective of the path, enqueued on the handle's stream after srt_trace: ncclGather of the
 * packed canvases (padded_rows * width float4 per rank) to `root`, where a kernel puts the rows back in
 * image order. Collective: every rank calls it. */
int srt_gather(srt_tracer *t, int root);
/* On the root, after srt_gather: the `average` kernel over the whole gathered image (asynchronous);
 * device pointers of the gathered canvas (height*width float4) and its resolved image; blocking
 * read-back of either (NULL = skip). */
int srt_resolve_gathered(srt_tracer *t, uint32_t ticks_stopped);
int srt_gathered_buffers(srt_tracer *t, void **canvas, void **argb);
int srt_read_gathered(srt_tracer *t, float *canvas_out, uint8_t *argb_out);
/* The root's unpermute step alone, on caller-owned device buffers (gathered: world x padded_rows x width
 * float4, rank-major; image: height x width float4); synchronous. For callers that gather by other
 * means, and for tests. */
int srt_unpermute_device(const void *gathered, void *image, int width, int height, int world, int rows_per_block);

/* One process driving several GPUs -- what a front-end that keeps the reference's single `Tracer`
 * object needs (host/tracer.hpp: Tracer(width, height, n_devices)). The group owns one handle per
 * device (devices == NULL: 0 .. n_devices-1, wrapping around when the node has fewer), partitioned in interleaved
 * blocks of rows_per_block rows, and one communicator (ncclCommInitAll). A list that names a device more than once makes
 * VIRTUAL devices: the members share GPUs, RCCL is not used, and the collection is a device-to-device copy per member
 * where the gather would be -- the N > 1 path on fewer GPUs than members, down to one (tests, rehearsals). Scene
 * (prepared on the host once per srt_group_update_scene), skybox and options are replicated;
 * srt_group_render = trace on every device, one gather to the group's first device, resolve there,
 * blocking read-back of width*height*4 bytes: the image equals the single-device one bit for bit. */
typedef struct srt_group srt_group;
int srt_group_create(int width, int height, int n_devices, const int *devices, int rows_per_block, srt_group **out);
void srt_group_destroy(srt_group *g);
const char *srt_group_last_error(const srt_group *g);
int srt_group_size(const srt_group *g);
srt_tracer *srt_group_tracer(srt_group *g, int i);
int srt_group_set_skybox(srt_group *g, const float *rgba, int width, int height);
int srt_group_set_acceleration(srt_group *g, int mode);
int srt_group_update_scene(srt_group *g, const srt_shape *shapes, size_t n_shapes, const srt_triangle *triangles, size_t n_triangles,
                           const srt_material *materials, size_t n_materials, const srt_scene_data *scene);
int srt_group_clear_canvas(srt_group *g);
int srt_group_trace_and_gather(srt_group *g, const srt_render_data *options); /* asynchronous */
int srt_group_render(srt_group *g, const srt_render_data *options, uint32_t ticks_stopped, uint8_t *argb_out);
int srt_group_read_canvas(srt_group *g, float *rgba_out); /* the gathered canvas, height*width float4 */
int srt_group_get_counters(srt_group *g, srt_counters *out); /* summed over the devices */

/* ---- frame pipeline for the interactive loop (new; src/main.cpp:277-337) ----------- */

/* srt_render with the read-back of frame N overlapped with the trace of frame N+1: the call enqueues
 * frame N (trace, resolve into one of two device images, copy to pinned host memory on a second
 * stream) and hands out frame N-1, waiting only for THAT frame's copy. *frame_delivered = index of the
 * frame written to argb_out (0, 1, ...), or -1 on the first call (argb_out untouched). Frames are the
 * same bytes srt_render would have produced, one call later. srt_pipeline_flush waits for and hands out
 * the newest frame still in flight (-1: none). */
int srt_render_pipelined(srt_tracer *t, const srt_render_data *options, uint32_t ticks_stopped, uint8_t *argb_out,
                         long long *frame_delivered);
int srt_pipeline_flush(srt_tracer *t, uint8_t *argb_out, long long *frame_delivered);

/* Device self-test of the deterministic math (tests only). Walks r = 0, stride, ... over
 * all 2^32 RNG outputs: out[0..2] = mismatch counts of the kernel-local sqrt / log / cos
 * specialisations against their generic definitions (must be 0); out[3..7] = sums of the
 * result bit patterns of detmath's log, cos, sqrt, atan2pi, pow on the device, to be
 * compared with the same sums from the host build of csrc/detmath.h; out[8], out[10] = mismatch
 * counts of the kernel's shared-reciprocal division (both forms: the reciprocal refined on the device, and the host's
 * rounded reciprocal of a sphere's radius with its out-of-range marker 0) and its unguarded Box-Muller square root
 * against IEEE `/` and sqrt (must be 0); out[9] = sum of the result bits of the built-in
 * normalize (detmath's division-free rsqrt), again for comparison with the host build; out[11] = mismatch count of the
 * RNG-scaling shortcuts (log of the raw count, theta from the raw count) against the plain forms
 * (must be 0); out[12] = mismatch count of the kernel's reciprocal-root square root against IEEE sqrt on
 * the floats in [2^-96, inf) (all of them at stride 1; must be 0); out[13] = mismatch count of the camera
 * rays' division by the image size through the host's reciprocal against IEEE `/` (must be 0);
 * out[14] = mismatch count of the kernel's branch-free sign() against detmath's on all bit patterns (must be 0);
 * out[15] = mismatch count of the sun lobe's power with a wave-uniform integer exponent against detmath's dm_powi, exponents
 * 1..32 over the RNG outputs, their negatives and all bit patterns (must be 0). */
int srt_selftest_math(srt_tracer *t, uint32_t stride, uint64_t out[16]);

/* Library / build identification, e.g. "srt-hip gfx950 parity fp-contract=off". */
const char *srt_version(void);

#ifdef __cplusplus
}
#endif

#endif /* SRT_ABI_H */
