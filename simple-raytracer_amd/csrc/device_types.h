/* device_types.h — device-side scene representation shared by kernels.hip and
 * srt_abi.hip. Not part of the public ABI. */
#ifndef SRT_DEVICE_TYPES_H
#define SRT_DEVICE_TYPES_H

#include <stdint.h>

#include "../../include/srt_types.h"

/* Scene data for the wave-uniform intersection loop. Every lane of a wave tests the SAME
 * shape at the same time (array order, render.cl:299), so these records are fetched with
 * scalar loads into SGPRs, not staged per lane. Consecutive shapes of one type are packed into
 * 64-byte BLOCKS, walked in array order (so the first-of-equal-t rule of render.cl:306 is kept):
 *   sphere: 4 dwords  {cx, cy, cz, r*r}                    -> 4 spheres per block
 *   plane : 8 dwords  {px, py, pz, 0, nx, ny, nz, 0}       -> 2 planes per block
 *   model : 8 dwords  {min.x, min.y, min.z, first_wtri(bits), max.x, max.y, max.z, count(bits)} -> 2 per block
 *           (with a BVH: the root node's index in place of first_wtri)
 * A block that is not full ends in records that cannot be hit (srt_abi.hip). Block b lives at
 * dword 16 * b of the packed array. Three consecutive blocks form a GROUP with one 16-byte header
 * (BlockGroup): the kernel issues the header load and the three 64-byte block loads together and
 * waits once -- a 7-shape scene is one group, i.e. one scalar-memory round trip per path segment. */
struct ShapeRun { /* host side only: a block while it is being packed */
	int32_t type;
	uint32_t first_shape;
	uint32_t count;
	uint32_t data_off;
};
struct BlockGroup {
	uint32_t code;     /* byte k = block k of the group: (shape type + 1) | shapes in the block << 2 | big model << 5 | its scan stack << 6; 0 = no block */
	uint32_t first[3]; /* index of each block's first shape */
};
static_assert(sizeof(BlockGroup) == 16, "BlockGroup 16 B");

/* Per-shape record for the per-lane winner lookup after the loop (staged in LDS):
 *   sphere: v = centre, w = radius; plane: v = normal; model: unused (rare path reads
 *   the srt_shape itself). */
struct WinnerRec {
	int32_t type;
	int32_t material;
	float vx, vy, vz, w;
	uint32_t first_wtri;
	float inv_w; /* spheres: 1.0f / radius as the host rounds it (an IEEE quotient) when 2^-40 <= |radius| <= 2^40, else 0 (kernels.hip div3_by_rcp) */
};
static_assert(sizeof(WinnerRec) == 32, "WinnerRec 32 B");

/* World-space triangle written by the pre-pass: v0, e1 = v1 - v0, e2 = v2 - v0 with
 * v = transform * pos in the operation order of render.cl:114-120, so each value is
 * bit-identical to what the reference recomputes per ray (render.cl:325-328,247-248). */
#define SRT_WTRI_FLOATS 9

/* Optional acceleration structure (srt_set_acceleration, include/srt_abi.h; SURVEY.md 8(f)
 * row 4). Per model instance the host builds a binary SAH hierarchy over the world-space triangles
 * (srt_bvh_node, include/srt_types.h: the form that is cached, refitted and handed out by
 * srt_bvh_build_host) and folds it into the WIDE form the kernel walks: 128-byte blocks, one cache
 * line and one fetch each.
 *   inner block  48 bytes of the 128: the boxes of up to FOUR children as bytes on a power-of-two grid relative to the
 *                block's origin. Dwords 0-2 the origin, 3 the grid's biased exponents (x | y << 8 | z << 16) and in its
 *                top byte the number of children, 4-6 lo.x / lo.y / lo.z of children 0..3 (child k in byte k), 7-9
 *                hi.x / hi.y / hi.z, 10 their tags (child k in byte k), 11 the block index of child 0 (the children's
 *                blocks lie side by side: child k is block first + k). bound = fmaf(byte, 2^(e - 127), origin), rounded
 *                outwards on the host and checked there in exactly this arithmetic, so the decoded box contains the
 *                child's (padded) box of the binary hierarchy; the kernel forms bound - o as fmaf(byte, 2^e, origin - o).
 *                Why bytes: a vector load costs the CU's address unit per lane that executes it, whatever its width;
 *                inner lanes fetch three quarters instead of seven (float boxes: 24 dwords + tags + first).
 *   leaf block   up to three triangles of SRT_BVH_TRI_FLOATS dwords {v0, e1, e2} (values as in
 *                SRT_WTRI_FLOATS above), dwords 0-26; dwords 28-30 (SRT_BVH_LEAF_J) their indices j
 *                inside the model, for the reference's first-in-array-order tie rule and for the
 *                vertex normals; all written by srt_prepass_kernel; unused slots stay zero.
 *                The lanes on a leaf fetch its first SEVEN quarters: the indices are looked at only where a hit
 *                is accepted at exactly the distance of the closest one so far, and by the shading.
 *   tag          5 bits: bit 4 = the child is a leaf block, bits 2-3 = triangles in it, bits 0-1 =
 *                the child's slot k. The walk sorts the children it enters by ONE dword each, the
 *                entry distance's bits with the tag in place of the five lowest (distances are >= 0,
 *                so the bits order like the values; rounding a distance DOWN by 2^-18 of itself only
 *                lets a few more children in), and everything it needs to enter a child later -- the
 *                kind of block, where it is -- is that dword and `first`.
 *   root         the model record's reference, as srt_bvh_wide_host hands it out: SRT_BVH_NONE =
 *                nothing to walk; else block index (absolute, <= SRT_BVH_INDEX_MAX so that byte
 *                offsets fit 32 bits), bit 31 = leaf, bits 28-29 = triangles in the leaf
 * A lane keeps the children it still has to enter on a stack of SRT_BVH_STACK_CAP entries; the host
 * checks every hierarchy against that bound (srt_abi.hip fold_wide) and falls back to a balanced one. */
typedef srt_bvh_node BvhNode; /* include/srt_types.h */
#define SRT_BVH_FIRST_DWORD 11 /* where an inner block keeps the index of its child 0 */
#define SRT_BVH_TRI_FLOATS 9
#define SRT_BVH_LEAF_J 28 /* dword of a leaf block with the first triangle's index inside the model */
#define SRT_BVH_NONE 0xffffffffu
#define SRT_BVH_LEAF_BIT 0x80000000u
#define SRT_BVH_INDEX_MASK 0x0fffffffu
#define SRT_BVH_INDEX_MAX 0x01ffffffu /* 4 GB of blocks */
#define SRT_BVH_KEY_INF 0x7f800000u   /* sort keys at or above: a child the ray does not enter */
#define SRT_BVH_TAG_MASK 31u
#define SRT_BVH_TAG_LEAF 16u
#define SRT_BVH_TAG(ref, slot) ((((ref) >> 31) << 4) | ((((ref) >> 28) & 3u) << 2) | (slot)) /* of a reference in the root's form */
#define SRT_BVH_STACK_CAP 64
#ifndef SRT_BVH_LEAF_MAX
#define SRT_BVH_LEAF_MAX 3 /* triangles per leaf block */
#endif

/* Work counters are kept per persistent wave (workgroup index), one 64-byte line each, and
 * summed on the host when asked for: thousands of waves ending together on three shared atomics
 * cost a small dispatch 100 us (profiles/README.md). Launches of a handle are stream-ordered and a
 * launch has one wave per index, so a plain read-modify-write is enough. */
#define SRT_REGION_MAX 56 /* kernels.hip SRT_REGION_LIST: regions of the trace kernel a -DSRT_REGION_COUNT build counts (waves, lanes) for */
#ifdef SRT_REGION_COUNT
#define SRT_WAVE_CTR_STRIDE (16 + 2 * SRT_REGION_MAX)
#else
#define SRT_WAVE_CTR_STRIDE 16 /* unsigned long long per wave: rays, sky, paths, tri, tri_pass_u, 3 diagnostics, 8 phase clocks (-DSRT_PHASE_CLOCK builds) */
#endif
#define SRT_WAVE_CTR_SLOTS 8192 /* >= CUs * 4 SIMDs * 8 waves; the buffer holds TWO such sets: sample batches alternate (their launches overlap) */

enum { SRT_CTR_RAYS = 0, SRT_CTR_SKY, SRT_CTR_TRI, SRT_CTR_TRI_PASS_U, SRT_CTR_NAN, SRT_CTR_PATHS, SRT_CTR_QUEUE, SRT_CTR_WATCHDOG, SRT_CTR_QUEUE2, SRT_CTR_COUNT }; /* QUEUE / QUEUE2: work cursors of even / odd sample batches */

/* TraceParams.material_flags. A draw whose outcome the scene fixes is not made: the generator is advanced past it (the draws
 * behind it see the same states) and the outcome is a constant. */
#define SRT_MF_NO_SPECULAR 1 /* every material's specular threshold is 0: `specular > random_float` is false for every output of the generator */
#define SRT_MF_PLAIN_COLORS 2 /* every material colour component is finite and not -0: mix(colour, 1, 0) = fma(1 - colour, 0, colour) is the colour itself, bit for bit */

struct TraceParams {
	srt_render_data rd;
	srt_scene_data sd;
	const BlockGroup *runs; /* group headers, num_runs of them */
	const float *run_data;
	const WinnerRec *winners;
	int32_t num_runs;
	int32_t num_materials;
	uint32_t lds_bytes; /* dynamic LDS given to the launch; 0 = winners/materials stay in global memory */
	int32_t _pad0;
	const srt_shape *shapes;
	const srt_triangle *triangles;
	const srt_material *materials;
	const float *wtris;
	const float *sky; /* RGBA32F */
	float *canvas;    /* float4 per owned pixel, packed rows (written by the reduce kernel) */
	float *radiance;  /* 3 floats per work-item of the current batch: [pixel][sample in batch] */
	unsigned long long *counters;      /* SRT_CTR_*: only the watchdog counter is touched by the trace kernel */
	unsigned long long *wave_counters; /* [SRT_WAVE_CTR_SLOTS][SRT_WAVE_CTR_STRIDE] */
	unsigned long long *queue;       /* global work cursor (items), zeroed before every launch */
	unsigned long long total_items;  /* owned pixels * batch_samples */
	uint32_t batch_samples;          /* samples per pixel in this batch */
	uint32_t first_sample;           /* sample index of the batch's first sample */
	uint32_t job_items;              /* items a wave reserves per atomic (multiple of the LDS sub-job size) */
	uint32_t stage_off;              /* float4 offset of the per-wave queues (sky ring, hit queue) inside dynamic LDS, behind the scene records */
	int32_t sky_w, sky_h;
	/* wave-uniform values precomputed on the host so they arrive in SGPRs instead of being
	 * recomputed (and kept in VGPRs / spilled masks) by every persistent wave */
	float f_width, f_height, f_sky_w, f_sky_h; /* exact int -> float conversions */
	int32_t sun_focus_int;                     /* dm_pow_small_int(sd.sun_focus): 1..32, or 0 = general pow */
	int32_t material_flags;                    /* SRT_MF_*: what srt_update_scene found true of EVERY material of the scene (kernels.hip SHADE) */
	int32_t num_models;
	int32_t rank, world, rows_per_block, owned_rows;
	int32_t use_bvh;          /* model records carry a BVH root instead of a first world triangle */
	const float *bvh_blocks;  /* all models' 128-byte blocks (wide hierarchy above) */
	float *scan_queue;        /* array scan: SRT_POOL_CTL_WORDS words of pool control (zeroed before the launch), SRT_POOL_REC_FLOATS of pool records (only when pool_blocks != 0), then
	                             SRT_SCAN_QUEUE_FLOATS per persistent wave (rays that wait for a big model's triangle scan, parked rays) */
	/* camera-ray set-up without per-lane integer or IEEE divisions (kernels.hip CAMERA; srt_abi.hip fills them per launch) */
	float inv_f_width, inv_f_height;  /* 1.0f / f_width, 1.0f / f_height, correctly rounded on the host (srt_div_by_rcp in kernels.hip) */
	uint32_t width_magic, width_shift; /* n / width = (mulhi(n, magic) + n) >> shift for n < 2^31 (srt_magic_u31) */
	uint32_t rpb_magic, rpb_shift;     /* the same for rows_per_block */
	uint32_t nbs_magic16;              /* n / batch_samples = (n * magic16) >> 16 for n < 256, when batch_samples < 128 (else unused) */
	uint32_t pool_blocks;              /* array scan, end of a launch: the waves' leftover rays are pooled, in at most this many blocks of 64 per stack
	                                      (kernels.hip; <= SRT_POOL_BLOCKS; 0 = no pool, every wave scans its own remainder) */
	int32_t unit_materials;            /* the device materials hold bernoulli() thresholds in place of metallic / specular / transmittance */
	int32_t all_materials_ok;          /* no shape with a negative material index: the closest shape is a hit without looking its material up (render.cl:404) */
};

/* n / d for 0 <= n < 2^31 as (mulhi(n, magic) + n) >> shift: Granlund & Montgomery's round-up method with shift = ceil(log2 d);
 * for n below 2^31 the sum cannot overflow 32 bits (mulhi(n, magic) < n). d >= 1. */
static inline void srt_magic_u31(uint32_t d, uint32_t *magic, uint32_t *shift) {
	uint32_t l = 0;
	while (l < 32 && ((uint64_t)1 << l) < d) l++;
	*shift = l;
	*magic = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
}

struct PrepassParams {
	const srt_shape *shapes;
	const srt_triangle *triangles;
	const uint32_t *wtri_offset; /* per shape: first world triangle (models only) */
	float *wtris;
	int32_t num_shapes;
	uint32_t num_triangles; /* size of the triangle array, for bounds clamping */
	const uint32_t *order;  /* BVH layout: record wtri_offset[shape] + s holds triangle order[wtri_offset[shape] + s]; NULL = array order, 9 floats */
	const uint32_t *dest;   /* BVH layout: (leaf block << 2) | slot of that record inside `wtris` (= the block array) */
};

/* Ordered reduction of a batch: per pixel, colour += radiance[pixel][k] for k in order
 * (render.cl:518), carried across batches in `running`; the last batch divides by
 * num_samples and adds into the canvas (render.cl:520-522). */
struct ReduceParams {
	const float *radiance;
	float *running; /* float4 per pixel, only used when a dispatch needs more than one batch */
	float *canvas;
	unsigned long long *counters;
	uint32_t num_pixels;
	uint32_t batch_samples;
	int32_t num_samples;
	uint32_t first_batch, last_batch;
	unsigned long long *queue_reset; /* the work cursor the batch's trace launch used: zeroed here for the launch that uses it next (one
	                                    command less per frame than a memset of 8 bytes in front of every trace launch) */
	uint8_t *argb;                   /* last batch only, may be NULL: the pixel's A,R,G,B bytes as srt_resolve_kernel makes them of the new
	                                    canvas value (srt_render: one launch less per frame) */
	uint32_t num_steps;              /* the resolve's divisor (ticks_stopped) */
};

struct ResolveParams {
	const float *canvas;
	uint8_t *argb;
	uint32_t num_steps;
	uint32_t num_pixels;
};

void srt_launch_trace(TraceParams p, bool count_triangles, int num_waves, void *stream);
void srt_launch_reduce(const ReduceParams &p, void *stream);
int srt_trace_waves_per_simd(int has_models, int use_bvh);
int srt_trace_resident_waves_per_cu(const TraceParams &p, bool count_triangles); /* from the runtime's occupancy calculator */
int srt_scan_suspend_min(void); /* array scan: models of at least this many triangles sit alone in their block and are flagged big */
int srt_sub_job_items(int has_models, int use_bvh); /* items per sub-job (the unit a wave's chunk is handed to its lanes in); chunks per atomic are multiples of it */
int srt_scan_queue_in_hbm(void); /* 1: TraceParams.scan_queue must point at SRT_SCAN_QUEUE_FLOATS floats per wave of the launch */
/* per persistent wave: two scan stacks of SRT_SQ_CAP records x 20 fields, one park stack of SRT_PK_CAP records x 15 fields.
 * A scan stack is taken back when it holds 64 rays and one EXTEND phase pushes at most 64: 190 is the most one can hold. */
#define SRT_SQ_CAP 192
#define SRT_PK_CAP 256
#define SRT_SCAN_QUEUE_FLOATS (2 * 20 * SRT_SQ_CAP + 15 * SRT_PK_CAP)
#define SRT_POOL_BLOCKS 8192 /* per stack: 524,288 rays (a launch's ~5,000 waves hand in fewer than 64 each, and again after each bounce) */
#define SRT_POOL_CTL_WORDS (16 + 2 * SRT_POOL_BLOCKS)
#define SRT_POOL_REC_FLOATS ((size_t)2 * SRT_POOL_BLOCKS * 20 * 64)
#define SRT_SCAN_SET_FLOATS(waves, pool) ((size_t)SRT_POOL_CTL_WORDS + ((pool) ? SRT_POOL_REC_FLOATS : (size_t)0) + (size_t)(waves) * SRT_SCAN_QUEUE_FLOATS) /* one set (pool: with the launch-end ray pool's records, 84 MB; TraceParams.pool_blocks != 0); a multiple of 4 */
int srt_bvh_suspends(void);
void srt_launch_prepass(const PrepassParams &p, uint64_t total_wtris, void *stream);
void srt_launch_resolve(const ResolveParams &p, void *stream);
void srt_launch_selftest(unsigned long long *out16, uint32_t stride, void *stream);

#endif
