// kernels.hip — hand-written gfx950 kernels for the path-tracing hot path.
//
// What they compute is the reference's `render` and `average` OpenCL kernels
// (/root/reference/src/render.cl:483-535 and the helpers :114-481); how they compute
// it is CDNA4-first and shares no structure with that file:
//
//  * srt_trace_kernel: ONE WORK-ITEM PER (PIXEL, SAMPLE) PATH, persistent waves. The
//    reference's three nested loops (samples x bounces x shapes) are flattened into a
//    single loop over path SEGMENTS; a lane whose path ends (sky miss, bounce limit,
//    show_normals) pulls the next item of the wave's job in the same iteration (ballot +
//    mbcnt compaction), waves pull jobs from one global cursor, so lanes stay busy until
//    the whole dispatch is done and any slice of the frame fills the chip. Each path's
//    radiance is written to HBM (12 B; 25 GB at 1920x1080x1024 spp, what 288 GB are for)
//    and srt_reduce_kernel sums them per pixel in sample order, which keeps the canvas
//    bit-identical to the reference's serial `color += trace(...)` (render.cl:518).
//  * Phases only a few lanes need at a time are not run masked in every iteration but
//    batched through LDS at full occupancy: camera-ray set-up for a whole sub-job when it
//    is opened, and the sky lookup of escaped paths through a 64-entry ring.
//  * The shape loop index is wave-uniform, so shape records and world-space triangles
//    arrive through SCALAR loads (s_load_dwordx*) into SGPRs and feed VALU ops as
//    scalar operands: no per-lane loads, no LDS traffic and no VGPRs for scene data.
//  * Winner data (normal, material) is fetched once per segment AFTER the loop, per
//    lane, instead of at every improving hit as render.cl:311-312,336-343 do: only the
//    last improving hit survives there, so deferring is exact.
//  * Triangles are pre-transformed to world space once per scene (srt_prepass_kernel)
//    in the reference's operation order, removing 63 of ~115 flops per triangle test.
//  * Model shapes optionally carry a BVH (srt_set_acceleration): a per-lane, stack-based walk over
//    four-wide 128-byte blocks (walk_bvh) replaces the array scan, same triangle test, same tie rule.
//  * No MFMA: nothing here is a contraction. Compiled with -ffp-contract=off; every
//    float op is an IEEE add/mul/div/sqrt or a detmath.h routine so that results match
//    the CPU oracle bit for bit (DESIGN.md "Numerics"). The kernel is VALU-issue bound, so
//    instruction count is what is tuned: shared-reciprocal division, guard-free sqrt where
//    the argument allows, no SLP vectorisation (packed fp32 ops are half rate on gfx950).
#include <hip/hip_runtime.h>

#include "detmath.h"
#include "device_types.h"

// ---- regions: where the trace kernel's instructions are executed (development aid) ------------------------------
// SRT_REGION(NAME) marks the start of a stretch of the trace kernel that runs as often as its first statement. In the
// product build it expands to nothing; scripts/isa_phase_mix.py reads the markers' source lines and assigns every
// instruction of the compiled kernel (by its line-table entry) to the region it was written in. A -DSRT_REGION_COUNT
// build counts, per region, how often a wave ran it and with how many lanes (two LDS words per region, summed per wave
// into its counter line; srt_debug_region_counters): static instruction mix x measured frequency = the executed mix.
#define SRT_REGION_LIST(X)                                                                                                             \
	X(PROLOGUE) X(LOOP_HEAD) X(EXTEND_SETUP) X(EXTEND_GROUP) X(EXTEND_SUSPEND)                                                       \
	X(EXTEND_SPHERES2_0) X(EXTEND_SPHERES2_1) X(EXTEND_SPHERES2_2) X(EXTEND_SPHERES4_0) X(EXTEND_SPHERES4_1) X(EXTEND_SPHERES4_2)   \
	X(EXTEND_PLANES_0) X(EXTEND_PLANES_1) X(EXTEND_PLANES_2) X(EXTEND_MODEL_0) X(EXTEND_MODEL_1) X(EXTEND_MODEL_2)                   \
	X(EXTEND_TRI_LOOP) X(EXTEND_TRI_EXACT) X(EXTEND_TRI_DIV) X(EXTEND_BVH_STEP) X(EXTEND_BVH_SPILL) X(EXTEND_BVH_PUSH2) X(EXTEND_BVH_PUSH3) X(EXTEND_BVH_POP) X(EXTEND_FINISH) X(SKY_PUSH) X(SKY_RESOLVE) X(SHADE_HEAD) X(SHADE_POP) X(SHADE_WINNER)   \
	X(SHADE_MESH_NORMAL) X(SHADE_MATERIAL) X(SHADE_BOUNCE) X(SHADE_OPAQUE) X(SHADE_GLASS) X(SHADE_REFRACT) X(SHADE_TAIL) X(PARK)     \
	X(HANDIN) X(HANDIN_ORPHAN) X(REFILL_HEAD) X(REFILL_SCANQ) X(REFILL_POOL) X(REFILL_UNPARK) X(REFILL_LOOP) X(REFILL_OPEN) X(REFILL_FLUSH) X(REFILL_CURSOR)         \
	X(REFILL_TAKE) X(CAMERA) X(LOOP_TAIL) X(EPILOGUE)
enum SrtRegion {
#define SRT_REGION_ENUM(n) R_##n,
	SRT_REGION_LIST(SRT_REGION_ENUM)
#undef SRT_REGION_ENUM
	R_COUNT
};
static_assert(R_COUNT <= SRT_REGION_MAX, "device_types.h SRT_REGION_MAX");
// Scheduling diagnostics (iterations, SHADE phases, stragglers, early write-outs; srt_debug_counters out[5..7]) cost a few
// scalar instructions and a vote per loop iteration: kept out of the product build, on in every development build.
#if defined(SRT_DIAG) || defined(SRT_REGION_COUNT) || defined(SRT_PHASE_CLOCK)
#define SRT_DIAG_ON 1
#else
#define SRT_DIAG_ON 0
#endif
#ifdef SRT_REGION_COUNT
#define SRT_REGION(name) region_hit(region_ctr, R_##name)
#define SRT_REGION_SLOT(name, slot) region_hit(region_ctr, R_##name##_0 + (slot)) // a stretch compiled once per block slot of a group (test_block)
#define SRT_RC_PARAM , uint32_t *region_ctr
#define SRT_RC_ARG , region_ctr
__device__ __forceinline__ void region_hit(uint32_t *ctr, int r) {
	const unsigned long long m = __ballot(1); // the lanes that are here
	if ((int)threadIdx.x == __ffsll((long long)m) - 1) {
		atomicAdd(&ctr[2 * r], 1u);
		atomicAdd(&ctr[2 * r + 1], (uint32_t)__popcll(m));
	}
}
#else
#define SRT_REGION(name)
#define SRT_REGION_SLOT(name, slot)
#define SRT_RC_PARAM
#define SRT_RC_ARG
#endif

namespace {

// The lanes of the wave for which p holds, straight from the compare's SGPR pair. HIP's __ballot / __any take an int: the
// bool is first materialised per lane (v_cndmask 0 / 1) and compared again (v_cmp_ne) -- two VALU instructions for each of the
// ~10 votes of a loop iteration.
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool any64(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

struct f3 {
	float x, y, z;
};

__device__ __forceinline__ f3 mk(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ f3 neg(f3 a) { return mk(-a.x, -a.y, -a.z); }
// the OpenCL built-ins dot and cross as detmath.h pins them (FMA forms: 3 and 6 instructions)
__device__ __forceinline__ float dot3(f3 a, f3 b) { return dm_dot3(a.x, a.y, a.z, b.x, b.y, b.z); }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
	return mk(dm_cross1(a.y, b.z, a.z, b.y), dm_cross1(a.z, b.x, a.x, b.z), dm_cross1(a.x, b.y, a.y, b.x));
}
// Correctly rounded sqrt. hipcc's own expansion (v_sqrt_f32 + two FMA residual tests
// against the neighbouring floats) spends 7 more instructions on 2^32 pre/post scaling
// for inputs below 2^-96 and on a zero/inf class fix-up. For x = 0, x >= 2^-96, inf, NaN
// and x <= -2^-96 the unscaled core already returns the IEEE result (see DESIGN.md
// "Numerics"), so only 0 < |x| < 2^-96 takes the compiler's full sequence (v_sqrt_f32
// flushes denormal inputs: sqrt(-denormal) must be NaN, not -0). Verified against
// __builtin_sqrtf on all 2^32 bit patterns by srt_selftest_math.
__device__ __forceinline__ float sqrt_ieee(float x) {
#ifndef SRT_NO_FAST_SQRT
	const uint32_t mag = dm_f2u(x) & 0x7fffffffu;
	if (__builtin_expect((mag - 1u) < 0x0f7fffffu, 0)) return __builtin_sqrtf(x); // 0 < |x| < 2^-96 @rare
	float s = __builtin_amdgcn_sqrtf(x); // within 1 ulp
	const uint32_t si = dm_f2u(s);
	float down = dm_u2f(si - 1u), up = dm_u2f(si + 1u);
	float vp = __builtin_fmaf(-down, s, x);
	float vs = __builtin_fmaf(-up, s, x);
	s = (vp <= 0.0f) ? down : s;
	s = (vs > 0.0f) ? up : s;
	return s;
#else
	return dm_sqrtf(x);
#endif
}
// sqrt_ieee without its slow-path guard, for arguments known to be 0, inf, NaN or of
// magnitude >= 2^-96.
__device__ __forceinline__ float sqrt_core(float x) {
#ifndef SRT_NO_FAST_SQRT
	float s = __builtin_amdgcn_sqrtf(x);
	const uint32_t si = dm_f2u(s);
	float down = dm_u2f(si - 1u), up = dm_u2f(si + 1u);
	float vp = __builtin_fmaf(-down, s, x);
	float vs = __builtin_fmaf(-up, s, x);
	s = (vp <= 0.0f) ? down : s;
	s = (vs > 0.0f) ? up : s;
	return s;
#else
	return dm_sqrtf(x);
#endif
}

// Correctly rounded sqrt of a NORMAL x >= 2^-96 from v_rsq_f32 and one residual step: y ~ 1/sqrt(x) (1 ulp), s = x y (within
// 2 ulp of the root), then s + (x - s s) (y / 2) rounded once (Markstein's form of the Newton step: the residual comes out of
// one fma, and the correction is far below the distance of any root of a float from a rounding boundary). One transcendental
// and four plain instructions, no compare / select pair (each of which costs wait states on gfx950, where a VALU may not read
// an SGPR or VCC a VALU wrote in the two slots before): sqrt_core above is 1 + 8 and two such pairs.
// EXHAUSTIVE: equal to __builtin_sqrtf on every float in [2^-96, inf) (scripts/microbench/exact_math_probe.hip;
// srt_selftest_math out[12] repeats the sweep inside the library). Outside that range: NaN for negative x and NaN (as IEEE),
// NaN for +inf (IEEE: inf), NaN for +-0 (IEEE: +-0), garbage for tiny x -- callers route those elsewhere or show that NaN and
// the IEEE value act alike where the result goes.
__device__ __forceinline__ float sqrt_rsq(float x) {
	const float y = __builtin_amdgcn_rsqf(x);
	const float s = x * y, h = 0.5f * y;
	const float r = __builtin_fmaf(-s, s, x);
	return __builtin_fmaf(r, h, s);
}
// The same with the reciprocal root clamped to [0, 2^100] (v_med3_f32; a NaN becomes 0): additionally +-0 -> +-0. For
// arguments that are -0, +0 or normal and >= 2^-96: Box-Muller's -2 log u for every u != 0 (u = 1 gives -0).
__device__ __forceinline__ float sqrt_rsq_zero_ok(float x) {
	const float y = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rsqf(x), 0.0f, 0x1p100f);
	const float s = x * y, h = 0.5f * y;
	const float r = __builtin_fmaf(-s, s, x);
	return __builtin_fmaf(r, h, s);
}

// sqrt_rsq / sqrt_rsq_zero_ok of N independent values, stage by stage. gfx950 wants one wait state between a transcendental
// instruction and the first use of its result: left to itself the scheduler emits each root as one chain (v_rsq, s_nop, ...),
// paying the s_nop -- an issue slot like any other -- N times. The barriers keep the N v_rsq together, which covers it.
template <int N, bool ZERO_OK>
__device__ __forceinline__ void sqrt_rsq_n(const float (&x)[N], float (&out)[N]) {
	float y[N], s[N], h[N];
#ifndef SRT_NO_SQRT_STAGES
	__builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
	for (int i = 0; i < N; i++) y[i] = __builtin_amdgcn_rsqf(x[i]);
#ifndef SRT_NO_SQRT_STAGES
	__builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
	for (int i = 0; i < N; i++) {
		if (ZERO_OK) y[i] = __builtin_amdgcn_fmed3f(y[i], 0.0f, 0x1p100f);
		s[i] = x[i] * y[i], h[i] = 0.5f * y[i];
	}
#pragma unroll
	for (int i = 0; i < N; i++) out[i] = __builtin_fmaf(__builtin_fmaf(-s[i], s[i], x[i]), h[i], s[i]);
}

// ---- IEEE division with the operand scaling factored out ------------------------------
// hipcc expands a / b into v_div_scale (x2), v_rcp, two Newton steps on the reciprocal,
// q = a*r with two residual corrections, v_div_fmas and v_div_fixup: 11 instructions, one
// of them transcendental, per quotient. v_div_scale / v_div_fmas only rescale by 2^+-64 when
// an operand or the quotient comes near the ends of the exponent range, and v_div_fixup only
// replaces the result for zero / inf / NaN operands and out-of-range quotients (CDNA3 ISA
// guide, V_DIV_SCALE_F32 / V_DIV_FIXUP_F32). For
//     2^-40 <= |b| <= 2^40   and   2^-60 <= |a| <= 2^50
// none of those cases applies (both normal, exponent(a) > 23, -126 < e_a - e_b < 96, 1/b
// normal), so the expansion reduces to the plain sequence below, bit for bit -- and its
// first three instructions depend on b alone, so quotients that share a denominator share
// them. Outside that box the compiler's division runs. A NaN numerator gives NaN either way.
// srt_selftest_math compares both against `/` on the device (out[8], out[9]).
__device__ __forceinline__ float rcp_refined(float b) {
	float r = __builtin_amdgcn_rcpf(b);
	float e = __builtin_fmaf(-b, r, 1.0f);
	return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float div_core(float a, float b, float r) {
	float q = a * r;
	float m = __builtin_fmaf(-b, q, a);
	q = __builtin_fmaf(m, r, q);
	m = __builtin_fmaf(-b, q, a);
	return __builtin_fmaf(m, r, q);
}
// a / b with y = the CORRECTLY ROUNDED 1 / b (an IEEE quotient made on the host), for 0 <= a < 2^32 that is zero or at least
// 2^-40 and 1 <= b <= 2^31: the camera's (pixel + jitter) / image size. q0 = a y is within 2 ulp of a / b, the first
// correction leaves a faithful quotient (its residual a - b q0 is exact in one fma), and for a faithful q and y = RN(1 / b)
// RN(q + (a - b q) y) is the correctly rounded quotient (Markstein, IBM J. Res. Dev. 34, 1990, theorem 8.5; no step over- or
// underflows in that range; a = +0 gives +0). Five plain instructions instead of the compiler's eleven, one of them
// transcendental. scripts/microbench/exact_math_probe.hip (2^32 quotients over eight image sizes) and srt_selftest_math
// compare it with `/` on the device.
__device__ __forceinline__ float div_by_rcp(float a, float b, float y) { return div_core(a, b, y); }
__device__ __forceinline__ bool div_num_ok(f3 a) {
	const float ax = dm_fabs(a.x), ay = dm_fabs(a.y), az = dm_fabs(a.z);
	const float mn = __builtin_fminf(__builtin_fminf(ax, ay), az); // v_min3 / v_max3: skip NaNs
	const float mx = __builtin_fmaxf(__builtin_fmaxf(ax, ay), az);
	return mn >= 0x1p-60f && mx <= 0x1p50f;
}
// a / b, component-wise
__device__ __forceinline__ f3 div3(f3 a, float b) {
#ifndef SRT_NO_FAST_DIV
	const float ab = dm_fabs(b);
	if (__builtin_expect(div_num_ok(a) && ab >= 0x1p-40f && ab <= 0x1p40f, 1)) {
		const float r = rcp_refined(b);
		return mk(div_core(a.x, b, r), div_core(a.y, b, r), div_core(a.z, b, r));
	}
#endif
	return a / b; // @rare (scripts/isa_phase_mix.py: behind a range guard, counted as never executed)
}
// a / b, component-wise, with y = the host's correctly rounded 1 / b, or 0 when b is outside [2^-40, 2^40] (or not a number): the
// sphere normal (p - c) / r with the radius' reciprocal from the winner record. Inside div3's box for the numerators the two
// residual steps of div_core give the IEEE quotient for y = RN(1 / b) (Markstein, see div_by_rcp) -- one transcendental and two
// fmas fewer than refining v_rcp_f32; everything else takes the compiler's division.
__device__ __forceinline__ f3 div3_by_rcp(f3 a, float b, float y) {
#if !defined(SRT_NO_FAST_DIV) && !defined(SRT_NO_HOST_RCP)
	if (__builtin_expect(div_num_ok(a) && y != 0.0f, 1)) return mk(div_core(a.x, b, y), div_core(a.y, b, y), div_core(a.z, b, y));
	return a / b; // @rare
#else
	return div3(a, b);
#endif
}
// the built-in normalize: a * rsqrt(dot(a, a)) with detmath.h's division-free rsqrt -- 15 plain instructions, no
// transcendental, no guard (before: IEEE sqrt and three IEEE quotients behind a range check)
__device__ __forceinline__ f3 normalize3(f3 a) {
	const float r = dm_rsqrtf(dot3(a, a));
	return mk(a.x * r, a.y * r, a.z * r);
}
__device__ __forceinline__ f3 mix3(f3 x, f3 y, float a) {
	return mk(dm_mix(x.x, y.x, a), dm_mix(x.y, y.y, a), dm_mix(x.z, y.z, a));
}
__device__ __forceinline__ f3 ld3(const srt_float3 &p) { return mk(p.x, p.y, p.z); }
__device__ __forceinline__ f3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }

// The built-in sign (detmath.h dm_sign: 1, -1, the zero itself, 0 for a NaN) without branches: the sign bit over 1.0 where x is
// less or greater than 0 (v_cmp_lg: false for +-0 and NaN), else x where it is a zero, else 0. Five plain instructions (the
// generic form compiles to two nested exec-mask branches); equal to dm_sign on all 2^32 bit patterns (srt_selftest_math out[14]).
__device__ __forceinline__ float sign_fast(float x) {
#ifndef SRT_NO_SIGN_FAST
	const float one = dm_u2f((dm_f2u(x) & 0x80000000u) | 0x3f800000u);
	const float zero_or_x = __builtin_amdgcn_class(x, 0x60) ? x : 0.0f; // class mask: -0 | +0
	return __builtin_islessgreater(x, 0.0f) ? one : zero_or_x;
#else
	return dm_sign(x);
#endif
}

// column-major 4x4 times (v, w): ((m0*v.x + m1*v.y) + m2*v.z) + m3*w  (render.cl:114-120)
__device__ __forceinline__ f3 mat_by_vec(const srt_float4 *m, f3 v, float w) {
	return mk(((m[0].x * v.x + m[1].x * v.y) + m[2].x * v.z) + m[3].x * w,
	          ((m[0].y * v.x + m[1].y * v.y) + m[2].y * v.z) + m[3].y * w,
	          ((m[0].z * v.x + m[1].z * v.y) + m[2].z * v.z) + m[3].z * w);
}

// the same, the matrix given by its columns' x, y, z
__device__ __forceinline__ f3 mat_cols_by_vec(f3 c0, f3 c1, f3 c2, f3 c3, f3 v, float w) {
	return mk(((c0.x * v.x + c1.x * v.y) + c2.x * v.z) + c3.x * w, ((c0.y * v.x + c1.y * v.y) + c2.y * v.z) + c3.y * w,
	          ((c0.z * v.x + c1.z * v.y) + c2.z * v.z) + c3.z * w);
}

// v - 2 (v.n) n  (render.cl:139-141)
__device__ __forceinline__ f3 reflect3(f3 v, f3 n) { return v - n * (2.0f * dot3(v, n)); }

// PCG-RXS-M-XS-32 (render.cl:143-148); (float)UINT_MAX == 2^32
// random_count = (float)r, random_float = random_count / 2^32. The scaling by 2^-32 is exact and
// never underflows (the smallest non-zero count is 1), so it commutes with any later rounding:
// users that can absorb it into a constant or an exponent take the count and save the multiply.
__device__ __forceinline__ float random_float_fwd(uint32_t &seed);
__device__ __forceinline__ uint32_t random_bits(uint32_t &seed) {
	seed = seed * 747796405u + 2891336453u;
	uint32_t r = ((seed >> ((seed >> 28) + 4u)) ^ seed) * 277803737u;
	return (r >> 22) ^ r;
}
__device__ __forceinline__ float random_count(uint32_t &seed) { return (float)random_bits(seed); }
// `probability > random_float(seed)` (render.cl:427-430) as an integer compare: random_float is a monotone function of the
// generator's 32 output bits r, so {r : p > random_float} is a prefix [0, T) of them; T(p) comes with the material from the
// host (srt_update_scene, found by bisection with the same int -> float conversion), the conversion and the scaling are not
// executed. Only for scenes whose probabilities all have T < 2^32 (p <= 1 does); others keep the float compare (unit_materials).
__device__ __forceinline__ bool bernoulli(float p_or_threshold, bool thresholds, uint32_t &seed) {
	if (thresholds) return random_bits(seed) < dm_f2u(p_or_threshold);
	return p_or_threshold > random_float_fwd(seed);
}
__device__ __forceinline__ float random_float(uint32_t &seed) {
	return random_count(seed) * 2.3283064365386963e-10f; // exact: division by 2^32
}
__device__ __forceinline__ float random_float_fwd(uint32_t &seed) { return random_float(seed); }

// dm_logf restricted to what random_float can return: 0 or a normal float in
// [2^-32, 1]. Same operations on that domain as detmath.h's dm_logf (whose negative /
// subnormal / inf / NaN handling can never trigger here), so the same bits.
// EXP_BIAS = 127 for u itself; 159 when handed the count c = u * 2^32 instead (same mantissa,
// exponent 32 higher, zero stays zero).
template <int EXP_BIAS, bool ZERO_OK>
__device__ __forceinline__ float log_unit_biased(float u) {
	const float LN2_HI = 6.93138123e-01f, LN2_LO = 9.05800061e-06f;
	const float L0 = 6.66666687e-01f, L1 = 4.00001287e-01f, L2 = 2.85499692e-01f, L3 = 2.33534276e-01f;
	uint32_t ix = dm_f2u(u);
	int k = (int)(ix >> 23) - EXP_BIAS;
	ix &= 0x007fffffu;
	uint32_t i = (ix + 0x4afb20u) & 0x00800000u;
	float x = dm_u2f(ix | (i ^ 0x3f800000u));
	k += (int)(i >> 23);
	float f = x - 1.0f;
	// f is +0 or a multiple of 2^-24 in [-0.293, 0.415] and 2 + f lies in [1.7, 2.42]: inside the
	// box of div_core (which also returns the +0 the division gives for f = +0)
	const float den = 2.0f + f;
#if !defined(SRT_NO_FAST_DIV) && !defined(SRT_NO_LOG_DIV1)
	// f / den from the raw v_rcp_f32 and ONE residual step. Not a general division: f takes 2^24 values here, and the quotient is
	// the IEEE one for every single u the RNG can return (exhaustive: exact_math_probe.hip "log div D1"; srt_selftest_math
	// out[1] compares this function with dm_logf on all 2^32 of them). 1 + 3 instructions (shared-reciprocal form: 1 + 7).
	const float rc = __builtin_amdgcn_rcpf(den);
	const float q0 = f * rc;
	float s = __builtin_fmaf(__builtin_fmaf(-den, q0, f), rc, q0);
#elif !defined(SRT_NO_FAST_DIV)
	float s = div_core(f, den, rcp_refined(den));
#else
	float s = f / den;
#endif
	float z = s * s;
	float R = z * dm_fmaf(z, dm_fmaf(z, dm_fmaf(z, L3, L2), L1), L0);
	float hfsq = (0.5f * f) * f;
	float dk = (float)k;
	float r = dm_fmaf(dk, LN2_HI, f - (hfsq - dm_fmaf(s, hfsq + R, dk * LN2_LO)));
	if (!ZERO_OK) return r; // the caller deals with u = 0 (for which r is some finite number)
	// Keep the zero test a select: left alone, the compiler sinks the whole polynomial into a branch
	// on u != 0, which also keeps the three logarithms of a bounce from being scheduled together.
	asm volatile("" : "+v"(r));
	return u == 0.0f ? -DM_INF_F : r;
}
__device__ __forceinline__ float log_unit(float u) { return log_unit_biased<127, true>(u); }
__device__ __forceinline__ float log_count(float c) { return log_unit_biased<159, true>(c); } // log(c / 2^32)

// dm_cosf restricted to finite x in [0, 8): detmath.h's range / NaN guard dropped.
__device__ __forceinline__ float cos_2pi(float x) {
	int k = (int)dm_fmaf(x, 6.36619747e-01f, 0.5f);
	float fk = (float)k;
	float r = dm_fmaf(-fk, 1.570796371e+00f, x);
	r = dm_fmaf(-fk, -4.371138829e-08f, r);
	r = dm_fmaf(-fk, -1.715124510e-15f, r);
	float z = r * r;
	int odd = k & 1;
	float c0 = odd ? -1.66666642e-01f : 4.16666642e-02f;
	float c1 = odd ? 8.33272468e-03f : -1.38882792e-03f;
	float c2 = odd ? -1.95828557e-04f : 2.45428964e-05f;
	float p = dm_fmaf(z, dm_fmaf(z, c2, c1), c0);
	float s_res = dm_fmaf(r * z, p, r);
	float c_res = dm_fmaf(z * z, p, dm_fmaf(-0.5f, z, 1.0f));
	float res = odd ? s_res : c_res;
#ifndef SRT_NO_COS_SIGN_BITS
	// -res in quadrants 1 and 2: bit 1 of k + 1, moved to the sign position and xor-ed in (three integer ops and no compare /
	// select pair; same bits as the select for every angle, srt_selftest_math out[2])
	return dm_u2f(dm_f2u(res) ^ ((((uint32_t)k << 30) + 0x40000000u) & 0x80000000u));
#else
	return (((k + 1) >> 1) & 1) ? -res : res;
#endif
}

// Box-Muller, theta drawn first (render.cl:150-154)
__device__ __forceinline__ float random_normal(uint32_t &seed) {
#if !defined(SRT_NO_COUNT_FOLD) && !defined(SRT_NO_FAST_SQRT) && !defined(SRT_NO_RSQ_SQRT)
	float theta = (6.28318548f * 2.3283064365386963e-10f) * random_count(seed); // = 6.28318548f * random_float, bit for bit
	// -2 log u is -0 (u = 1), +inf (u = 0) or in [1.19e-7, 44.4] for every u random_float can return: the root by sqrt_rsq_zero_ok,
	// and u = 0 (whose logarithm is left some finite number here) selected to sqrt(+inf) = +inf afterwards. Equal to the IEEE
	// sqrt(-2 log u) for all 2^32 u: srt_selftest_math out[10].
	const float cnt = random_count(seed);
	float rho = sqrt_rsq_zero_ok(-2.0f * log_unit_biased<159, false>(cnt));
	asm volatile("" : "+v"(rho)); // keep the zero test a select (see log_unit_biased)
	rho = cnt == 0.0f ? DM_INF_F : rho;
#elif !defined(SRT_NO_COUNT_FOLD)
	float theta = (6.28318548f * 2.3283064365386963e-10f) * random_count(seed); // = 6.28318548f * random_float, bit for bit
	// -2 log u is -0, +inf or in [1.19e-7, 44.4] for every u random_float can return: no small-argument guard
	float rho = sqrt_core(-2.0f * log_count(random_count(seed)));
#else
	float theta = 6.28318548f * random_float(seed);
	float rho = sqrt_core(-2.0f * log_unit(random_float(seed)));
#endif
	return rho * cos_2pi(theta);
}

// Three of them (render.cl:156-158: x, y, z in that order, theta before rho each time), the three square roots side by side
__device__ __forceinline__ f3 random_normal3(uint32_t &seed) {
#if !defined(SRT_NO_COUNT_FOLD) && !defined(SRT_NO_FAST_SQRT) && !defined(SRT_NO_RSQ_SQRT) && !defined(SRT_NO_NORMAL3)
	float th[3], cnt[3], arg[3], rho[3];
#pragma unroll
	for (int k = 0; k < 3; k++) {
		th[k] = (6.28318548f * 2.3283064365386963e-10f) * random_count(seed);
		cnt[k] = random_count(seed);
	}
#pragma unroll
	for (int k = 0; k < 3; k++) arg[k] = -2.0f * log_unit_biased<159, false>(cnt[k]);
	sqrt_rsq_n<3, true>(arg, rho);
#pragma unroll
	for (int k = 0; k < 3; k++) {
		asm volatile("" : "+v"(rho[k])); // keep the zero test a select (see log_unit_biased)
		rho[k] = cnt[k] == 0.0f ? DM_INF_F : rho[k];
	}
	return mk(rho[0] * cos_2pi(th[0]), rho[1] * cos_2pi(th[1]), rho[2] * cos_2pi(th[2]));
#else
	const float gx = random_normal(seed);
	const float gy = random_normal(seed);
	const float gz = random_normal(seed);
	return mk(gx, gy, gz);
#endif
}

// fp64 Schlick (render.cl:173-178); r0 = ((1-mu)/(1+mu))^2 is a per-material constant
__device__ __forceinline__ float schlick(float r0, float cos_theta) {
	double x = 1.0 - (double)cos_theta;
	double x5 = x * ((x * x) * (x * x)); // dm_pown_d(x, 5)
	return (float)((double)r0 + (1.0 - (double)r0) * x5);
}


// ---- wave-uniform scene data: 64-byte blocks fetched with ONE scalar load each ----
// The persistent kernel stores to global memory (radiance) inside its main loop, after which
// the compiler can no longer prove that scene data is not clobbered and would fall back to
// per-lane VMEM loads of the same address. Reading through the CONSTANT address space states
// what is true here -- runs, packed records and world triangles are never written by this
// kernel -- and keeps these loads on the scalar unit (s_load_dwordx*, results in SGPRs).
#define SRT_AS_CONST __attribute__((address_space(4)))
struct Blk16 {
	float v[16];
};
struct Tri2 {
	float v[18];
};
template <int N, int ALIGN>
__device__ __forceinline__ void ld_uniform(const float *p, float (&out)[N]) {
	const SRT_AS_CONST float *c = (const SRT_AS_CONST float *)__builtin_assume_aligned(p, ALIGN);
#pragma unroll
	for (int i = 0; i < N; i++) out[i] = c[i];
}
__device__ __forceinline__ Blk16 ld_blk16(const float *p) {
	Blk16 b;
	ld_uniform<16, 64>(p, b.v);
	return b;
}
__device__ __forceinline__ Tri2 ld_tri2(const float *p) {
	Tri2 t;
	ld_uniform<18, 8>(p, t.v);
	return t;
}

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }

// render.cl:180-204 against FOUR spheres held in SGPRs (one 64-byte block {cx, cy, cz, r*r} x 4; the host fills a
// run's last block with spheres of r*r = -inf, whose discriminant is -inf or NaN: never a hit). Straight-line:
// the four tests are independent chains the scheduler can interleave, and the four square roots share ONE
// small-argument guard (sqrt_ieee above) instead of a branch each. Updates the lane's closest hit in array order.
// ---- closest-hit update with floats compared as unsigned integers --------------------------------------------------
// render.cl keeps a hit when `!(t < 0) && t < tmin` (after `disc < 0` / `denom == 0` have returned a miss). For floats that
// are not -0, "t >= 0 and t < tmin" is ONE unsigned compare of the bit patterns: non-negative floats (and +inf) order like
// their bits, every negative float and every NaN has bits above +inf's, and tmin is never negative (it only ever takes a t
// that passed this test; it starts at +inf). A miss reported through a NaN or an infinity needs no test of its own then:
// the root of a negative discriminant is NaN, n.(p - o) / 0 is +-inf or NaN. -0 is the one value the two orders disagree on
// (the reference accepts t = -0 and afterwards rejects every t >= +0 against tmin = -0): a wave that holds one -- as tmin, or
// as a plane's quotient; a sphere's bq -+ sq cannot be -0 when sq > 0 -- runs the reference's own sequence instead.
// Per sphere: sub, add, v_min_u32, compare, two selects (before: sub, add, compare, select, two compares, two selects and the
// wait states of one more compare -> select pair).
__device__ __forceinline__ bool is_neg_zero(float x) { return dm_f2u(x) == 0x80000000u; }
__device__ __forceinline__ void take_if_closer(float t_key, int idx, float &tmin, int &best) {
	if (dm_f2u(t_key) < dm_f2u(tmin)) {
		tmin = t_key;
		best = idx;
	}
}

// render.cl:180-204 against N of the FOUR spheres held in a 64-byte block {cx, cy, cz, r*r} x 4 (the host fills a run's last
// block with spheres of r*r = -inf, whose discriminant is -inf or NaN: never a hit; with N = 2 only the first two are
// looked at). Straight-line: the tests are independent chains the scheduler can interleave. Updates the lane's closest hit
// in array order.
template <int N>
__device__ __forceinline__ void test_spheres(const Blk16 &s, f3 org, f3 dir, int idx0, float &tmin, int &best) {
	float bq[4], disc[4];
#if !defined(SRT_NO_FAST_SQRT) && !defined(SRT_NO_RSQ_SQRT)
	bool slow = is_neg_zero(tmin);
#pragma unroll
	for (int i = 0; i < N; i++) {
		f3 L = mk(s.v[4 * i] - org.x, s.v[4 * i + 1] - org.y, s.v[4 * i + 2] - org.z);
		bq[i] = dot3(L, dir);
		float c = dot3(L, L) - s.v[4 * i + 3];
		disc[i] = bq[i] * bq[i] - c;
		slow = slow || dm_fabs(disc[i]) < 0x1p-96f; // +-0 and 0 < |x| < 2^-96: one compare with |.| as a source modifier (a NaN is not "tiny")
	}
	if (__builtin_expect(any64(slow), 0)) { // (wave-uniform) the reference's sequence, IEEE square root
#pragma unroll
		for (int i = 0; i < N; i++) { // @rare
			const float sq = __builtin_sqrtf(disc[i]); // @rare
			float t = bq[i] - sq; // @rare
			if (t < 0.0f) t = bq[i] + sq; // @rare
			if (!(disc[i] < 0.0f) && !(t < 0.0f) && t < tmin) tmin = t, best = idx0 + i; // @rare
		}
	} else {
		// disc is NaN, negative, +inf or normal and >= 2^-96 here. sqrt_rsq is the IEEE root on the last range and NaN on the others;
		// IEEE sqrt is NaN on the first two and +inf for +inf -- and a discriminant of +inf never updates the hit either way: with
		// sq = +inf, bq -+ inf is -inf then +inf (or NaN), and +inf < tmin is false; with sq = NaN every t is NaN.
		float dd[N], sq[N];
#pragma unroll
		for (int i = 0; i < N; i++) dd[i] = disc[i];
		sqrt_rsq_n<N, false>(dd, sq);
#pragma unroll
		for (int i = 0; i < N; i++) {
			// the smaller root if it is not negative, else the larger: the smaller of the two bit patterns (bq - sq <= bq + sq)
			const uint32_t k = min(dm_f2u(bq[i] - sq[i]), dm_f2u(bq[i] + sq[i]));
			take_if_closer(dm_u2f(k), idx0 + i, tmin, best);
		}
	}
#else
	float sq[4];
	bool tiny = false;
#pragma unroll
	for (int i = 0; i < N; i++) {
		f3 L = mk(s.v[4 * i] - org.x, s.v[4 * i + 1] - org.y, s.v[4 * i + 2] - org.z);
		bq[i] = dot3(L, dir);
		float c = dot3(L, L) - s.v[4 * i + 3];
		disc[i] = bq[i] * bq[i] - c;
#ifndef SRT_NO_FAST_SQRT
		tiny = tiny || ((dm_f2u(disc[i]) & 0x7fffffffu) - 1u) < 0x0f7fffffu; // 0 < |x| < 2^-96
#endif
	}
#ifndef SRT_NO_FAST_SQRT
	if (__builtin_expect(tiny, 0)) {
#pragma unroll
		for (int i = 0; i < N; i++) sq[i] = __builtin_sqrtf(disc[i]); // @rare
	} else {
#pragma unroll
		for (int i = 0; i < N; i++) sq[i] = sqrt_core(disc[i]);
	}
#else
#pragma unroll
	for (int i = 0; i < N; i++) sq[i] = dm_sqrtf(disc[i]);
#endif
#pragma unroll
	for (int i = 0; i < N; i++) {
		float t = bq[i] - sq[i];
		if (t < 0.0f) t = bq[i] + sq[i];
		bool hit = !(disc[i] < 0.0f) && !(t < 0.0f);
		if (hit && t < tmin) {
			tmin = t;
			best = idx0 + i;
		}
	}
#endif
}

// render.cl:206-221 against TWO planes (one 64-byte block {p, 0, n, 0} x 2; a run's last block is filled with a
// plane of normal 0: denom == 0, never a hit). (The unsigned-key update of take_if_closer does not pay here: a plane's
// quotient can be -0, and testing for it costs what the key saves.)
__device__ __forceinline__ void test_planes2(const Blk16 &b, uint32_t count, f3 org, f3 dir, int idx0, float &tmin, int &best) {
#pragma unroll
	for (int i = 0; i < 2; i++) {
		if (i == 1 && count < 2u) break; // (wave-uniform) a run's last block may hold one plane: the filler's test, division included, is skipped
		f3 n = mk(b.v[8 * i + 4], b.v[8 * i + 5], b.v[8 * i + 6]);
		float denom = dot3(n, dir);
		float t = dot3(n, mk(b.v[8 * i] - org.x, b.v[8 * i + 1] - org.y, b.v[8 * i + 2] - org.z)) / denom;
		// render.cl:209 `denom == 0 -> miss` needs no test of its own: x / 0 is +-inf or NaN, and +inf or a NaN is never below
		// tmin, -inf is below 0
#ifndef SRT_PLANE_DENOM_TEST
		bool hit = !(t < 0.0f);
#else
		bool hit = !(dm_fabs(denom) == 0.0f) && !(t < 0.0f);
#endif
		if (hit && t < tmin) {
			tmin = t;
			best = idx0 + i;
		}
	}
}

// render.cl:279-290 with tmax = the lane's current closest t
__device__ __forceinline__ bool test_aabb(float lx, float ly, float lz, float hx, float hy, float hz, f3 org, f3 inv, float tmax) {
	float t0 = 0.0f, t1 = tmax;
	float a1 = (lx - org.x) * inv.x, a2 = (hx - org.x) * inv.x;
	t0 = dm_max(t0, dm_min(a1, a2));
	t1 = dm_min(t1, dm_max(a1, a2));
	a1 = (ly - org.y) * inv.y, a2 = (hy - org.y) * inv.y;
	t0 = dm_max(t0, dm_min(a1, a2));
	t1 = dm_min(t1, dm_max(a1, a2));
	a1 = (lz - org.z) * inv.z, a2 = (hz - org.z) * inv.z;
	t0 = dm_max(t0, dm_min(a1, a2));
	t1 = dm_min(t1, dm_max(a1, a2));
	return t0 < t1;
}

// One Moller-Trumbore test (render.cl:243-275) against a pre-pass triangle in SGPRs.
//
// The reference rejects at `u < 0 || u > 1` with u = fl(fl(1/a) * sh); that needs an IEEE
// reciprocal (11 instructions) before the first reject. For brute force over 10^5
// triangles almost every lane of almost every wave fails that test, so a conservative,
// division-free pre-reject runs first. With sh = dot(s, h):
//   R1  a == 0                                            (the reference's own test)
//   R2  |sh| > 1.001 |a|              =>  |u| > 1         (u > 1 or u < 0: miss either way)
//   R3  sh*a < 0 and |sh| >= 0.001 |a| =>  u < 0, not an underflow to -0
// Each implies the reference's miss for every finite, infinite or denormal a (margins of
// 2^-10 dwarf the 2^-22 worst-case relative error of fl(1/a)*sh; NaNs compare false and
// fall through). Lanes not rejected compute the reference's q and dot(dir, q) and meet two
// more such rejects, on v (R4, R5 below); what is left runs the reference's exact sequence.
// The wave skips each stage when no lane is left (s_cbranch_execz). Results are therefore
// bit-identical.
// Returns true when the reference accepts the triangle; t is then its hit distance.
template <bool COUNT_TRIS>
__device__ __forceinline__ bool moller_trumbore(float v0x, float v0y, float v0z, float e1x, float e1y, float e1z, float e2x, float e2y,
                                                float e2z, f3 org, f3 dir, bool counted, float &t, uint32_t &n_tri_u SRT_RC_PARAM) {
	f3 e1 = mk(e1x, e1y, e1z), e2 = mk(e2x, e2y, e2z);
	f3 h = cross3(dir, e2);
	float a = dot3(e1, h);
	f3 sv = mk(org.x - v0x, org.y - v0y, org.z - v0z);
	float sh = dot3(sv, h);
	float aa = dm_fabs(a), ash = dm_fabs(sh);
	bool reject = (a == 0.0f) || (ash > aa * 1.001f) || ((sh * a < 0.0f) && (ash >= aa * 0.001f));
	bool ok = false;
	if (!reject) {
		SRT_REGION(EXTEND_TRI_EXACT);
		// The reference's q and dot(dir, q) first, and two more division-free rejects on v = f * dv before the IEEE reciprocal
		// (11 instructions that a wave pays as soon as ONE lane is left):
		//   R4  dv*a < 0 and |dv| >= 0.001 |a|  =>  v < 0, not an underflow to -0        (as R3 for u)
		//   R5  |sh + dv| > 1.01 |a|             =>  |u + v| > 1: u + v > 1, or one of u, v is below -0.5
		// (u + v as the reference rounds it differs from (sh + dv) / a by parts in 10^6; NaNs compare false and fall through).
		// Meshes whose triangles are large on screen send a third of all wave-tests past R1-R3 with a handful of lanes each;
		// most of those lanes fail on v. The instrumented variant counts the lanes that pass the u test and keeps them all.
		// (Measured and left out: the same for t -- dt*a < 0 => t not > 0, |dt| > 1.001 tmin |a| => not closer -- costs the
		// wave-tests that get here more than the reciprocals it saves: configs[2] 107.2 -> 108.6 ms, configs[4] 4,297 -> 4,340.)
		f3 q = cross3(sv, e1);
		float dv = dot3(dir, q);
		bool reject2 = false;
#ifndef SRT_NO_TRI_V_REJECT
		if (!COUNT_TRIS) reject2 = ((dv * a < 0.0f) && (dm_fabs(dv) >= aa * 0.001f)) || (dm_fabs(sh + dv) > aa * 1.01f);
#endif
		if (!reject2) {
			SRT_REGION(EXTEND_TRI_DIV);
			float f = 1.0f / a;
			float u = f * sh;
			ok = !(u < 0.0f || u > 1.0f);
			if (COUNT_TRIS) n_tri_u += (ok && counted) ? 1u : 0u; // padding triangles (NaN rays reach here) are not tests
			float v = f * dv;
			ok = ok && !(v < 0.0f || u + v > 1.0f);
			t = f * dot3(e2, q);
			ok = ok && t > 0.0f;
		}
	}
	return ok;
}

template <bool COUNT_TRIS>
__device__ __forceinline__ void test_triangle(float v0x, float v0y, float v0z, float e1x, float e1y, float e1z, float e2x, float e2y,
                                              float e2z, f3 org, f3 dir, int idx, uint32_t j, uint32_t count, float &tmin, int &best,
                                              uint32_t &best_tri, uint32_t &n_tri_u SRT_RC_PARAM) {
	float t = 0.0f;
	if (moller_trumbore<COUNT_TRIS>(v0x, v0y, v0z, e1x, e1y, e1z, e2x, e2y, e2z, org, dir, j < count, t, n_tri_u SRT_RC_ARG) && t < tmin) {
		tmin = t;
		best = idx;
		best_tri = j;
	}
}

// Each model's block of world triangles starts on a multiple of 4 and is padded to a
// multiple of 4 with all-zero triangles (which fail R1), so the loop below needs no tail
// handling. Triangles are fetched two at a time (18 dwords: s_load_dwordx16 + x2) into
// TWO alternating SGPR sets: the load of the next pair is issued before the current
// pair is tested, which hides the scalar-cache / L2 latency that a single buffer would
// expose once per block (the loop is otherwise latency-bound at low occupancy).

template <bool COUNT_TRIS>
__device__ __forceinline__ void test_pair(const Tri2 &t, f3 org, f3 dir, int idx, uint32_t j, uint32_t count, float &tmin, int &best,
                                          uint32_t &best_tri, uint32_t &n_tri_u SRT_RC_PARAM) {
	test_triangle<COUNT_TRIS>(t.v[0], t.v[1], t.v[2], t.v[3], t.v[4], t.v[5], t.v[6], t.v[7], t.v[8], org, dir, idx, j, count, tmin, best, best_tri,
	                          n_tri_u SRT_RC_ARG);
	test_triangle<COUNT_TRIS>(t.v[9], t.v[10], t.v[11], t.v[12], t.v[13], t.v[14], t.v[15], t.v[16], t.v[17], org, dir, idx, j + 1u, count, tmin,
	                          best, best_tri, n_tri_u SRT_RC_ARG);
}

template <bool COUNT_TRIS>
__device__ __forceinline__ void test_triangles(const float *__restrict__ wtris, uint32_t first, uint32_t count, f3 org, f3 dir, int idx,
                                               float &tmin, int &best, uint32_t &best_tri, uint32_t &n_tri_u SRT_RC_PARAM) {
	const float *__restrict__ blk = wtris + (size_t)first * SRT_WTRI_FLOATS;
	const uint32_t npair = ((count + 3u) >> 2) << 1; // pairs, always even
	Tri2 a = ld_tri2(blk);
	for (uint32_t b = 0; b < npair; b += 2) {
		SRT_REGION(EXTEND_TRI_LOOP);
		const Tri2 c = ld_tri2(blk + 18u * (b + 1u)); // in flight while `a` is tested
		test_pair<COUNT_TRIS>(a, org, dir, idx, 2u * b, count, tmin, best, best_tri, n_tri_u SRT_RC_ARG);
		a = ld_tri2(blk + 18u * (b + 2u)); // in flight while `c` is tested (one pair of slack is allocated past the end)
		test_pair<COUNT_TRIS>(c, org, dir, idx, 2u * b + 2u, count, tmin, best, best_tri, n_tri_u SRT_RC_ARG);
	}
}

// ---- BVH walk (opt-in; device_types.h "wide hierarchy") ------------------------------------
// Per lane: rays of a wave are incoherent after the first bounce, so blocks come through per-lane
// loads. What binds the walk is the CU's vector memory pipe -- ONE address unit for its 20 waves, busy
// two thirds of a launch; a divergent load costs it ~7 ns plus ~0.3 ns per lane that executes it, whatever
// its width (profiles/r04_bvh_vmem_probe.md, scripts/microbench/ta_rates.hip) -- so a step is built around as
// few lane-loads as the data allows: an inner block holds the boxes of FOUR children as bytes on a grid of its
// own (48 bytes: three quarters; a ray takes about a quarter of the steps of a binary walk), a leaf block up to
// three triangles (27 dwords: seven quarters, four of them fetched only by the lanes that stand on a leaf).
// Children are visited nearest first by their entry distance; the others wait, with that distance, on a
// per-lane stack in scratch memory whose top entry lives in registers.
// Same Moller-Trumbore as the array scan, so every accepted hit has the same t; what the walk must
// guarantee is that the triangle the array-order scan would settle on is visited and wins:
//  * boxes were padded on the host and the slab test errs towards "hit" (safe inverse for zero
//    direction components, relative slack on the exit distance and on the stacked entry distance);
//  * the scan keeps the FIRST triangle of equal t (strict <, render.cl:254-256): a hit with
//    t == tmin inside the same model replaces the incumbent only if its index j is lower.
struct BvhStackEntry {
	uint32_t key;   // entry distance | tag (device_types.h)
	uint32_t first; // block of the parent's child 0: the entry is block first + (key & 3)
};

// one plane quarter of a block: the hierarchy's base stays in SGPRs, the lane supplies a 32-bit byte offset
__device__ __forceinline__ float4 bvh_quarter(const float4 *__restrict__ blocks, uint32_t byte_offset, uint32_t imm) {
	return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(blocks) + (size_t)byte_offset + imm);
}

// index inside its model of the triangle in record rec = (leaf block << 2) | slot
__device__ __forceinline__ uint32_t bvh_tri_in_model(const float4 *__restrict__ blocks, uint32_t rec) {
	return reinterpret_cast<const uint32_t *>(blocks)[(size_t)(rec >> 2) * 32u + SRT_BVH_LEAF_J + (rec & 3u)];
}

__device__ __forceinline__ void bvh_order2(uint32_t &a, uint32_t &b) {
	const uint32_t lo = a < b ? a : b, hi = a < b ? b : a; // v_min_u32 / v_max_u32
	a = lo, b = hi;
}

#ifndef SRT_BVH_PUSH_FORM
#define SRT_BVH_PUSH_FORM 1 // 0: three unconditional stores, the idle ones into a spare slot: 37.7 / 37.8 ms against 34.7 / 33.9 (measured on the float-box walk)
#endif
template <bool COUNT_TRIS>
__device__ __forceinline__ void walk_bvh(const float4 *__restrict__ blocks, BvhStackEntry *__restrict__ stack, uint32_t root, f3 org, f3 dir, int idx,
                                         float &tmin, int &best, uint32_t &best_rec, uint32_t &n_tri, uint32_t &n_tri_u SRT_RC_PARAM) {
	// 1/d, or +-2^100 where |d| < 2^-100: (lo - o) * inv stays finite (no 0 * inf = NaN), and keeps its sign
	f3 inv;
	inv.x = dm_fabs(dir.x) >= 0x1p-100f ? 1.0f / dir.x : __builtin_copysignf(0x1p100f, dir.x);
	inv.y = dm_fabs(dir.y) >= 0x1p-100f ? 1.0f / dir.y : __builtin_copysignf(0x1p100f, dir.y);
	inv.z = dm_fabs(dir.z) >= 0x1p-100f ? 1.0f / dir.z : __builtin_copysignf(0x1p100f, dir.z);
	// which planes of a box the ray meets first: lo for a positive direction, hi for a negative one
	const bool sx = inv.x < 0.0f, sy = inv.y < 0.0f, sz = inv.z < 0.0f;
	uint32_t cur = root == SRT_BVH_NONE ? SRT_BVH_NONE : (root & SRT_BVH_INDEX_MASK);
	uint32_t cur_key = SRT_BVH_TAG(root, 0u);
	// The youngest waiting entry lives in registers, stack[0 .. sp) holds the older ones. Under them all lies a sentinel that
	// always passes the distance test and leads to block NONE: popping it ends the walk, so no pop asks whether the stack is empty.
	uint32_t top_key = 0u, top_first = SRT_BVH_NONE;
	uint32_t sp = 0u;
	stack[0].key = 0u, stack[0].first = SRT_BVH_NONE; // (what a pop of the sentinel itself reads back into the registers)
	while (cur != SRT_BVH_NONE) {
		SRT_REGION(EXTEND_BVH_STEP);
		bool pending = true; // nothing to enter from here: take the youngest waiting child
		uint32_t next = SRT_BVH_NONE, next_key = 0u;
		bool inner = false;
		uint32_t k0 = SRT_BVH_KEY_INF, k1 = SRT_BVH_KEY_INF, k2 = SRT_BVH_KEY_INF, k3 = SRT_BVH_KEY_INF, first = 0u;
		{
		// An inner block is 48 bytes: every lane fetches three quarters, the lanes that stand on a leaf the other four (what a
		// load costs the CU's address unit it costs per lane that executes it: scripts/microbench/ta_rates.hip).
		const bool leaf = (cur_key & SRT_BVH_TAG_LEAF) != 0u;
		const uint32_t at = cur << 7;
		const float4 q0 = bvh_quarter(blocks, at, 0u), q1 = bvh_quarter(blocks, at, 16u), q2 = bvh_quarter(blocks, at, 32u);
		if (leaf) {
			const uint32_t cnt = (cur_key >> 2) & 3u, rec0 = cur << 2;
			const float4 q3 = bvh_quarter(blocks, at, 48u), q4 = bvh_quarter(blocks, at, 64u);
#if SRT_BVH_LEAF_MAX > 2
			const float4 q5 = bvh_quarter(blocks, at, 80u), q6 = bvh_quarter(blocks, at, 96u); // (only for the one leaf in four that holds a third triangle: 30.6 against 30.3 ms)
#else
			const float4 q5 = q3, q6 = q3;
#endif
			if (COUNT_TRIS) n_tri += cnt;
			auto tri = [&](float v0x, float v0y, float v0z, float e1x, float e1y, float e1z, float e2x, float e2y, float e2z, uint32_t k) {
				float t = 0.0f;
				if (moller_trumbore<COUNT_TRIS>(v0x, v0y, v0z, e1x, e1y, e1z, e2x, e2y, e2z, org, dir, true, t, n_tri_u SRT_RC_ARG)) {
					bool wins = t < tmin;
					if (t == tmin && best == idx) // the reference keeps the FIRST triangle of equal t: the indices inside the model decide (fetched only here)
						wins = bvh_tri_in_model(blocks, rec0 + k) < bvh_tri_in_model(blocks, best_rec);
					if (wins) {
						tmin = t;
						best = idx;
						best_rec = rec0 + k;
					}
				}
			};
			tri(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, 0u);
			if (cnt > 1u) tri(q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, 1u);
			if (cnt > 2u) tri(q4.z, q4.w, q5.x, q5.y, q5.z, q5.w, q6.x, q6.y, q6.z, 2u);
		} else {
			// four children, their boxes as bytes on a power-of-two grid relative to the block's origin: bound - o = fmaf(byte, 2^e,
			// origin - o) (the host rounded the bytes outwards and checked them in fmaf(byte, 2^e, origin)); no product can be
			// 0 * inf here: byte * 2^e is finite or, for the all-embracing boxes of hostile input, +inf
			const uint32_t ex = f2u(q0.w), nk = ex >> 24;
			const float gx = dm_u2f((ex & 255u) << 23), gy = dm_u2f(((ex >> 8) & 255u) << 23), gz = dm_u2f(((ex >> 16) & 255u) << 23);
			const float cx = q0.x - org.x, cy = q0.y - org.y, cz = q0.z - org.z;
			// the planes the ray meets first / last on each axis: lo / hi for a positive direction, hi / lo for a negative one
			const uint32_t nxw = sx ? f2u(q1.w) : f2u(q1.x), fxw = sx ? f2u(q1.x) : f2u(q1.w);
			const uint32_t nyw = sy ? f2u(q2.x) : f2u(q1.y), fyw = sy ? f2u(q1.y) : f2u(q2.x);
			const uint32_t nzw = sz ? f2u(q2.y) : f2u(q1.z), fzw = sz ? f2u(q1.z) : f2u(q2.y);
			const uint32_t tags = f2u(q2.z);
			first = f2u(q2.w);
			auto child = [&](int k, bool there) -> uint32_t {
				auto at_byte = [k](uint32_t word) { return (float)((word >> (8 * k)) & 255u); };
				const float tnx = dm_fmaf(at_byte(nxw), gx, cx) * inv.x, tfx = dm_fmaf(at_byte(fxw), gx, cx) * inv.x;
				const float tny = dm_fmaf(at_byte(nyw), gy, cy) * inv.y, tfy = dm_fmaf(at_byte(fyw), gy, cy) * inv.y;
				const float tnz = dm_fmaf(at_byte(nzw), gz, cz) * inv.z, tfz = dm_fmaf(at_byte(fzw), gz, cz) * inv.z;
				const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz), 0.0f);
				const float tf = __builtin_fminf(__builtin_fminf(__builtin_fminf(tfx, tfy), tfz), tmin);
				const bool hit = tn <= tf * 1.000001f && there;
				return ((hit ? f2u(tn) : SRT_BVH_KEY_INF) & ~SRT_BVH_TAG_MASK) | ((tags >> (8 * k)) & 255u);
			};
			k0 = child(0, true), k1 = child(1, true), k2 = child(2, nk > 2u), k3 = child(3, nk > 3u);
			inner = true;
		}
		}
		if (inner) {
			bvh_order2(k0, k1);
			bvh_order2(k2, k3);
			bvh_order2(k0, k2);
			bvh_order2(k1, k3);
			bvh_order2(k1, k2); // nearest first; the children that are not entered (keys >= KEY_INF) last
			// k0 is entered now. The n others wait, farthest deepest: the registers' entry goes to memory and k1 takes its place,
			// k3 and k2 go between them.
			const uint32_t w1 = k1 < SRT_BVH_KEY_INF ? 1u : 0u, w2 = k2 < SRT_BVH_KEY_INF ? 1u : 0u, w3 = k3 < SRT_BVH_KEY_INF ? 1u : 0u;
			const uint32_t n = w1 + w2 + w3;
#if SRT_BVH_PUSH_FORM == 0
			const uint32_t i0 = w1 ? sp : SRT_BVH_STACK_CAP, i3 = w3 ? sp + 1u : SRT_BVH_STACK_CAP, i2 = w2 ? sp + n - 1u : SRT_BVH_STACK_CAP;
			stack[i0].key = top_key, stack[i0].first = top_first;
			stack[i3].key = k3, stack[i3].first = first;
			stack[i2].key = k2, stack[i2].first = first;
#else
			if (w1) {
				SRT_REGION(EXTEND_BVH_SPILL);
				stack[sp].key = top_key, stack[sp].first = top_first;
			}
			if (w3) {
				SRT_REGION(EXTEND_BVH_PUSH3);
				stack[sp + 1u].key = k3, stack[sp + 1u].first = first;
			}
			if (w2) {
				SRT_REGION(EXTEND_BVH_PUSH2);
				stack[sp + n - 1u].key = k2, stack[sp + n - 1u].first = first;
			}
#endif
			top_key = w1 ? k1 : top_key, top_first = w1 ? first : top_first;
			sp += n;
			if (k0 < SRT_BVH_KEY_INF) next = first + (k0 & 3u), next_key = k0, pending = false;
		}
		// the youngest waiting child that the closest hit so far has not put out of reach (its distance was rounded down: compare
		// against the limit's bits with the tag bits set)
		const uint32_t reach = f2u(tmin * 1.000001f) | SRT_BVH_TAG_MASK;
		while (pending) {
			SRT_REGION(EXTEND_BVH_POP);
			if (top_key <= reach) next = top_first + (top_key & 3u), next_key = top_key, pending = false;
			sp = sp > 0u ? sp - 1u : 0u;
			top_key = stack[sp].key, top_first = stack[sp].first;
		}
		cur = next, cur_key = next_key;
	}
}

// global y of packed local row (include/srt_abi.h srt_set_partition)
__device__ __forceinline__ int global_row(int local_row, int rank, int world, int rpb) {
	int lb = local_row / rpb;
	return (lb * world + rank) * rpb + (local_row - lb * rpb);
}

// Manual float bilinear, OpenCL 3.0 §8.2 CLAMP_TO_EDGE + LINEAR, normalized coords
__device__ __forceinline__ f3 sample_sky(const float *__restrict__ sky, int W, int H, float fW, float fH, float s, float t) {
	float fu = s * fW - 0.5f;
	float fv = t * fH - 0.5f;
	float cu = dm_clamp(fu, -1.0f, fW);
	float cv = dm_clamp(fv, -1.0f, fH);
	if (!(cu == cu)) cu = 0.0f;
	if (!(cv == cv)) cv = 0.0f;
	float x0f = __builtin_floorf(cu), y0f = __builtin_floorf(cv);
	float a = fu - x0f, b = fv - y0f;
	int x0 = (int)x0f, y0 = (int)y0f;
	int i0 = min(max(x0, 0), W - 1), i1 = min(max(x0 + 1, 0), W - 1);
	int j0 = min(max(y0, 0), H - 1), j1 = min(max(y0 + 1, 0), H - 1);
	const float4 *img = reinterpret_cast<const float4 *>(sky);
	float4 T00 = img[(size_t)j0 * W + i0];
	float4 T10 = img[(size_t)j0 * W + i1];
	float4 T01 = img[(size_t)j1 * W + i0];
	float4 T11 = img[(size_t)j1 * W + i1];
	float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
	return mk(dm_bilinear(w00, T00.x, w10, T10.x, w01, T01.x, w11, T11.x), dm_bilinear(w00, T00.y, w10, T10.y, w01, T01.y, w11, T11.y),
	          dm_bilinear(w00, T00.z, w10, T10.z, w01, T01.z, w11, T11.z));
}

// The launch parameters as they lie in the kernel-argument segment, behind a pointer the compiler cannot see through: what
// is read through it is loaded (scalar loads, scalar-cache hits) where it is used instead of living in SGPRs for the whole
// launch. The persistent loop has far more wave-uniform state than SGPRs; parameters only the sky and the camera rays need
// (sun, image size, camera matrix: ~45 dwords) were being spilled to VGPR lanes and read back with a v_readlane each.
#ifndef SRT_COLD_PARAMS_LIVE
__device__ __forceinline__ const SRT_AS_CONST TraceParams *cold_params() {
	const SRT_AS_CONST TraceParams *kp = (const SRT_AS_CONST TraceParams *)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(kp));
	return kp;
}
#define SRT_COLD(p) (*cold_params())
#else
#define SRT_COLD(p) (p)
#endif

// dm_powi (detmath.h) for a WAVE-UNIFORM exponent 1 <= n <= 32: the same squarings and products in double, in the same
// order, but the exponent's bits steer scalar branches. Written as in detmath.h the compiler turns the loop's `first ? b :
// r * b` and the conditional squaring into selects on 64-bit values: ~45 v_cndmask per call, in runs (which stall on gfx950),
// for what is four squarings and two products when n = 25.
__device__ __forceinline__ float powi_uniform(float x, int n) {
#ifndef SRT_NO_POWI_UNIFORM
	uint32_t un = (uint32_t)__builtin_amdgcn_readfirstlane(n);
	double b = (double)x;
	while (!(un & 1u)) { // (n >= 1: there is a set bit) squarings below the lowest set bit
		b = b * b;
		un >>= 1;
	}
	double r = b; // detmath's `first` product
	un >>= 1;
	while (un) {
		b = b * b;
		if (un & 1u) r = r * b;
		un >>= 1;
	}
	return (float)r;
#else
	return dm_powi(x, n);
#endif
}

// render.cl:380-394
__device__ __forceinline__ f3 sky_box(const TraceParams &p_live, f3 dir) {
	const auto &p = SRT_COLD(p_live);
	f3 sun_dir = mk(p.sd.sun_direction.x, p.sd.sun_direction.y, p.sd.sun_direction.z);
	// dm_powf(x, sun_focus) with its (wave-uniform) choice of path made once on the host
	const float lobe_x = dm_max(dot3(dir, neg(sun_dir)), 0.0f);
	float lobe;
	if (p.sun_focus_int > 0) {
		lobe = powi_uniform(lobe_x, p.sun_focus_int); // dm_powf's x == 1 and NaN cases fall out of the products
	} else {
		lobe = dm_powf(lobe_x, p.sd.sun_focus);
	}
	f3 sun = (mk(p.sd.sun_color.x, p.sd.sun_color.y, p.sd.sun_color.z) * lobe) * p.sd.sun_intensity;
	float u = dm_atan2pif(dir.z, dir.x) * 0.5f + 0.5f;
	float v = dir.y * 0.5f + 0.5f;
	return sample_sky((const float *)p.sky, p.sky_w, p.sky_h, p.f_sky_w, p.f_sky_h, u, v) + sun;
}


} // namespace

// ---------------------------------------------------------------------------------
// Trace kernel. 64-thread workgroups = one persistent wave pulling (pixel, sample) items.
// ---------------------------------------------------------------------------------
// The wave alternates between TWO phases that each run on (nearly) all 64 lanes:
//   EXTEND  closest_intersection for every lane that holds a ray;
//   SHADE   the bounce of 64 paths that hit something.
// Rays that escape go to the sky ring (resolved 64 at a time). Rays that hit are shaded
// at once when, together with the paths waiting in the wave's LDS hit queue, they fill
// the wave; the lanes freed by escapes take waiting paths from the queue. Otherwise the hits
// are PARKED in the queue, every lane is free, and all 64 take new camera rays. Either way
// no phase runs for a fraction of the lanes (before: the shading of ~41 hits ran with the
// other lanes masked off in every iteration, and a lane that freed up set up its camera ray
// through an LDS staging slot).
//
// A path that ends stores its 12 bytes of radiance itself (store_radiance below; rounds 1-3 staged sub-jobs of 64 items in
// LDS and wrote whole lines). SUB-JOBS of 64 consecutive items remain the unit in which a wave's chunk is handed to its
// lanes: one scalar division per sub-job places it in the frame, and a lane's pixel is that pixel or the next one.
#ifndef SRT_SUB_PLAIN
#define SRT_SUB_PLAIN 64
#endif
#ifndef SRT_SUB_MODELS
#define SRT_SUB_MODELS 64
#endif
#ifndef SRT_SUB_BVH
#define SRT_SUB_BVH 64 // (round 2 measured 128 faster -- with chunks of 5 sub-jobs: it was the chunk, not the sub-job; srt_abi.hip)
#endif
// SHADE runs when hits + queued paths reach this many lanes (64 = always a full wave)
#ifndef SRT_SHADE_MIN
#define SRT_SHADE_MIN 64
#endif
// paths the hit queue holds; when hits + queued paths exceed it they are shaded even if they do not fill the wave. 64: a phase's
// hits can always be parked, so SHADE runs with a full wave except at the end of a launch (round 4, in the LDS the radiance
// staging buffers used to take; with 40, every phase of 41..63 ready paths was shaded as it was: 59.4 lanes per SHADE phase
// and 59.3 rays per EXTEND phase, now 63.9 and 62.8 -- 5.5 % fewer loop iterations for the same rays)
#ifndef SRT_HQ_CAP
#define SRT_HQ_CAP 64
#endif
#ifndef SRT_HQ_CAP_MODELS
#define SRT_HQ_CAP_MODELS 64 // (40 / 48 / 56 / 64 at full size: configs[2] array scan 103.3 / 103.1 / 103.2 / 103.2 ms -- the scan does not care)
#endif
// Array-scan kernels: a model of at least this many triangles ("big", srt_abi.hip packs it alone in its block) is not
// scanned by the few lanes whose rays happen to enter its box in one EXTEND phase; those rays wait in one of the
// wave's two scan stacks (the host deals the big models out to them) until SRT_SCAN_FULL of them have gathered.
#ifndef SRT_SCAN_SUSPEND_MIN
#define SRT_SCAN_SUSPEND_MIN 128
#endif
// a scan stack is taken back -- by ALL lanes of the wave; the rays they hold meanwhile are parked -- once it holds this many rays
#ifndef SRT_SCAN_FULL
#define SRT_SCAN_FULL 64
#endif
#ifndef SRT_HQ_CAP_BVH
#define SRT_HQ_CAP_BVH 64 // (40 / 48 / 56 / 64: configs[2] BVH 34.9 / 34.6 / 34.6 / 34.3 ms, configs[4] BVH 34.6 / 34.7 / 34.5 / 34.1)
#endif
// entries of the sky ring (<= 64): the ring is resolved when full, one entry per lane
#ifndef SRT_RING_CAP
#define SRT_RING_CAP 64
#endif
// new camera rays are only set up when at least this many lanes are free
#ifndef SRT_REFILL_MIN
#define SRT_REFILL_MIN 32
#endif

// Development aid (-DSRT_PHASE_CLOCK): per-wave cycles spent in each phase of the main loop, summed into the wave's
// counter line (slots 8..15: extend, ring, shade, park, deliver, refill, spare, total); srt_debug_counters reports them.
#ifdef SRT_PHASE_CLOCK
#define SRT_CLK_DECL unsigned long long clk_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, clk_last = __builtin_amdgcn_s_memtime(), clk_start = clk_last
#define SRT_CLK(i)                                                  \
	do {                                                            \
		const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
		clk_t[i] += now_ - clk_last;                                \
		clk_last = now_;                                            \
	} while (0)
#else
#define SRT_CLK_DECL
#define SRT_CLK(i)
#endif

// Radiances leave the kernel one path at a time: a path that ends stores its 12 bytes itself (round 4). Rounds 1-3 staged the
// 64 radiances of a sub-job in LDS and wrote whole 64-byte lines; that cost 1.5 KB of LDS per wave, a flush per sub-job, two
// votes per hand-in, and a store of its own for the 8 % of paths that outlived their buffer anyway. Without it the hit queue
// holds 64 paths in the same LDS (every SHADE phase runs with a full wave), and the kernel is faster at the same bytes:
// configs[1] 122.5 -> 114.4 ms (profiles/README.md, round 4). The stores are write-through (sc1): nothing stays dirty in an
// L2, so a line whose other items arrive later is not read back for ownership (plain / sc1 / nt at full size with the queue
// of 64: 116.7 / 114.4 / - ms; with the queue of 40: 122.6 / 120.9 / 122.3). Each item is stored exactly once, by whichever wave ends
// its path (the array scan's ray pool hands paths between waves), and read by srt_reduce_kernel after the launch.
#ifndef SRT_RADIANCE_STORE
#define SRT_RADIANCE_STORE 1 // 0 plain, 1 sc1, 2 nt
#endif
namespace {
__device__ __forceinline__ void store_radiance(float *__restrict__ radiance, uint32_t item, f3 c) {
	typedef float f3v __attribute__((ext_vector_type(3)));
	f3v v;
	v.x = c.x, v.y = c.y, v.z = c.z;
	float *g = radiance + 3ull * item;
	if (SRT_RADIANCE_STORE == 1) asm volatile("global_store_dwordx3 %0, %1, off sc1" : : "v"(g), "v"(v) : "memory");
	else if (SRT_RADIANCE_STORE == 2) asm volatile("global_store_dwordx3 %0, %1, off nt" : : "v"(g), "v"(v) : "memory");
	else asm volatile("global_store_dwordx3 %0, %1, off" : : "v"(g), "v"(v) : "memory");
}

// ---- lane sets as wave-uniform masks (round 4) --------------------------------------------------------------------------------
// Which lanes hold a ray, hit, escaped, ended ... lives in scalar registers as 64-bit masks. A divergent region is entered
// with in_mask(m) (llvm.amdgcn.inverse.ballot: the mask becomes the exec mask as it is), counts are s_bcnt1, ranks v_mbcnt, set
// algebra is scalar. Rounds 1-3 kept per-lane bools across the phases of the main loop: every vote on such a bool is a
// v_cndmask + v_cmp pair, every merge of two of them a chain of scalar mask instructions, and none of it is free -- measured in
// place (scripts/r04_issue_cost.sh, profiles/r04_issue_cost.json) a scalar instruction costs the launch 1.0-1.3 v_add_f32.
__device__ __forceinline__ bool in_mask(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
__device__ __forceinline__ uint32_t popc64(unsigned long long m) { return (uint32_t)__builtin_popcountll(m); }
__device__ __forceinline__ uint32_t lane_rank(unsigned long long m) { // set bits of m below this lane
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ unsigned long long lanes_below(uint32_t n) { return n >= 64u ? ~0ull : ((1ull << n) - 1ull); } // lanes 0 .. n-1

// Evaluate the sky for the first n queued escapes (n <= 64), one per lane, and finish their
// paths: mask *= sky; color += mask (render.cl:464-465). Called with all 64 lanes in
// wave-uniform control flow.
__device__ __forceinline__ void resolve_ring(const TraceParams &p, const float *__restrict__ ring, uint32_t n, int lane SRT_RC_PARAM) {
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	if ((uint32_t)lane < n) {
		SRT_REGION(SKY_RESOLVE);
		constexpr uint32_t RC = SRT_RING_CAP;
		const f3 d = mk(ring[0 * RC + lane], ring[1 * RC + lane], ring[2 * RC + lane]);
		f3 m = mk(ring[3 * RC + lane], ring[4 * RC + lane], ring[5 * RC + lane]);
		f3 c = mk(ring[6 * RC + lane], ring[7 * RC + lane], ring[8 * RC + lane]);
		const uint32_t item = dm_f2u(ring[9 * RC + lane]);
		m = m * sky_box(p, d);
		c = c + m;
		store_radiance(p.radiance, item, c);
	}
	asm volatile("" ::: "memory");
}
} // namespace

// Waves per SIMD the register allocator is asked for. LDS per wave (sky ring, hit queue, scene records: 7.4 KB in sphere /
// plane scenes) bounds residency at 21 per CU; the sphere kernel's 78 VGPRs would allow a sixth wave per SIMD.
#ifndef SRT_TRACE_WAVES_PER_SIMD
#define SRT_TRACE_WAVES_PER_SIMD 5
#endif
#ifndef SRT_BVH_SUSPEND
#define SRT_BVH_SUSPEND 0 // BVH kernels gather the rays that enter a big model's box before they walk, as the array scan does
#endif
#ifndef SRT_TRACE_WAVES_PER_SIMD_MODELS
#define SRT_TRACE_WAVES_PER_SIMD_MODELS 5
#endif
#ifndef SRT_TRACE_WAVES_PER_SIMD_BVH
#define SRT_TRACE_WAVES_PER_SIMD_BVH 5 // (4: configs[2] 42.0 ms, configs[4] 38.0 ms; 5: 40.8 / 36.5 -- the walk waits for memory, a fifth wave fills the gaps)
#endif

// HAS_MODELS = false compiles every AABB / triangle / mesh-normal path out: scenes of
// spheres and planes (BASELINE configs 0, 1, 3) get a leaner kernel; the host picks the
// instantiation from the scene.
template <bool COUNT_TRIS, bool USE_LDS, bool HAS_MODELS, bool USE_BVH>
__global__ __launch_bounds__(64, USE_BVH ? SRT_TRACE_WAVES_PER_SIMD_BVH : HAS_MODELS ? SRT_TRACE_WAVES_PER_SIMD_MODELS : SRT_TRACE_WAVES_PER_SIMD) void srt_trace_kernel(const TraceParams p) {
	extern __shared__ float4 lds[]; // [2*n_shapes] winner records, [4*n_materials] materials, (sphere / plane scenes: group headers, shape blocks,) sky ring, hit queue
	constexpr uint32_t SUB = USE_BVH ? SRT_SUB_BVH : HAS_MODELS ? SRT_SUB_MODELS : SRT_SUB_PLAIN;
	const int width = p.rd.width;
	const int lane = threadIdx.x;
	const int ns = p.rd.num_samples;
	const int nb = p.rd.num_bounces;
	const int n_shapes = p.sd.num_shapes;
	const bool all_materials_ok = p.all_materials_ok != 0;
	const bool unit_materials = p.unit_materials != 0; // the materials carry integer thresholds in place of their three probabilities (bernoulli)
	const bool no_specular = (p.material_flags & (SRT_MF_NO_SPECULAR | SRT_MF_PLAIN_COLORS)) == (SRT_MF_NO_SPECULAR | SRT_MF_PLAIN_COLORS); // (wave-uniform) see SHADE
	const BlockGroup *__restrict__ runs = p.runs;
	const float *__restrict__ run_data = p.run_data;
	const float *__restrict__ wtris = p.wtris;

	if (USE_LDS) {
		const float4 *__restrict__ gw = reinterpret_cast<const float4 *>(p.winners);
		const float4 *__restrict__ gm = reinterpret_cast<const float4 *>(p.materials);
		for (int i = threadIdx.x; i < 2 * n_shapes; i += 64) lds[i] = gw[i];
		for (int i = threadIdx.x; i < 4 * p.num_materials; i += 64) lds[2 * n_shapes + i] = gm[i];
		if (!HAS_MODELS) { // sphere / plane scenes: group headers and shape blocks too (see EXTEND)
			const float4 *__restrict__ gh4 = reinterpret_cast<const float4 *>(p.runs);
			const float4 *__restrict__ gd4 = reinterpret_cast<const float4 *>(p.run_data);
			float4 *__restrict__ dst = lds + 2 * n_shapes + 4 * p.num_materials;
			for (int i = threadIdx.x; i < p.num_runs; i += 64) dst[i] = gh4[i];
			for (int i = threadIdx.x; i < 12 * p.num_runs; i += 64) dst[p.num_runs + i] = gd4[i];
		}
		__syncthreads();
	}
	// header of the shape group EXTEND tests next; group 0 to begin with (sphere / plane scenes read their shape blocks from LDS)
	uint32_t gh_code = 0;
	int gh_f0 = 0, gh_f1 = 0, gh_f2 = 0;
	if (USE_LDS && !HAS_MODELS && p.num_runs > 0) {
		const float4 hv = lds[2 * n_shapes + 4 * p.num_materials];
		gh_code = (uint32_t)__builtin_amdgcn_readfirstlane((int)f2u(hv.x));
		gh_f0 = __builtin_amdgcn_readfirstlane((int)f2u(hv.y)), gh_f1 = __builtin_amdgcn_readfirstlane((int)f2u(hv.z));
		gh_f2 = __builtin_amdgcn_readfirstlane((int)f2u(hv.w));
	}

	// ---- work distribution: one work-item = one (pixel, sample) path -------------------
	// Items of this dispatch: item = q * batch_samples + k, q = packed owned pixel (row-major),
	// sample = first_sample + k; fewer than 2^32 per launch (the host sizes sample batches so).
	// Consecutive items are consecutive samples of one pixel, so the 64 lanes of a wave start out
	// on (nearly) the same camera ray. Persistent waves reserve chunks of p.job_items items from
	// ONE global cursor (the first chunk of a wave is its own: chunk number = workgroup number, so
	// thousands of waves starting together do not queue up on one atomic) and work through them in
	// sub-jobs of SUB items. srt_reduce_kernel adds the radiances up per pixel in sample order.
	const uint32_t total_items = (uint32_t)p.total_items;
	const uint32_t nbs = p.batch_samples;
	const unsigned long long own_chunks_end = (unsigned long long)gridDim.x * p.job_items;
	uint32_t chunk_cur = 0, chunk_end = 0; // wave-uniform
	if ((unsigned long long)blockIdx.x * p.job_items < (unsigned long long)total_items) {
		chunk_cur = blockIdx.x * p.job_items;
		chunk_end = (total_items - chunk_cur < p.job_items) ? total_items : chunk_cur + p.job_items;
	}
	bool queue_dry = (total_items == 0);
	// the sub-job being handed out (wave-uniform): items [sj_next, sj_end); sj_next is sample sj_off of packed pixel sj_qpix
	uint32_t sj_next = 0, sj_end = 0, sj_off = 0, sj_qpix = 0;
	float *__restrict__ ring = reinterpret_cast<float *>(lds + p.stage_off); // [10][64] escaped paths awaiting their sky lookup
	float *__restrict__ hq = ring + 10u * (uint32_t)SRT_RING_CAP; // [16..18][HQ] paths that hit, awaiting their bounce (FIFO)
	constexpr uint32_t HQ = USE_BVH ? SRT_HQ_CAP_BVH : HAS_MODELS ? SRT_HQ_CAP_MODELS : SRT_HQ_CAP;
#ifdef SRT_REGION_COUNT
	uint32_t *region_ctr = reinterpret_cast<uint32_t *>(hq + (HAS_MODELS ? (USE_BVH ? 18u : 17u) : 16u) * HQ);
	for (int i = lane; i < 2 * SRT_REGION_MAX; i += 64) region_ctr[i] = 0u;
	__syncthreads();
#endif
	// Array-scan kernels: rays waiting for the triangle scan of a big model. A scan costs the wave its triangle count whether
	// one lane takes part or all 64, so a scan is started for a FULL wave of rays only. Every persistent wave owns, in HBM
	// (device_types.h SRT_SCAN_QUEUE_FLOATS):
	//   two scan stacks of SQ records x 20 fields -- the ray with everything closest_intersection has found so far and the
	//   block it continues at; big model number k of the scene uses stack k & 1, so that the rays one stack gives back all
	//   scan the same model (a scene of one or two big models; with more, a stack mixes models and its scans are less full);
	//   one park stack of PK records x 15 fields -- rays that were about to set out when the wave took a scan stack back.
	// In LDS the stacks would cost the kernel most of its waves; a record is written and read once per triangle scan of at
	// least 128 triangles -- microseconds of memory latency against tens of microseconds of scanning. Stores are plain
	// (write-through), loads bypass the vector L1 (a slot is reused, and the L1 keeps no track of this CU's own stores) and
	// wait for the wave's stores first (REFILL below).
	constexpr bool SUSPEND = HAS_MODELS && (!USE_BVH || SRT_BVH_SUSPEND);
	constexpr uint32_t SQ = (uint32_t)SRT_SQ_CAP, PK = (uint32_t)SRT_PK_CAP;
	float *__restrict__ sq_base = SUSPEND ? const_cast<float *>((const float *)SRT_COLD(p).scan_queue) + (size_t)SRT_POOL_CTL_WORDS + (SRT_COLD(p).pool_blocks != 0u ? SRT_POOL_REC_FLOATS : (size_t)0) + (size_t)blockIdx.x * (size_t)SRT_SCAN_QUEUE_FLOATS : nullptr;
	float *__restrict__ pk = sq_base + 2u * 20u * SQ;
	uint32_t sq_count0 = 0, sq_count1 = 0, pk_count = 0; // wave-uniform
	// The END of a launch: a wave that has run out of camera rays holds a remainder of fewer than 64 rays per stack, and every
	// triangle scan for them would run with idle lanes -- in 5,000 waves at once, and again after each of their bounces. The
	// waves pool these rays instead (TraceParams.pool_*, one pool per stack): a wave with nothing else left hands its remainder
	// in, then takes a full block of 64 out if there is one, else leaves. The last wave to leave takes what is left.
	// Nobody waits for anybody: a block is taken only once its 64 records have been published. Waves of different XCDs meet
	// here, whose L2s do not see each other's lines: records and radiances go through sc1 (write-through) stores, acknowledged
	// (s_waitcnt vmcnt(0)) before the agent-scope atomic add that publishes them, and are read with sc1 loads -- no cache
	// write-back or invalidation, which cost microseconds apiece and would be paid by every wave at every hand-over.
	const bool use_pool = SUSPEND && SRT_COLD(p).pool_blocks != 0u;
	bool pool_leave = false, pool_last = false; // wave-uniform: this wave has signed off / is the last one and clears the pool
	uint32_t w_pool_taken = 0, w_pool_given = 0, w_pool_last_taken = 0; // diagnostics: blocks taken out, records handed in, blocks taken as the last wave
	uint32_t ring_count = 0, hq_head = 0, hq_count = 0;                        // wave-uniform

	f3 org = mk(0.f, 0.f, 0.f), dir = mk(0.f, 0.f, 0.f), mask = mk(1.f, 1.f, 1.f), color = mk(0.f, 0.f, 0.f);
	uint32_t seed = 0;
	uint32_t item = 0;
	int bounce = 0;
	int best = -1;
	uint32_t best_tri = 0; // index inside the model; with a BVH: (leaf block << 2) | slot
	BvhStackEntry bvh_stack[USE_BVH ? SRT_BVH_STACK_CAP + 1 : 1]; // per lane, in scratch memory (walk_bvh)
	const float4 *__restrict__ bvh_blocks = reinterpret_cast<const float4 *>(p.bvh_blocks);
	unsigned long long actm = 0ull; // (wave-uniform) the lanes that hold a ray awaiting closest_intersection
	unsigned long long resm = 0ull; // (wave-uniform) SUSPEND: of those, the rays taken back from a scan stack or the pool, which scan the model of their block `pos` now
	float tmin = DM_INF_F; // closest hit so far of the ray under way (kept across a suspension)
	uint32_t pos = 0;      // SUSPEND: first shape block this ray still has to see (0 = a fresh ray)
	// rays / sky / paths are counted per WAVE with popcounts of the exec mask (scalar adds, no
	// VGPRs); only the instrumented triangle counters stay per lane.
	// (paths and sky lookups of a wave stay below the launch's 2^32 items; iterations are diagnostics)
	unsigned long long w_rays = 0;
	uint32_t w_sky = 0, w_paths = 0, w_iter = 0, w_shade = 0;
	uint32_t w_scans = 0, w_scan_lanes = 0; // SUSPEND diagnostics, per lane: triangle scans of big models this lane led / took part in
	uint32_t n_tri = 0, n_tri_u = 0;
	uint32_t idle_spins = 0;
	// SHADE runs when hits + queued paths exceed this: a full wave (or more than the queue holds); anything at all once the work cursor is dry
	constexpr uint32_t SHADE_THR = ((uint32_t)SRT_SHADE_MIN - 1u) < HQ ? ((uint32_t)SRT_SHADE_MIN - 1u) : HQ;
	uint32_t shade_thr = queue_dry ? 0u : SHADE_THR;

	SRT_CLK_DECL;
	SRT_REGION(PROLOGUE);
	for (;;) {
		SRT_REGION(LOOP_HEAD);
		unsigned long long hitm = 0ull, missm = 0ull, finm = 0ull; // (wave-uniform) lanes whose ray hit / escaped, whose path ended in this iteration
		uint32_t susp = 0u; // SUSPEND, per lane: 1 + the scan stack the lane's ray went to in this iteration
		int key = -1;       // per lane: >= 0 when the ray hit a shape that has a material
		if (SRT_DIAG_ON) w_iter++;
		SRT_CLK(6);
#ifdef SRT_DUMMY_KIND
#include "issue_probe.h" // (regime probe, development builds only: 100 extra instructions of one kind per loop iteration)
#endif
		// ================= EXTEND: closest_intersection (render.cl:293-378), winner deferred =================
		if (actm != 0ull) {
			constexpr bool MASKED = HAS_MODELS;
			if (nb <= 0) { // render.cl:403: no bounce loop at all -> colour 0
				finm = actm;
			} else {
				w_rays += SUSPEND ? popc64(actm & ~resm) : popc64(actm); // a resumed ray was counted when it set out
				const bool resumed = SUSPEND ? in_mask(resm) : false;
				// Sphere / plane scenes: the tests run for all 64 lanes, the lanes without a ray compute on whatever they hold and are
				// sorted out by `actm` afterwards -- no exec-mask bookkeeping around the phase. (With models a lane without a ray must
				// not scan or walk.)
				if (!MASKED || in_mask(actm)) {
					SRT_REGION(EXTEND_SETUP);
					{
						if (!SUSPEND || !resumed) {
							tmin = DM_INF_F;
							best = -1;
							best_tri = 0;
							pos = 0;
						}
						bool part = true;       // SUSPEND: false once the ray has gone to the scan queue
						uint32_t sq_pushed0 = 0, sq_pushed1 = 0; // records pushed by this EXTEND phase so far. Uniform among the lanes in here only:
						                                          // the counts, which the lanes outside this branch read too, are brought up to date after it
						f3 inv = mk(0.f, 0.f, 0.f);
						if (HAS_MODELS) inv = mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);

					// Blocks of same-type shapes in array order; header and data of the NEXT block are fetched (scalar
					// loads, one allocated past the end) before this one is tested, so only the first block's load
					// latency is exposed per segment.
					// Groups of three 64-byte blocks of same-type shapes, in array order (device_types.h). The header and
					// the three blocks of a group are fetched with four scalar loads issued together: one scalar-memory
					// round trip per group (a 7-shape scene is one group). The mesh kernels, whose triangle loops need
					// the scalar registers, fetch the blocks of a group one by one instead.
					auto test_block = [&](const Blk16 &b, uint32_t code, int base, uint32_t bidx, int slot) {
						const uint32_t type1 = code & 3u; // shape type + 1; 0 = no block
						const bool on = !SUSPEND || (part && bidx >= pos);
						if (type1 == SRT_SHAPE_SPHERE + 1u) {
							if (on) {
								// (wave-uniform) a run's last block may hold one or two spheres: the fillers' tests are skipped
								if (((code >> 2) & 7u) <= 2u) {
									SRT_REGION_SLOT(EXTEND_SPHERES2, slot);
									test_spheres<2>(b, org, dir, base, tmin, best);
								} else {
									SRT_REGION_SLOT(EXTEND_SPHERES4, slot);
									test_spheres<4>(b, org, dir, base, tmin, best);
								}
							}
						} else if (type1 == SRT_SHAPE_PLANE + 1u) {
							if (on) {
								SRT_REGION_SLOT(EXTEND_PLANES, slot);
								test_planes2(b, (code >> 2) & 7u, org, dir, base, tmin, best);
							}
						} else if (HAS_MODELS && type1 == SRT_SHAPE_MODEL + 1u) {
							SRT_REGION_SLOT(EXTEND_MODEL, slot);
							// the model's own box first, exactly as the reference (render.cl:316-323), then its triangles
							const bool enter0 = on && test_aabb(b.v[0], b.v[1], b.v[2], b.v[4], b.v[5], b.v[6], org, inv, tmin);
							bool scan0 = enter0;
							if (SUSPEND && ((code >> 5) & 1u)) {
								// A big model. Few of a wave's rays enter its box at a time; scanning 10^5 triangles for them would
								// leave the other lanes idle. Those rays wait in the model's scan stack -- with everything
								// closest_intersection has found so far, so that they continue exactly where they left -- until a wave-full
								// has gathered. The scan runs now when the wave holds rays that were taken back for THIS block (a ray that
								// has scanned one big model and enters the next one's box waits again), or when the stack is full.
								const unsigned long long want = ballot64(enter0);
								const uint32_t n_want = (uint32_t)__popcll(want);
								const uint32_t sid = (code >> 6) & 1u; // (wave-uniform) the model's stack
								const uint32_t held = sid ? sq_count1 + sq_pushed1 : sq_count0 + sq_pushed0;
								const bool now = held + n_want > SQ || any64(enter0 && resumed && pos == bidx);
								if (!now) {
									if (enter0) {
										SRT_REGION(EXTEND_SUSPEND);
										float *__restrict__ sq = sq_base + sid * (20u * SQ);
										const uint32_t e = held + __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
										sq[0 * SQ + e] = org.x, sq[1 * SQ + e] = org.y, sq[2 * SQ + e] = org.z;
										sq[3 * SQ + e] = dir.x, sq[4 * SQ + e] = dir.y, sq[5 * SQ + e] = dir.z;
										sq[6 * SQ + e] = mask.x, sq[7 * SQ + e] = mask.y, sq[8 * SQ + e] = mask.z;
										sq[9 * SQ + e] = color.x, sq[10 * SQ + e] = color.y, sq[11 * SQ + e] = color.z;
										sq[12 * SQ + e] = dm_u2f(seed), sq[13 * SQ + e] = dm_u2f((uint32_t)bounce), sq[14 * SQ + e] = dm_u2f(item);
										sq[15 * SQ + e] = tmin, sq[16 * SQ + e] = dm_u2f((uint32_t)best), sq[17 * SQ + e] = dm_u2f(best_tri);
										sq[18 * SQ + e] = dm_u2f(bidx);
										part = false;
										susp = 1u + sid;
									}
									if (sid) sq_pushed1 += n_want;
									else sq_pushed0 += n_want;
									scan0 = false;
								}
							}
							if (SUSPEND && !COUNT_TRIS && ((code >> 5) & 1u)) {
								// per LANE (this is divergent code): summed over the wave at the end
								const unsigned long long sb = ballot64(scan0);
								if (scan0) {
									w_scan_lanes++;
									if ((sb & ((1ull << lane) - 1ull)) == 0ull) w_scans++; // the scan's first lane counts the scan
								}
							}
							if (scan0) {
								if (USE_BVH) {
									walk_bvh<COUNT_TRIS>(bvh_blocks, bvh_stack, f2u(b.v[3]), org, dir, base, tmin, best, best_tri, n_tri, n_tri_u SRT_RC_ARG);
								} else {
									if (COUNT_TRIS) n_tri += f2u(b.v[7]);
									test_triangles<COUNT_TRIS>(wtris, f2u(b.v[3]), f2u(b.v[7]), org, dir, base, tmin, best, best_tri, n_tri_u SRT_RC_ARG);
								}
							}
							if (((code >> 2) & 7u) > 1u && on && test_aabb(b.v[8], b.v[9], b.v[10], b.v[12], b.v[13], b.v[14], org, inv, tmin)) {
								if (USE_BVH) {
									walk_bvh<COUNT_TRIS>(bvh_blocks, bvh_stack, f2u(b.v[11]), org, dir, base + 1, tmin, best, best_tri, n_tri, n_tri_u SRT_RC_ARG);
								} else {
									if (COUNT_TRIS) n_tri += f2u(b.v[15]);
									test_triangles<COUNT_TRIS>(wtris, f2u(b.v[11]), f2u(b.v[15]), org, dir, base + 1, tmin, best, best_tri, n_tri_u SRT_RC_ARG);
								}
							}
						}
					};
					const int n_groups = p.num_runs;
					for (int g = 0; g < n_groups; g++) {
						SRT_REGION(EXTEND_GROUP);
						float gh[4];
						ld_uniform<4, 16>(reinterpret_cast<const float *>(runs + g), gh);
						const uint32_t code = f2u(gh[0]);
						const float *__restrict__ gd = run_data + 48 * g;
						if (!HAS_MODELS && USE_LDS) {
							// Small sphere / plane scenes: the blocks were staged in LDS with the winner records. All lanes read the same
							// address (a broadcast: no bank conflicts) and get the shape data in VGPRs; measured, the scalar-cache
							// round trip of the path below costs a wave ~750 cycles per segment, an LDS read a fraction of that.
							const float4 *__restrict__ lgh = lds + 2 * n_shapes + 4 * p.num_materials;
							const float4 *__restrict__ lb = lgh + n_groups + 12 * g;
							// The header of the group under test lives in scalar registers (gh_*): read BEHIND the previous group's tests, for
							// the group that comes next -- and never again in a scene of one group (up to 12 spheres / 6 planes), whose
							// first block's reads so start at once instead of behind a header read, a wait and four v_readfirstlane.
							const uint32_t lcode = gh_code;
							const int f0 = gh_f0, f1 = gh_f1, f2 = gh_f2;
							auto ld_lds = [&](int k) {
								Blk16 b;
								const float4 q0 = lb[4 * k], q1 = lb[4 * k + 1], q2 = lb[4 * k + 2], q3 = lb[4 * k + 3];
								b.v[0] = q0.x, b.v[1] = q0.y, b.v[2] = q0.z, b.v[3] = q0.w, b.v[4] = q1.x, b.v[5] = q1.y, b.v[6] = q1.z, b.v[7] = q1.w;
								b.v[8] = q2.x, b.v[9] = q2.y, b.v[10] = q2.z, b.v[11] = q2.w, b.v[12] = q3.x, b.v[13] = q3.y, b.v[14] = q3.z, b.v[15] = q3.w;
								return b;
							};
							test_block(ld_lds(0), lcode & 255u, f0, 0u, 0);
							if ((lcode >> 8) & 255u) test_block(ld_lds(1), (lcode >> 8) & 255u, f1, 0u, 1);
							if ((lcode >> 16) & 255u) test_block(ld_lds(2), (lcode >> 16) & 255u, f2, 0u, 2);
							int ng = n_groups;
							asm volatile("" : "+s"(ng)); // (not to be recognised as loop-invariant: unswitching would duplicate the whole loop)
							if (ng > 1) {
								const float4 hv = lgh[g + 1 < n_groups ? g + 1 : 0];
								gh_code = (uint32_t)__builtin_amdgcn_readfirstlane((int)f2u(hv.x));
								gh_f0 = __builtin_amdgcn_readfirstlane((int)f2u(hv.y)), gh_f1 = __builtin_amdgcn_readfirstlane((int)f2u(hv.z));
								gh_f2 = __builtin_amdgcn_readfirstlane((int)f2u(hv.w));
							}
						} else if (!HAS_MODELS) {
							const Blk16 b0 = ld_blk16(gd), b1 = ld_blk16(gd + 16), b2 = ld_blk16(gd + 32);
#ifdef SRT_PHASE_CLOCK_LOADS
							asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
							SRT_CLK(3); // (diagnosis) time until the group's scalar loads have arrived, booked on the PARK slot
#endif
							test_block(b0, code & 255u, (int)f2u(gh[1]), 0u, 0);
							test_block(b1, (code >> 8) & 255u, (int)f2u(gh[2]), 0u, 1);
							test_block(b2, (code >> 16) & 255u, (int)f2u(gh[3]), 0u, 2);
						} else {
							test_block(ld_blk16(gd), code & 255u, (int)f2u(gh[1]), 3u * g, 0);
							if ((code >> 8) & 255u) test_block(ld_blk16(gd + 16), (code >> 8) & 255u, (int)f2u(gh[2]), 3u * g + 1u, 1);
							if ((code >> 16) & 255u) test_block(ld_blk16(gd + 32), (code >> 16) & 255u, (int)f2u(gh[3]), 3u * g + 2u, 2);
						}
					}
					SRT_REGION(EXTEND_FINISH);
						if (!SUSPEND || part) { // else: the ray waits in the scan queue, with all of its state
							// a shape without a material counts as a miss (render.cl:404: material_index >= 0)
							key = best;
							if (!all_materials_ok) { // (wave-uniform; else every shape of the scene has a material: closest shape = hit)
								key = -1;
								if (best >= 0) key = USE_LDS ? (int)f2u(reinterpret_cast<const float *>(lds)[8 * best + 1]) : p.winners[best].material;
							}
							if (!MASKED) org = org + dir * tmin; // (a lane that hit nothing, or held no ray, has no further use for its origin)
							else if (key >= 0) org = org + dir * tmin; // rayhit->position (render.cl:312,343,362)
						}
					}
				}
				// wave-uniform again
				hitm = ballot64(key >= 0) & actm;
				missm = actm & ~hitm;
				if (SUSPEND) {
					const unsigned long long s0 = ballot64(susp == 1u), s1 = ballot64(susp == 2u); // went to scan stack 0 / 1: neither hit nor escaped yet
					sq_count0 += popc64(s0), sq_count1 += popc64(s1);
					missm &= ~(s0 | s1);
					resm = 0ull;
				}
			}
			actm = 0ull;
		}

		SRT_CLK(0);
		// ---- escaped paths queue for the sky (wave-uniform control flow) ----
		if (missm != 0ull) {
			SRT_REGION(SKY_PUSH);
			constexpr uint32_t RC = SRT_RING_CAP;
			const uint32_t n_miss = popc64(missm);
			const uint32_t rank = lane_rank(missm);
			if (ring_count + n_miss > RC) { // does not fit: the sky lookups of what is queued first (ring_count lanes busy)
				resolve_ring(p, ring, ring_count, lane SRT_RC_ARG);
				ring_count = 0;
			}
			if (RC == 64u) {
				// an empty ring holds a whole wave's escapes: one round, no loop
				if (in_mask(missm)) {
					const uint32_t e = ring_count + rank;
					ring[0 * RC + e] = dir.x, ring[1 * RC + e] = dir.y, ring[2 * RC + e] = dir.z;
					ring[3 * RC + e] = mask.x, ring[4 * RC + e] = mask.y, ring[5 * RC + e] = mask.z;
					ring[6 * RC + e] = color.x, ring[7 * RC + e] = color.y, ring[8 * RC + e] = color.z;
					ring[9 * RC + e] = dm_u2f(item);
				}
				ring_count += n_miss;
			} else {
				const bool missed = in_mask(missm);
				uint32_t done = 0;
				while (done < n_miss) { // more lanes may have escaped than the ring holds
					const uint32_t take = (RC - ring_count) < (n_miss - done) ? (RC - ring_count) : (n_miss - done);
					if (missed && rank >= done && rank < done + take) {
						const uint32_t e = ring_count + (rank - done);
						ring[0 * RC + e] = dir.x, ring[1 * RC + e] = dir.y, ring[2 * RC + e] = dir.z;
						ring[3 * RC + e] = mask.x, ring[4 * RC + e] = mask.y, ring[5 * RC + e] = mask.z;
						ring[6 * RC + e] = color.x, ring[7 * RC + e] = color.y, ring[8 * RC + e] = color.z;
						ring[9 * RC + e] = dm_u2f(item);
					}
					ring_count += take;
					done += take;
					if (done < n_miss) {
						resolve_ring(p, ring, ring_count, lane SRT_RC_ARG);
						ring_count = 0;
					}
				}
			}
			w_sky += n_miss;
		}

		SRT_CLK(1);
		// ================= SHADE or PARK =================
		const uint32_t n_hit = popc64(hitm);
		const uint32_t n_ready = n_hit + hq_count;
		if (n_ready > shade_thr) {
			SRT_REGION(SHADE_HEAD);
			unsigned long long shm = hitm; // the lanes that shade: those with a hit, and the free ones that take the oldest waiting paths
			const uint32_t n_free_s = 64u - n_hit;
			const uint32_t n_pop = n_free_s < hq_count ? n_free_s : hq_count;
			if (n_pop != 0u) {
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				const unsigned long long fb = ~hitm;
				const uint32_t rank = lane_rank(fb);
				const unsigned long long popm = ballot64(rank < n_pop) & fb;
				if (in_mask(popm)) {
					SRT_REGION(SHADE_POP);
					uint32_t e = hq_head + rank;
					e = e >= HQ ? e - HQ : e;
					org = mk(hq[0 * HQ + e], hq[1 * HQ + e], hq[2 * HQ + e]);
					dir = mk(hq[3 * HQ + e], hq[4 * HQ + e], hq[5 * HQ + e]);
					mask = mk(hq[6 * HQ + e], hq[7 * HQ + e], hq[8 * HQ + e]);
					color = mk(hq[9 * HQ + e], hq[10 * HQ + e], hq[11 * HQ + e]);
					seed = dm_f2u(hq[12 * HQ + e]);
					best = (int)dm_f2u(hq[13 * HQ + e]);
					bounce = (int)dm_f2u(hq[14 * HQ + e]);
					item = dm_f2u(hq[15 * HQ + e]);
					if (HAS_MODELS) best_tri = dm_f2u(hq[16 * HQ + e]);
				}
				asm volatile("" ::: "memory");
				hq_head += n_pop;
				hq_head = hq_head >= HQ ? hq_head - HQ : hq_head;
				hq_count -= n_pop;
				shm |= popm;
			}
			// render.cl:415-416: the last bounce only collects the emission. Lanes on their last bounce run the bounce below along
			// with the others and drop what it computes (the wave pays for it either way; not masking them out saves the
			// exec-mask bookkeeping and the copies of mask / direction the compiler keeps around such a branch) -- unless
			// the whole wave is on its last bounce.
			const unsigned long long lastm = ballot64(bounce == nb - 1) & shm;
			const bool show_normals = p.rd.show_normals != 0;
			if (SRT_DIAG_ON) w_shade++;
			if (in_mask(shm)) {
				SRT_REGION(SHADE_WINNER);
				// ---- winner: normal, material (render.cl:311-312,337-343,361-362,372-375); org = hit position ----
				int type, material_index;
				f3 wv;
				float ww, winv;
				uint32_t first_wtri;
				if (USE_LDS) {
					const float4 w0 = lds[2 * best], w1 = lds[2 * best + 1];
					type = (int)f2u(w0.x);
					material_index = (int)f2u(w0.y);
					wv = mk(w0.z, w0.w, w1.x);
					ww = w1.y;
					first_wtri = f2u(w1.z);
					winv = w1.w;
				} else {
					const WinnerRec *__restrict__ wr = p.winners + best;
					type = wr->type;
					material_index = wr->material;
					wv = mk(wr->vx, wr->vy, wr->vz);
					ww = wr->w;
					first_wtri = wr->first_wtri;
					winv = wr->inv_w;
				}
				const f3 pos = org;
				f3 nrm = wv; // a plane's normal as stored
				if (type == SRT_SHAPE_SPHERE) {
					nrm = div3_by_rcp(pos - wv, ww, winv);
				} else if (HAS_MODELS && type != SRT_SHAPE_PLANE) {
					SRT_REGION(SHADE_MESH_NORMAL);
					const srt_model *__restrict__ m = &p.shapes[best].shape.model;
					const float *__restrict__ w = USE_BVH ? p.bvh_blocks + (size_t)(best_tri >> 2) * 32u + (best_tri & 3u) * SRT_BVH_TRI_FLOATS
					                                      : wtris + (size_t)(first_wtri + best_tri) * SRT_WTRI_FLOATS;
					const uint32_t tri_in_model = USE_BVH ? bvh_tri_in_model(reinterpret_cast<const float4 *>(p.bvh_blocks), best_tri) : best_tri;
					f3 v0 = mk(w[0], w[1], w[2]);
					f3 e1 = mk(w[3], w[4], w[5]);
					f3 e2 = mk(w[6], w[7], w[8]);
					// barycentric_weights (render.cl:223-241), "shifted" (w2, w0, w1)
					f3 v2 = pos - v0;
					float d00 = dot3(e1, e1), d01 = dot3(e1, e2), d11 = dot3(e2, e2);
					float d20 = dot3(v2, e1), d21 = dot3(v2, e2);
					float den = d00 * d11 - d01 * d01;
					float w0 = (d11 * d20 - d01 * d21) / den;
					float w1 = (d00 * d21 - d01 * d20) / den;
					float w2 = 1.0f - w0 - w1;
					const srt_triangle *__restrict__ tr = p.triangles + (m->triangle_index + tri_in_model);
					f3 n = (ld3(tr->vertices[0].normal) * w2 + ld3(tr->vertices[1].normal) * w0) + ld3(tr->vertices[2].normal) * w1;
					n = mat_by_vec(m->transform, n, 0.0f); // forward matrix, as the reference
					nrm = normalize3(n);
				}
				const bool front = dot3(nrm, dir) < 0.0f;
				nrm = nrm * (front ? 1.0f : -1.0f);

				if (show_normals) { // (wave-uniform)
					color = mk(nrm.x * 0.5f + 0.5f, nrm.y * 0.5f + 0.5f, nrm.z * 0.5f + 0.5f); // render.cl:407-410
				} else {
					SRT_REGION(SHADE_MATERIAL);
					float4 m0, m1, mc, me;
					if (USE_LDS) {
						const float4 *__restrict__ lm = lds + 2 * n_shapes + 4 * material_index;
						m0 = lm[0], m1 = lm[1], mc = lm[2], me = lm[3];
					} else {
						const float4 *__restrict__ gm = reinterpret_cast<const float4 *>(p.materials + material_index);
						m0 = gm[0], m1 = gm[1], mc = gm[2], me = gm[3];
					}
					const float smoothness = m0.x, metallic = m0.y, specular = m0.z, emission_strength = m0.w;
					const float transmittance = m1.x, ior = m1.y;
					const f3 mcolor = mk(mc.x, mc.y, mc.z);
					color = color + (mask * mk(me.x, me.y, me.z)) * emission_strength; // render.cl:413
					if ((shm & ~lastm) != 0ull) { // (wave-uniform) somebody bounces on: see `lastm` above
						SRT_REGION(SHADE_BOUNCE);
						// cosine weighted direction: 6 draws (render.cl:421, 156-163)
						f3 rd_ = normalize3(random_normal3(seed));
						f3 hemi = rd_ * sign_fast(dot3(nrm, rd_));
						f3 random_dir = normalize3(nrm + hemi);
						f3 reflected_dir = reflect3(dir, nrm);
						// the three material draws (render.cl:427-430; nothing else draws in between)
						bool is_metallic, is_specular, is_transparent;
						if (no_specular) {
							// (wave-uniform) No material of the scene is specular: `0 > random_float` is false whatever the generator returns, so
							// the draw is not made -- the state steps over it (the compiler folds the two steps to the transmittance draw into
							// one multiply-add) -- and mix(colour, 1, 0) below is the colour (SRT_MF_PLAIN_COLORS). BASELINE configs[0..4].
							is_metallic = bernoulli(metallic, true, seed);
							seed = seed * 747796405u + 2891336453u; // random_bits' state step, its output unused
							is_specular = false;
							is_transparent = bernoulli(transmittance, true, seed);
						} else if (unit_materials) { // (wave-uniform)
							is_metallic = bernoulli(metallic, true, seed), is_specular = bernoulli(specular, true, seed), is_transparent = bernoulli(transmittance, true, seed);
						} else {
							is_metallic = bernoulli(metallic, false, seed), is_specular = bernoulli(specular, false, seed), is_transparent = bernoulli(transmittance, false, seed);
						}
						f3 rough_dir = mix3(random_dir, reflected_dir, smoothness);
						if (!is_transparent) {
							SRT_REGION(SHADE_OPAQUE);
							dir = mix3(random_dir, rough_dir, (is_metallic || is_specular) ? 1.0f : 0.0f);
							if (no_specular) mask = mask * mcolor; // = mix3(mcolor, 1, 0) for finite colours that are not -0
							else mask = mask * mix3(mcolor, mk(1.0f, 1.0f, 1.0f), is_specular ? 1.0f : 0.0f);
						} else {
							SRT_REGION(SHADE_GLASS);
							f3 in_dir = reflect3(rough_dir, nrm);
							// 1/ior and both Schlick r0 values come precomputed with the material (srt_update_scene)
							float mu = front ? m1.z : ior;
							float r0 = front ? m1.w : mc.w;
							float cos_theta = dm_min(1.0f, dot3(in_dir, neg(nrm)));
							// 1 - x with x >= 0 is 0, or at least 2^-25 in magnitude (x < 0.5: above 0.5; x >= 0.5: a multiple
							// of 2^-24), or inf / NaN: never inside sqrt_ieee's guarded interval (0, 2^-96) -- here and below
							float sin_theta = sqrt_core(1.0f - cos_theta * cos_theta);
							bool reflected = mu * sin_theta > 1.0f;
							if (!reflected) reflected = schlick(r0, cos_theta) > random_float(seed); // short-circuit ||
							if (reflected) {
								dir = rough_dir;
							} else {
								SRT_REGION(SHADE_REFRACT);
								f3 out_perp = (in_dir + nrm * cos_theta) * mu;
								float lsq = (out_perp.x * out_perp.x + out_perp.y * out_perp.y) + out_perp.z * out_perp.z;
								f3 out_parallel = nrm * (-sqrt_core(dm_fabs(1.0f - lsq)));
								dir = out_perp + out_parallel;
								mask = mask * mcolor;
							}
						}
						SRT_REGION(SHADE_TAIL);
						dir = normalize3(dir);
						org = pos + (nrm * sign_fast(dot3(nrm, dir))) * 0.001f; // render.cl:462
						bounce++;
					}
				}
			}
			// what became of the shaded paths: ended (show_normals, or the last bounce: only its emission counts) or on their way again
			if (show_normals) {
				finm |= shm;
			} else {
				finm |= lastm;
				actm = shm & ~lastm;
			}
			SRT_CLK(2);
		} else if (n_hit != 0u) {
			// PARK: every hit waits in the queue; all lanes are free for new camera rays
			if (in_mask(hitm)) {
				SRT_REGION(PARK);
				uint32_t e = hq_head + hq_count + lane_rank(hitm);
				e = e >= HQ ? e - HQ : e;
				hq[0 * HQ + e] = org.x, hq[1 * HQ + e] = org.y, hq[2 * HQ + e] = org.z;
				hq[3 * HQ + e] = dir.x, hq[4 * HQ + e] = dir.y, hq[5 * HQ + e] = dir.z;
				hq[6 * HQ + e] = mask.x, hq[7 * HQ + e] = mask.y, hq[8 * HQ + e] = mask.z;
				hq[9 * HQ + e] = color.x, hq[10 * HQ + e] = color.y, hq[11 * HQ + e] = color.z;
				hq[12 * HQ + e] = dm_u2f(seed);
				hq[13 * HQ + e] = dm_u2f((uint32_t)best);
				hq[14 * HQ + e] = dm_u2f((uint32_t)bounce);
				hq[15 * HQ + e] = dm_u2f(item);
				if (HAS_MODELS) hq[16 * HQ + e] = dm_u2f(best_tri);
			}
			asm volatile("" ::: "memory");
			hq_count += n_hit;
			SRT_CLK(3);
		}

		// ---- paths that ended in this iteration hand in their radiance ----
		if (finm != 0ull) {
			if (in_mask(finm)) {
				SRT_REGION(HANDIN);
				store_radiance(p.radiance, item, color);
			}
		}

		SRT_CLK(4);
		// ================= REFILL: free lanes take new camera rays =================
		SRT_REGION(REFILL_HEAD);
		uint32_t n_act = popc64(actm); // lanes that hold a ray
		if (SUSPEND) {
			// A scan stack that holds a wave-full is taken back by ALL 64 lanes: the scan then runs without an idle lane. Rays the
			// lanes hold at this point -- bounced or new, about to set out -- are parked and come back into lanes that fall free.
			// At the very end (no camera ray left, nothing else under way) the fuller stack is taken back as it is.
			const bool full0 = sq_count0 >= (uint32_t)SRT_SCAN_FULL, full1 = sq_count1 >= (uint32_t)SRT_SCAN_FULL;
			const bool tail = queue_dry && n_act == 0u && hq_count == 0u && pk_count == 0u;
			auto ld = [&](const float *a) { return dm_u2f(__hip_atomic_load(reinterpret_cast<const uint32_t *>(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); };
			bool took_pool = false;
			if (use_pool && tail && !full0 && !full1 && !pool_leave) {
				SRT_REGION(REFILL_POOL);
				constexpr uint32_t NF = USE_BVH ? 20u : 19u;
				const uint32_t NB = SRT_COLD(p).pool_blocks; // blocks per stack (<= SRT_POOL_BLOCKS, which the layout is made for)
				uint32_t *__restrict__ ctl = reinterpret_cast<uint32_t *>(SRT_COLD(p).scan_queue); // [0,1] records reserved, [2,3] blocks taken, [4] waves gone, [5,6] permits, [16 + stack * NB + block] records published
				float *__restrict__ prec = SRT_COLD(p).scan_queue + SRT_POOL_CTL_WORDS; // [stack][block][field][64]
				if (!pool_last && (sq_count0 | sq_count1) != 0u) {
					// (whoever ends these paths stores their radiance; an item is stored once, so there is nothing of this wave's to order it behind)
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's own stack records have arrived
					for (uint32_t sid = 0; sid < 2u; sid++) {
						const uint32_t k = sid ? sq_count1 : sq_count0;
						if (k == 0u) continue;
						uint32_t base = 0;
						if (lane == 0) base = __hip_atomic_fetch_add(ctl + sid, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
						const uint32_t room = base < NB * 64u ? NB * 64u - base : 0u;
						const uint32_t d = k < room ? k : room; // a full pool: the rays beyond stay with this wave
						const float *__restrict__ src = sq_base + sid * (20u * SQ);
						for (uint32_t r = (uint32_t)lane; r < d; r += 64u) {
							const uint32_t e = k - 1u - r, g = base + r;
							uint32_t *__restrict__ dst = reinterpret_cast<uint32_t *>(prec) + ((size_t)sid * SRT_POOL_BLOCKS + (g >> 6)) * (20u * 64u) + (g & 63u);
							for (uint32_t f = 0; f < NF; f++) __hip_atomic_store(dst + f * 64u, dm_f2u(ld(src + f * SQ + e)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						}
						// records (and the radiances above) have been acknowledged by memory before they are published: lane j adds,
						// for the j-th block the records went into, how many went there (d <= 190: four blocks at most)
						asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
						if (d != 0u) {
							const uint32_t b0 = base >> 6, bj = b0 + (uint32_t)lane;
							const uint32_t lo = bj * 64u > base ? bj * 64u : base, hi = (bj + 1u) * 64u < base + d ? (bj + 1u) * 64u : base + d;
							if ((uint32_t)lane < 4u && hi > lo) (void)__hip_atomic_fetch_add(ctl + 16u + sid * (uint32_t)SRT_POOL_BLOCKS + bj, hi - lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							// a permit for every block this reservation has completed (its last place reserved)
							const uint32_t done = ((base + d) >> 6) - b0;
							if (lane == 0 && done != 0u) (void)__hip_atomic_fetch_add(ctl + 5u + sid, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						}
						if (sid) sq_count1 = k - d;
						else sq_count0 = k - d;
						w_pool_given += d;
					}
				}
				const bool kept = (sq_count0 | sq_count1) != 0u; // (a full pool, or the last wave's own)
				uint32_t got_sid = 2u, got_blk = 0u, got_cnt = 0u;
				if (lane == 0 && !pool_last) {
					// One permit per block whose 64 places have all been reserved (ctl[5 + stack], signed); a wave that gets one
					// draws the number of its block from the head counter. No compare-and-swap loop: with thousands of waves at one
					// counter every success makes all the others fail and try again (measured: 2,100 rounds per attempt).
					for (uint32_t sid = 0; sid < 2u && got_sid == 2u; sid++) {
						if ((int)__hip_atomic_load(ctl + 5u + sid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 0) continue;
						if ((int)__hip_atomic_fetch_sub(ctl + 5u + sid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) {
							got_sid = sid, got_cnt = 64u;
							got_blk = __hip_atomic_fetch_add(ctl + 2u + sid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						} else {
							(void)__hip_atomic_fetch_add(ctl + 5u + sid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						}
					}
				}
				got_sid = (uint32_t)__builtin_amdgcn_readfirstlane((int)got_sid);
				if (got_sid == 2u && !kept && !pool_last) {
					uint32_t gone = 0;
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (what this wave published has arrived)
					if (lane == 0) gone = __hip_atomic_fetch_add(ctl + 4u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					gone = (uint32_t)__builtin_amdgcn_readfirstlane((int)gone);
					if (gone == gridDim.x - 1u) pool_last = true; // every other wave has published what it had and is gone
					else pool_leave = true;
				}
				if (got_sid == 2u && !kept && pool_last) {
					if (lane == 0) {
						for (uint32_t sid = 0; sid < 2u && got_sid == 2u; sid++) {
							const uint32_t h = __hip_atomic_load(ctl + 2u + sid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							uint32_t res = __hip_atomic_load(ctl + sid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							res = res < NB * 64u ? res : NB * 64u;
							if (h < NB && res > h * 64u) {
								got_sid = sid, got_blk = h, got_cnt = res - h * 64u < 64u ? res - h * 64u : 64u;
								__hip_atomic_store(ctl + 2u + sid, h + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							}
						}
					}
					got_sid = (uint32_t)__builtin_amdgcn_readfirstlane((int)got_sid);
				}
				if (got_sid != 2u) {
					got_blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)got_blk);
					got_cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)got_cnt);
					// All places of the block are reserved; the waves that reserved the last ones may still be writing (straight-line
					// code between their reservation and its publication: microseconds). Bounded all the same.
					uint32_t spins = 0;
					while (__hip_atomic_load(ctl + 16u + got_sid * (uint32_t)SRT_POOL_BLOCKS + got_blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < got_cnt) {
						__builtin_amdgcn_s_sleep(8);
						if (++spins > (1u << 22)) {
							if (lane == 0) atomicAdd((unsigned long long *)SRT_COLD(p).counters + SRT_CTR_WATCHDOG, 1ull);
							got_cnt = 0u; // (the launch is reported as failed)
							break;
						}
					}
					// The records are read only behind the count that publishes them: nothing lets the compiler move the (relaxed) loads
					// below in front of the loop's, but nothing forbade it either. (The hardware returns a wave's loads in order.)
					asm volatile("" ::: "memory");
					const float *__restrict__ src = prec + ((size_t)got_sid * SRT_POOL_BLOCKS + got_blk) * (20u * 64u) + (uint32_t)lane;
					actm = lanes_below(got_cnt);
					if (in_mask(actm)) {
						org = mk(ld(src + 0 * 64), ld(src + 1 * 64), ld(src + 2 * 64));
						dir = mk(ld(src + 3 * 64), ld(src + 4 * 64), ld(src + 5 * 64));
						mask = mk(ld(src + 6 * 64), ld(src + 7 * 64), ld(src + 8 * 64));
						color = mk(ld(src + 9 * 64), ld(src + 10 * 64), ld(src + 11 * 64));
						seed = dm_f2u(ld(src + 12 * 64)), bounce = (int)dm_f2u(ld(src + 13 * 64)), item = dm_f2u(ld(src + 14 * 64));
						tmin = ld(src + 15 * 64), best = (int)dm_f2u(ld(src + 16 * 64)), best_tri = dm_f2u(ld(src + 17 * 64));
						pos = dm_f2u(ld(src + 18 * 64));
					}
					asm volatile("" ::: "memory");
					resm = actm;
					n_act = got_cnt;
					took_pool = true;
					w_pool_taken++;
					if (pool_last) w_pool_last_taken++;
				}
			}
			const bool rest = tail && !took_pool && (sq_count0 | sq_count1) != 0u;
			if (took_pool) {
				// (the wave is full of rays from the pool)
			} else if ((full0 || full1 || rest) && pk_count + n_act <= PK) {
				SRT_REGION(REFILL_SCANQ);
				if (n_act != 0u) {
					const uint32_t e = pk_count + lane_rank(actm);
					if (in_mask(actm)) {
						pk[0 * PK + e] = org.x, pk[1 * PK + e] = org.y, pk[2 * PK + e] = org.z;
						pk[3 * PK + e] = dir.x, pk[4 * PK + e] = dir.y, pk[5 * PK + e] = dir.z;
						pk[6 * PK + e] = mask.x, pk[7 * PK + e] = mask.y, pk[8 * PK + e] = mask.z;
						pk[9 * PK + e] = color.x, pk[10 * PK + e] = color.y, pk[11 * PK + e] = color.z;
						pk[12 * PK + e] = dm_u2f(seed), pk[13 * PK + e] = dm_u2f((uint32_t)bounce), pk[14 * PK + e] = dm_u2f(item);
					}
					pk_count += n_act;
				}
				const uint32_t sid = full0 ? 0u : full1 ? 1u : (sq_count1 > sq_count0 ? 1u : 0u);
				const uint32_t held = sid ? sq_count1 : sq_count0;
				const uint32_t n_pop = held < 64u ? held : 64u;
				const float *__restrict__ sq = sq_base + sid * (20u * SQ);
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the records' stores have arrived (acknowledged by the L2)
				actm = lanes_below(n_pop);
				if (in_mask(actm)) {
					const uint32_t e = held - 1u - (uint32_t)lane;
					org = mk(ld(sq + 0 * SQ + e), ld(sq + 1 * SQ + e), ld(sq + 2 * SQ + e));
					dir = mk(ld(sq + 3 * SQ + e), ld(sq + 4 * SQ + e), ld(sq + 5 * SQ + e));
					mask = mk(ld(sq + 6 * SQ + e), ld(sq + 7 * SQ + e), ld(sq + 8 * SQ + e));
					color = mk(ld(sq + 9 * SQ + e), ld(sq + 10 * SQ + e), ld(sq + 11 * SQ + e));
					seed = dm_f2u(ld(sq + 12 * SQ + e)), bounce = (int)dm_f2u(ld(sq + 13 * SQ + e)), item = dm_f2u(ld(sq + 14 * SQ + e));
					tmin = ld(sq + 15 * SQ + e), best = (int)dm_f2u(ld(sq + 16 * SQ + e)), best_tri = dm_f2u(ld(sq + 17 * SQ + e));
					pos = dm_f2u(ld(sq + 18 * SQ + e));
				}
				asm volatile("" ::: "memory");
				resm = actm;
				if (sid) sq_count1 -= n_pop;
				else sq_count0 -= n_pop;
				n_act = n_pop;
			} else if (pk_count != 0u && n_act != 64u) {
				SRT_REGION(REFILL_UNPARK);
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				const uint32_t n_pop = 64u - n_act < pk_count ? 64u - n_act : pk_count;
				const unsigned long long fm = ~actm;
				const uint32_t rank = lane_rank(fm);
				const unsigned long long um = ballot64(rank < n_pop) & fm;
				if (in_mask(um)) {
					const uint32_t e = pk_count - 1u - rank;
					org = mk(ld(pk + 0 * PK + e), ld(pk + 1 * PK + e), ld(pk + 2 * PK + e));
					dir = mk(ld(pk + 3 * PK + e), ld(pk + 4 * PK + e), ld(pk + 5 * PK + e));
					mask = mk(ld(pk + 6 * PK + e), ld(pk + 7 * PK + e), ld(pk + 8 * PK + e));
					color = mk(ld(pk + 9 * PK + e), ld(pk + 10 * PK + e), ld(pk + 11 * PK + e));
					seed = dm_f2u(ld(pk + 12 * PK + e)), bounce = (int)dm_f2u(ld(pk + 13 * PK + e)), item = dm_f2u(ld(pk + 14 * PK + e));
				}
				asm volatile("" ::: "memory");
				pk_count -= n_pop;
				actm |= um;
				n_act += n_pop;
			}
		}
		const uint32_t n_free = 64u - n_act;
		if (!queue_dry && n_free >= (uint32_t)SRT_REFILL_MIN && (!SUSPEND || pk_count == 0u)) {
			const unsigned long long freem = ~actm;
			const uint32_t rank = lane_rank(freem);
			uint32_t given = 0; // free lanes served so far (wave-uniform)
			uint32_t off = 0, qpix = 0;
			while (given < n_free) {
				SRT_REGION(REFILL_LOOP);
				if (sj_next == sj_end) {
					SRT_REGION(REFILL_OPEN);
					// the current sub-job is handed out (or there is none yet): open the next one of the wave's chunk
					if (chunk_cur == chunk_end) {
						SRT_REGION(REFILL_CURSOR);
						unsigned long long start = total_items;
						if (own_chunks_end < (unsigned long long)total_items) { // else every chunk is some wave's first: nothing to ask the cursor for
							if (lane == 0) start = atomicAdd((unsigned long long *)SRT_COLD(p).queue, (unsigned long long)SRT_COLD(p).job_items);
							// lane 0's value as a scalar (wave-uniform control flow: lane 0 is active), so that everything derived from
							// it -- chunk bounds, sub-job bases -- stays in SGPRs
							const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)start);
							const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(start >> 32));
							start = (((unsigned long long)hi << 32) | lo) + own_chunks_end;
						}
						if (start >= (unsigned long long)total_items) {
							queue_dry = true;
							shade_thr = 0u;
							break;
						}
						chunk_cur = (uint32_t)start;
						const uint32_t job_items = SRT_COLD(p).job_items;
						chunk_end = (total_items - chunk_cur < job_items) ? total_items : chunk_cur + job_items;
					}
					const uint32_t left = chunk_end - chunk_cur;
					const uint32_t n = left < SUB ? left : SUB;
					const uint32_t q0 = chunk_cur / nbs; // one division per sub-job
					sj_next = chunk_cur, sj_end = chunk_cur + n, sj_qpix = q0, sj_off = chunk_cur - q0 * nbs;
					chunk_cur += n;
					continue;
				}
				SRT_REGION(REFILL_TAKE);
				const uint32_t avail = sj_end - sj_next;
				const uint32_t take = avail < n_free - given ? avail : n_free - given;
				const uint32_t r = rank - given; // (rank < given: a huge number)
				const unsigned long long takem = ballot64(r < take) & freem; // the free lanes number given .. given + take - 1
				if (in_mask(takem)) {
					item = sj_next + r;
					off = sj_off + r; // < nbs + SUB
					qpix = sj_qpix;
				}
				sj_next += take, sj_off += take;
				given += take;
				w_paths += take;
			}
			const unsigned long long gotm = ballot64(rank < given) & freem; // every lane served holds a camera ray from here on
			actm |= gotm;
			if (in_mask(gotm)) {
				SRT_REGION(CAMERA);
				// ---- camera ray (render.cl:488,496-516) ----
				// No per-lane integer division and no IEEE division sequence here: pixel, row and sample come from multiplications
				// by host-made magic numbers (srt_magic_u31), the two quotients by the image size from div_by_rcp.
				const auto &c = SRT_COLD(p);
#ifndef SRT_NO_CAMERA_DIET
				// off < nbs + SUB: one pixel further at most when a pixel has at least SUB samples in this batch, else off / nbs by a
				// 16-bit reciprocal (exact below 256 for divisors below 128: the error off * (magic * nbs - 2^16) stays under 2^15)
				static_assert(SUB <= 128u, "off / nbs by the 16-bit reciprocal is exact for off < 256 and nbs < 128 only (tests/csrc/magic_check.cpp)");
				const uint32_t dq = (nbs >= SUB) ? (off >= nbs ? 1u : 0u) : (off * c.nbs_magic16) >> 16;
				const uint32_t q = qpix + dq; // owned pixels < 2^31 (checked by the host)
				const uint32_t sample = c.first_sample + (off - dq * nbs);
				const uint32_t lrow = (__umulhi(q, c.width_magic) + q) >> c.width_shift;
				const int px = (int)(q - lrow * (uint32_t)width);
				int py = (int)lrow;
				if (c.world != 1) { // (wave-uniform) global y of packed local row (include/srt_abi.h srt_set_partition)
					const uint32_t lb = (__umulhi(lrow, c.rpb_magic) + lrow) >> c.rpb_shift;
					py = (int)((lb * (uint32_t)c.world + (uint32_t)c.rank) * (uint32_t)c.rows_per_block + (lrow - lb * (uint32_t)c.rows_per_block));
				}
				const uint32_t id = (uint32_t)px + (uint32_t)py * (uint32_t)width;
				seed = (sample + id * (uint32_t)ns) * c.rd.time * 5304u;
				float ndc_x = div_by_rcp((float)px + random_float(seed), c.f_width, c.inv_f_width);
				float ndc_y = div_by_rcp((float)py + random_float(seed), c.f_height, c.inv_f_height);
#else
				const uint32_t dq = (nbs >= SUB) ? (off >= nbs ? 1u : 0u) : off / nbs;
				const uint32_t q = qpix + dq; // owned pixels < 2^31 (checked by the host)
				const uint32_t sample = c.first_sample + (off - dq * nbs);
				const uint32_t lrow = q / (uint32_t)width;
				const int px = (int)(q - lrow * (uint32_t)width);
				const int py = c.world == 1 ? (int)lrow : global_row((int)lrow, c.rank, c.world, c.rows_per_block); // (wave-uniform choice)
				const uint32_t id = (uint32_t)px + (uint32_t)py * (uint32_t)width;
				seed = (sample + id * (uint32_t)ns) * c.rd.time * 5304u;
				float ndc_x = ((float)px + random_float(seed)) / c.f_width;
				float ndc_y = ((float)py + random_float(seed)) / c.f_height;
#endif
				float sx = ((2.f * ndc_x - 1.f) * c.rd.aspect_ratio) * c.rd.fov_scale;
				float sy = (1.f - 2.f * ndc_y) * c.rd.fov_scale;
				const f3 c0 = mk(c.rd.camera_to_world[0].x, c.rd.camera_to_world[0].y, c.rd.camera_to_world[0].z);
				const f3 c1 = mk(c.rd.camera_to_world[1].x, c.rd.camera_to_world[1].y, c.rd.camera_to_world[1].z);
				const f3 c2 = mk(c.rd.camera_to_world[2].x, c.rd.camera_to_world[2].y, c.rd.camera_to_world[2].z);
				org = mk(c.rd.camera_to_world[3].x, c.rd.camera_to_world[3].y, c.rd.camera_to_world[3].z);
				dir = normalize3(mat_cols_by_vec(c0, c1, c2, org, mk(sx, sy, -1.0f), 0.0f)); // (the position column times w = 0 stays: inf * 0 is a NaN the reference has, too)
				mask = mk(1.f, 1.f, 1.f);
				color = mk(0.f, 0.f, 0.f);
				bounce = 0;
			}
		}

		SRT_CLK(5);
		SRT_REGION(LOOP_TAIL);
		if (queue_dry) { // (wave-uniform; while the cursor has work a wave always holds some)
			if (actm == 0ull && hq_count == 0u && (sq_count0 | sq_count1 | pk_count) == 0u) {
				if (!use_pool || pool_leave || pool_last) break; // (with a pool: not before the wave has signed off, REFILL above)
				// Bounded: a wave that spins here without ever getting work leaves with the watchdog counter set instead of hanging.
				if (++idle_spins > (1u << 20)) {
					if (lane == 0) atomicAdd((unsigned long long *)SRT_COLD(p).counters + SRT_CTR_WATCHDOG, 1ull);
					break;
				}
			} else {
				idle_spins = 0;
			}
		}
	}
	// queue dry, no lane holds a ray, nothing parked: the escapes still in the ring are what is left
	SRT_REGION(EPILOGUE);
	if (ring_count != 0u) resolve_ring(p, ring, ring_count, lane SRT_RC_ARG);

	// per-wave counters: this wave's own 64-byte line, no atomics (device_types.h)
	unsigned long long t3 = COUNT_TRIS ? n_tri : w_scans, t4 = COUNT_TRIS ? n_tri_u : w_scan_lanes;
	if (COUNT_TRIS || SUSPEND) {
		for (int off = 32; off > 0; off >>= 1) {
			t3 += __shfl_down(t3, off);
			t4 += __shfl_down(t4, off);
		}
	}
	if (lane == 0) {
		unsigned long long *__restrict__ w = (unsigned long long *)SRT_COLD(p).wave_counters + (size_t)blockIdx.x * SRT_WAVE_CTR_STRIDE;
		w[0] += w_rays;
		w[1] += w_sky;
		w[2] += w_paths;
		if (COUNT_TRIS) {
			w[3] += t3;
			w[4] += t4;
		} else if (SUSPEND) { // diagnostics in the slots the instrumented variant uses for triangle counts
			w[8] += t3;
			w[9] += t4;
#ifndef SRT_PHASE_CLOCK
			w[10] += w_pool_taken, w[11] += w_pool_given, w[12] += (unsigned long long)w_pool_taken * w_pool_taken, w[13] += w_pool_last_taken;
#endif
		}
		w[6] += w_iter;
		w[7] += w_shade;
#ifdef SRT_PHASE_CLOCK
		for (int i = 0; i < 7; i++) w[8 + i] += clk_t[i];
		w[15] += __builtin_amdgcn_s_memtime() - clk_start;
#endif
	}
#ifdef SRT_REGION_COUNT
	__syncthreads();
	{
		unsigned long long *__restrict__ w = (unsigned long long *)SRT_COLD(p).wave_counters + (size_t)blockIdx.x * SRT_WAVE_CTR_STRIDE + 16;
		for (int i = lane; i < 2 * SRT_REGION_MAX; i += 64) w[i] += region_ctr[i];
	}
#endif
}

namespace {
__device__ __forceinline__ float aces1(float x) {
	const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
	return dm_clamp((x * (x * a + b)) / (x * (x * c + d) + e), 0.0f, 1.0f);
}
__device__ __forceinline__ uint32_t to_uchar(float v) { return (v == v) ? ((uint32_t)(int)v & 255u) : 0u; }
} // namespace

// ---------------------------------------------------------------------------------
// Ordered reduction: lane = pixel, serial over the batch's samples in sample order, so
// the float sums are the reference's `color += trace(...)` sequence bit for bit no matter
// which wave traced which sample. 12 B per path in, 16 B RMW per pixel out: HBM-bound.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void srt_reduce_kernel(const ReduceParams p) {
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q == 0u && p.queue_reset) *p.queue_reset = 0ull; // the trace launch this reduction follows is over: its cursor, ready for the next one
	if (q >= p.num_pixels) return;
	f3 c = mk(0.f, 0.f, 0.f);
	float4 *run = reinterpret_cast<float4 *>(p.running) + q;
	if (!p.first_batch) {
		float4 v = *run;
		c = mk(v.x, v.y, v.z);
	}
	const uint32_t n = p.batch_samples;
	const float *__restrict__ r = p.radiance + (size_t)q * n * 3u;
	uint32_t k = 0;
	if ((n & 3u) == 0u) {
		// 4 samples = 48 B = three aligned 16-byte loads; additions stay in sample order
		const float4 *__restrict__ r4 = reinterpret_cast<const float4 *>(r);
		// SRT_REDUCE_DEPTH x 4 samples per trip: every lane streams its own 12 KB run, so HBM efficiency
		// comes from bytes in flight per lane. Measured at config 2 (25.5 GB): 3 loads per trip 7.16 ms
		// (3.6 TB/s), 12 loads 4.75 ms (5.4 TB/s). The additions stay in sample order.
#ifndef SRT_REDUCE_DEPTH
#define SRT_REDUCE_DEPTH 4
#endif
		for (; k + 4 * SRT_REDUCE_DEPTH <= n; k += 4 * SRT_REDUCE_DEPTH) {
			float4 v[3 * SRT_REDUCE_DEPTH];
#pragma unroll
			for (int i = 0; i < 3 * SRT_REDUCE_DEPTH; i++) v[i] = r4[i];
			r4 += 3 * SRT_REDUCE_DEPTH;
#pragma unroll
			for (int i = 0; i < 3 * SRT_REDUCE_DEPTH; i += 3) {
				const float4 a = v[i], b = v[i + 1], d = v[i + 2];
				c = c + mk(a.x, a.y, a.z);
				c = c + mk(a.w, b.x, b.y);
				c = c + mk(b.z, b.w, d.x);
				c = c + mk(d.y, d.z, d.w);
			}
		}
		for (; k < n; k += 4) {
			const float4 a = r4[0], b = r4[1], d = r4[2];
			r4 += 3;
			c = c + mk(a.x, a.y, a.z);
			c = c + mk(a.w, b.x, b.y);
			c = c + mk(b.z, b.w, d.x);
			c = c + mk(d.y, d.z, d.w);
		}
	} else {
		for (; k < n; k++) c = c + mk(r[3 * k], r[3 * k + 1], r[3 * k + 2]);
	}
	if (p.last_batch) {
		c = c / (float)p.num_samples; // render.cl:520 (num_samples == 0 -> 0/0 = NaN, as the reference)
		float4 *out = reinterpret_cast<float4 *>(p.canvas) + q;
		float4 o = *out;
		o.x += c.x;
		o.y += c.y;
		o.z += c.z;
		*out = o; // render.cl:522
		if (c.x != c.x || c.y != c.y || c.z != c.z) atomicAdd(&p.counters[SRT_CTR_NAN], 1ull);
		if (p.argb) { // the resolve of this pixel (srt_resolve_kernel's expressions on the value just written), fused for srt_render
			const float n = (float)p.num_steps;
			const float r = sqrt_ieee(aces1(o.x / n)), g = sqrt_ieee(aces1(o.y / n)), b = sqrt_ieee(aces1(o.z / n));
			reinterpret_cast<uint32_t *>(p.argb)[q] = 255u | (to_uchar(r * 255.0f) << 8) | (to_uchar(g * 255.0f) << 16) | (to_uchar(b * 255.0f) << 24);
		}
	} else {
		*run = make_float4(c.x, c.y, c.z, 0.f);
	}
}

// ---------------------------------------------------------------------------------
// Pre-pass: world-space triangles per model instance. blockIdx.y = shape.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void srt_prepass_kernel(const PrepassParams p) {
	const int si = blockIdx.y;
	if (si >= p.num_shapes) return;
	const srt_shape *sh = p.shapes + si;
	if (sh->type != SRT_SHAPE_MODEL) return;
	const srt_model *m = &sh->shape.model;
	const uint32_t n = m->num_triangles;
	const uint32_t base = p.wtri_offset[si];
	for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
		const uint32_t j = p.order ? p.order[base + s] : s; // BVH layout: record s of the model holds its triangle order[s]
		const srt_triangle *t = p.triangles + (m->triangle_index + j);
		// render.cl:325-328 then :247-248
		f3 p0 = mat_by_vec(m->transform, ld3(t->vertices[0].pos), 1.0f);
		f3 p1 = mat_by_vec(m->transform, ld3(t->vertices[1].pos), 1.0f);
		f3 p2 = mat_by_vec(m->transform, ld3(t->vertices[2].pos), 1.0f);
		f3 e1 = p1 - p0, e2 = p2 - p0;
		// BVH layout: record base + s lives in slot (dest & 3) of leaf block (dest >> 2)
		float *w = p.order ? p.wtris + (size_t)(p.dest[base + s] >> 2) * 32u + (p.dest[base + s] & 3u) * SRT_BVH_TRI_FLOATS
		                   : p.wtris + (size_t)(base + s) * SRT_WTRI_FLOATS;
		w[0] = p0.x, w[1] = p0.y, w[2] = p0.z;
		w[3] = e1.x, w[4] = e1.y, w[5] = e1.z;
		w[6] = e2.x, w[7] = e2.y, w[8] = e2.z;
		if (p.order) p.wtris[(size_t)(p.dest[base + s] >> 2) * 32u + SRT_BVH_LEAF_J + (p.dest[base + s] & 3u)] = dm_u2f(j);
	}
}

// ---------------------------------------------------------------------------------
// Resolve: canvas / num_steps -> ACES -> sqrt -> A,R,G,B bytes (render.cl:473-481,525-535)
// 16 B in, 4 B out per pixel; HBM-bound.
// ---------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void srt_resolve_kernel(const ResolveParams p) {
	const float4 *__restrict__ canvas = reinterpret_cast<const float4 *>(p.canvas);
	uint32_t *__restrict__ out = reinterpret_cast<uint32_t *>(p.argb);
	const float n = (float)p.num_steps;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < p.num_pixels; i += gridDim.x * blockDim.x) {
		float4 c = canvas[i];
		float r = sqrt_ieee(aces1(c.x / n));
		float g = sqrt_ieee(aces1(c.y / n));
		float b = sqrt_ieee(aces1(c.z / n));
		// memory order A, R, G, B (little endian word)
		out[i] = 255u | (to_uchar(r * 255.0f) << 8) | (to_uchar(g * 255.0f) << 16) | (to_uchar(b * 255.0f) << 24);
	}
}

// ---------------------------------------------------------------------------------
// Math self-test: the kernel-local specialisations against their generic definitions,
// and checksums of the generic ones for comparison with the HOST build of detmath.h.
// r walks 0, stride, 2*stride, ... over all 2^32 values; u = r * 2^-32 is exactly what
// random_float returns for that r.
//   out[0] sqrt_ieee(bits r) != __builtin_sqrtf   out[1] log_unit(u) != dm_logf(u)
//   out[2] cos_2pi(t) != dm_cosf(t), t = 2pi*u    out[3] sum of bits of dm_logf(u)
//   out[4] sum of bits of dm_cosf(t)              out[5] sum of bits of sqrt(u)
//   out[6] sum of bits of dm_atan2pif(u - 0.5, 0.37 - u)   out[7] sum of bits of dm_powf(u, 25)
//   out[8] div3(a, b) != a / b, or div3_by_rcp(a, b, RN(1 / b) or 0) != a / b          out[9] sum of bits of normalize(u - 0.5, 0.37 - u, (r & 0xffff) * 1e-3 - 30)
//          (a, b: random mantissas and signs, exponents straddling the fast paths' guards,
//           zero components mixed in)
//   out[10] Box-Muller's rho as random_normal computes it (sqrt_rsq_zero_ok of -2 log of the raw count, u = 0 selected to +inf)
//           != IEEE sqrt(-2 dm_logf(u))
//   out[11] the 2^-32 scaling folded away: log_count(r) != log_unit(u), or K' * r != 6.28318548f * u
//   out[12] sqrt_rsq(bits r) != __builtin_sqrtf for r a float in [2^-96, +inf) (every one of them at stride 1)
//   out[13] div_by_rcp((px + u), W, 1 / W) != (px + u) / W over eight image sizes W (the host's 1 / W passed in)
//   out[14] sign_fast(bits r) != dm_sign
//   out[15] powi_uniform(x, n) != dm_powi(x, n) for x = u, -u and the float with r's bits, n = 1 + (block % 32)
// ---------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ bool same_float(float a, float b) { return (a != a && b != b) || dm_f2u(a) == dm_f2u(b); }
__device__ __forceinline__ unsigned long long canon_bits(float a) { return (a != a) ? 0x7fc00000ull : (unsigned long long)dm_f2u(a); }
__device__ __forceinline__ uint32_t mix32(uint32_t &h) {
	h = h * 747796405u + 2891336453u;
	uint32_t r = ((h >> ((h >> 28) + 4u)) ^ h) * 277803737u;
	return (r >> 22) ^ r;
}
// random sign and mantissa, biased exponent uniform in [lo, lo + span)
__device__ __forceinline__ float rand_float_exp(uint32_t &h, uint32_t lo, uint32_t span) {
	const uint32_t m = mix32(h), e = lo + mix32(h) % span;
	return dm_u2f((m & 0x807fffffu) | (e << 23));
}
__device__ __forceinline__ bool same_f3(f3 a, f3 b) { return same_float(a.x, b.x) && same_float(a.y, b.y) && same_float(a.z, b.z); }
} // namespace

struct SelftestSizes {
	float w[8], inv_w[8]; // image sizes and their reciprocals as the HOST rounds them
};
__global__ __launch_bounds__(256) void srt_selftest_kernel(unsigned long long *out, uint32_t stride, const SelftestSizes sz) {
	unsigned long long bad_sqrt = 0, bad_log = 0, bad_cos = 0, s_log = 0, s_cos = 0, s_sqrt = 0, s_atan = 0, s_pow = 0;
	unsigned long long bad_div = 0, bad_norm = 0, bad_rn = 0, bad_fold = 0, bad_rsq = 0, bad_cam = 0, bad_sign = 0, bad_powi = 0;
	const int pw_n = 1 + (int)(blockIdx.x & 31u); // (uniform per workgroup, as powi_uniform requires)
	const unsigned long long total = (0x100000000ull + stride - 1) / stride;
	for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < total;
	     i += (unsigned long long)gridDim.x * blockDim.x) {
		const uint32_t r = (uint32_t)(i * stride);
		const float asbits = dm_u2f(r);
		bad_sqrt += same_float(sqrt_ieee(asbits), __builtin_sqrtf(asbits)) ? 0 : 1;
		const float u = (float)r * 2.3283064365386963e-10f;
		const float lg = dm_logf(u);
		bad_log += same_float(log_unit(u), lg) ? 0 : 1;
		const float th = 6.28318548f * u;
		const float cs = dm_cosf(th);
		bad_cos += same_float(cos_2pi(th), cs) ? 0 : 1;
		s_log += canon_bits(lg);
		s_cos += canon_bits(cs);
		s_sqrt += canon_bits(dm_sqrtf(u));
		s_atan += canon_bits(dm_atan2pif(u - 0.5f, 0.37f - u));
		s_pow += canon_bits(dm_powf(u, 25.0f));
		const float cnt = (float)r;
		bad_fold += (same_float(log_count(cnt), log_unit(u)) && same_float((6.28318548f * 2.3283064365386963e-10f) * cnt, th)) ? 0 : 1;
		const float arg = -2.0f * lg;
#if !defined(SRT_NO_COUNT_FOLD) && !defined(SRT_NO_FAST_SQRT) && !defined(SRT_NO_RSQ_SQRT)
		{
			float rho = sqrt_rsq_zero_ok(-2.0f * log_unit_biased<159, false>(cnt));
			rho = cnt == 0.0f ? DM_INF_F : rho;
			bad_rn += same_float(rho, __builtin_sqrtf(arg)) ? 0 : 1;
		}
#else
		bad_rn += same_float(sqrt_core(arg), __builtin_sqrtf(arg)) ? 0 : 1;
#endif
		bad_sign += same_float(sign_fast(asbits), dm_sign(asbits)) ? 0 : 1;
		bad_powi += (same_float(powi_uniform(u, pw_n), dm_powi(u, pw_n)) && same_float(powi_uniform(-u, pw_n), dm_powi(-u, pw_n)) &&
		             same_float(powi_uniform(asbits, pw_n), dm_powi(asbits, pw_n)))
		                ? 0
		                : 1;
		if (r >= 0x0f800000u && r < 0x7f800000u) bad_rsq += same_float(sqrt_rsq(asbits), __builtin_sqrtf(asbits)) ? 0 : 1;
		{
			const float W = sz.w[r & 7u];
			const float a = (float)((r >> 3) % (uint32_t)W) + u;
			bad_cam += same_float(div_by_rcp(a, W, sz.inv_w[r & 7u]), a / W) ? 0 : 1;
		}
		// guards: numerators 2^-60 .. 2^50, denominator 2^-40 .. 2^40, squared length 2^-80 .. 2^80
		uint32_t h = r ^ 0x9e3779b9u;
		f3 a = mk(rand_float_exp(h, 127 - 64, 118), rand_float_exp(h, 127 - 64, 118), rand_float_exp(h, 127 - 64, 118));
		if ((r & 15u) == 3u) a.x = 0.0f;
		if ((r & 31u) == 5u) a.y = -0.0f;
		if ((r & 0xfffu) == 7u) a.z = dm_u2f(mix32(h)); // any bit pattern: denormals, inf, NaN
		float b = rand_float_exp(h, 127 - 44, 88);
		if ((r & 0xffffu) == 11u) b = dm_u2f(0x7fc00000u | (mix32(h) & 0x3fffffu)); // a NaN denominator now and then
		bad_div += same_f3(div3(a, b), a / b) ? 0 : 1;
		{ // the sphere normal's form: the host's correctly rounded 1 / b for b in [2^-40, 2^40], else 0 (WinnerRec.inv_w, srt_abi.hip)
			const float ab = dm_fabs(b);
			const float y = (ab >= 0x1p-40f && ab <= 0x1p40f) ? 1.0f / b : 0.0f;
			bad_div += same_f3(div3_by_rcp(a, b, y), a / b) ? 0 : 1;
		}
		// the built-in normalize on a vector made from r with plain float operations: checksum against the host build
		const f3 nv = normalize3(mk(u - 0.5f, 0.37f - u, (float)(r & 0xffffu) * 1e-3f - 30.0f));
		bad_norm += canon_bits(nv.x) + canon_bits(nv.y) + canon_bits(nv.z);
	}
	atomicAdd(&out[0], bad_sqrt);
	atomicAdd(&out[1], bad_log);
	atomicAdd(&out[2], bad_cos);
	atomicAdd(&out[3], s_log);
	atomicAdd(&out[4], s_cos);
	atomicAdd(&out[5], s_sqrt);
	atomicAdd(&out[6], s_atan);
	atomicAdd(&out[7], s_pow);
	atomicAdd(&out[8], bad_div);
	atomicAdd(&out[9], bad_norm);
	atomicAdd(&out[10], bad_rn);
	atomicAdd(&out[11], bad_fold);
	atomicAdd(&out[12], bad_rsq);
	atomicAdd(&out[13], bad_cam);
	atomicAdd(&out[14], bad_sign);
	atomicAdd(&out[15], bad_powi);
}

void srt_launch_selftest(unsigned long long *out, uint32_t stride, void *stream) {
	SelftestSizes sz;
	const float w[8] = {1920.f, 1080.f, 256.f, 3840.f, 2160.f, 960.f, 37.f, 16777216.f};
	for (int i = 0; i < 8; i++) sz.w[i] = w[i], sz.inv_w[i] = 1.0f / w[i];
	hipLaunchKernelGGL(srt_selftest_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, out, stride, sz);
}

// ---------------------------------------------------------------------------------
// launch wrappers (host)
// ---------------------------------------------------------------------------------
int srt_trace_waves_per_simd(int has_models, int use_bvh) {
	return !has_models ? SRT_TRACE_WAVES_PER_SIMD : use_bvh ? SRT_TRACE_WAVES_PER_SIMD_BVH : SRT_TRACE_WAVES_PER_SIMD_MODELS;
}
int srt_scan_suspend_min(void) { return SRT_SCAN_SUSPEND_MIN; }
int srt_scan_queue_in_hbm(void) { return 1; }
int srt_bvh_suspends(void) { return SRT_BVH_SUSPEND; }
int srt_sub_job_items(int has_models, int use_bvh) { return !has_models ? SRT_SUB_PLAIN : use_bvh ? SRT_SUB_BVH : SRT_SUB_MODELS; }

int srt_trace_lds_floats(int has_models, int use_bvh) {
	// the sky ring (10 fields), the hit queue (16..17 fields)
	int n = 10 * SRT_RING_CAP + (has_models ? (use_bvh ? 17 * SRT_HQ_CAP_BVH : 17 * SRT_HQ_CAP_MODELS) : 16 * SRT_HQ_CAP);
#ifdef SRT_REGION_COUNT
	n += 2 * SRT_REGION_MAX; // (waves, lanes) per region
#endif
	return n;
}

namespace {
typedef void (*TraceKernel)(const TraceParams);
TraceKernel pick_trace_kernel(bool models, bool use_bvh, bool use_lds, bool count_triangles) {
	if (!models) return use_lds ? srt_trace_kernel<false, true, false, false> : srt_trace_kernel<false, false, false, false>;
	if (use_bvh) {
		if (use_lds) return count_triangles ? srt_trace_kernel<true, true, true, true> : srt_trace_kernel<false, true, true, true>;
		return count_triangles ? srt_trace_kernel<true, false, true, true> : srt_trace_kernel<false, false, true, true>;
	}
	if (use_lds) return count_triangles ? srt_trace_kernel<true, true, true, false> : srt_trace_kernel<false, true, true, false>;
	return count_triangles ? srt_trace_kernel<true, false, true, false> : srt_trace_kernel<false, false, true, false>;
}
// winners + materials go to LDS when small enough not to cost occupancy
size_t scene_lds_bytes(const TraceParams &p) {
	size_t scene = (size_t)p.sd.num_shapes * sizeof(WinnerRec) + (size_t)p.num_materials * sizeof(srt_material);
	if (p.num_models == 0) scene += (size_t)p.num_runs * (sizeof(BlockGroup) + 192); // sphere / plane scenes: group headers + shape blocks
	return scene <= 4608 ? scene : 0; // all record types are multiples of 16 B
}
} // namespace

// Persistent waves (= one-wave workgroups) of this launch configuration that one CU holds at once, as the runtime
// computes it from the kernel's registers and its dynamic LDS; the grid must not exceed CUs x this, or the surplus
// waves would only start -- each with a first chunk of its own -- when others have drained the queue.
int srt_trace_resident_waves_per_cu(const TraceParams &p, bool count_triangles) {
	const size_t scene_lds = scene_lds_bytes(p);
	const size_t need = scene_lds + (size_t)srt_trace_lds_floats(p.num_models > 0, p.use_bvh) * sizeof(float);
	int blocks = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, pick_trace_kernel(p.num_models > 0, p.use_bvh != 0, scene_lds != 0, count_triangles), 64, need) != hipSuccess ||
	    blocks <= 0) {
		(void)hipGetLastError();
		blocks = 4 * srt_trace_waves_per_simd(p.num_models > 0, p.use_bvh);
	}
	return blocks;
}

void srt_launch_trace(TraceParams p, bool count_triangles, int num_waves, void *stream) {
	if (p.total_items == 0 || num_waves <= 0) return;
	dim3 grid((unsigned)num_waves), block(64);
	const size_t scene_lds = scene_lds_bytes(p);
	p.lds_bytes = (uint32_t)scene_lds;
	p.stage_off = (uint32_t)(scene_lds / sizeof(float4));
	const size_t need = scene_lds + (size_t)srt_trace_lds_floats(p.num_models > 0, p.use_bvh) * sizeof(float);
	hipLaunchKernelGGL(pick_trace_kernel(p.num_models > 0, p.use_bvh != 0, scene_lds != 0, count_triangles), grid, block, need, (hipStream_t)stream, p);
}

void srt_launch_reduce(const ReduceParams &p, void *stream) {
	if (p.num_pixels == 0) return;
	hipLaunchKernelGGL(srt_reduce_kernel, dim3((p.num_pixels + 255) / 256), dim3(256), 0, (hipStream_t)stream, p);
}

void srt_launch_prepass(const PrepassParams &p, uint64_t max_tris_per_model, void *stream) {
	if (p.num_shapes <= 0 || max_tris_per_model == 0) return;
	unsigned gx = (unsigned)((max_tris_per_model + 255) / 256);
	if (gx > 4096) gx = 4096;
	dim3 grid(gx, (unsigned)p.num_shapes), block(256);
	hipLaunchKernelGGL(srt_prepass_kernel, grid, block, 0, (hipStream_t)stream, p);
}

void srt_launch_resolve(const ResolveParams &p, void *stream) {
	if (p.num_pixels == 0) return;
	unsigned gx = (p.num_pixels + 255) / 256;
	if (gx > 4096) gx = 4096;
	hipLaunchKernelGGL(srt_resolve_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, p);
}
