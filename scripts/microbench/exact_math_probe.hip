// exact_math_probe.hip — which cheaper instruction sequences still give the IEEE result on gfx950?
// Exhaustive (2^32 inputs) mismatch counts of candidate sqrt / division / cosine-sign sequences against the
// compiler's correctly rounded ones, on exactly the domains the trace kernel feeds them. A candidate is used in
// csrc/kernels.hip only when its count here is 0 (and srt_selftest_math then re-checks it inside the library).
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I simple-raytracer_amd/csrc scripts/microbench/exact_math_probe.hip -o scripts/microbench/bin/exact_math_probe && scripts/microbench/bin/exact_math_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#include "detmath.h"

__device__ __forceinline__ bool same(float a, float b) { return (a != a && b != b) || dm_f2u(a) == dm_f2u(b); }
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
// ---- sqrt candidates ----
__device__ __forceinline__ float sqrt_s1(float x) { // rsq + one Markstein step: 1 trans + 4
	float y = __builtin_amdgcn_rsqf(x);
	float s = x * y, h = 0.5f * y;
	float r = __builtin_fmaf(-s, s, x);
	return __builtin_fmaf(r, h, s);
}
__device__ __forceinline__ float sqrt_s2(float x) { // rsq + Goldschmidt step + Markstein step: 1 trans + 7
	float y = __builtin_amdgcn_rsqf(x);
	float s = x * y, h = 0.5f * y;
	float e = __builtin_fmaf(-s, h, 0.5f);
	s = __builtin_fmaf(s, e, s);
	h = __builtin_fmaf(h, e, h);
	float r = __builtin_fmaf(-s, s, x);
	return __builtin_fmaf(r, h, s);
}
__device__ __forceinline__ float sqrt_s3(float x) { // v_sqrt (1 ulp) + rsq for the correction: 2 trans + 3
	float s = __builtin_amdgcn_sqrtf(x);
	float h = 0.5f * __builtin_amdgcn_rsqf(x);
	float r = __builtin_fmaf(-s, s, x);
	return __builtin_fmaf(r, h, s);
}
__device__ __forceinline__ float sqrt_s4(float x) { // rsq + Markstein twice: 1 trans + 6
	float y = __builtin_amdgcn_rsqf(x);
	float s = x * y, h = 0.5f * y;
	float r = __builtin_fmaf(-s, s, x);
	s = __builtin_fmaf(r, h, s);
	r = __builtin_fmaf(-s, s, x);
	return __builtin_fmaf(r, h, s);
}
__device__ __forceinline__ float sqrt_s5(float x) { // S1 with the rsq clamped (keeps -0 -> -0): 1 trans + 5
	float y = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rsqf(x), 0.0f, 0x1p100f);
	float s = x * y, h = 0.5f * y;
	float r = __builtin_fmaf(-s, s, x);
	return __builtin_fmaf(r, h, s);
}
// ---- the logarithm's f / (2 + f) ----
__device__ __forceinline__ float div_d1(float a, float b) { // raw rcp + one correction: 1 trans + 3
	float r = __builtin_amdgcn_rcpf(b);
	float q = a * r;
	float m = __builtin_fmaf(-b, q, a);
	return __builtin_fmaf(m, r, q);
}
__device__ __forceinline__ float div_d2(float a, float b) { // refined rcp + one correction: 1 trans + 5
	float r = rcp_refined(b);
	float q = a * r;
	float m = __builtin_fmaf(-b, q, a);
	return __builtin_fmaf(m, r, q);
}
__device__ __forceinline__ float div_d4(float a, float b) { // raw rcp + two corrections: 1 trans + 5
	float r = __builtin_amdgcn_rcpf(b);
	return div_core(a, b, r);
}
__device__ __forceinline__ float div_fix(float a, float b) { // refined rcp, two corrections, v_div_fixup, no scaling: 1 trans + 8
	float r = rcp_refined(b);
	return __builtin_amdgcn_div_fixupf(div_core(a, b, r), b, a);
}
__device__ __forceinline__ float div_fix1(float a, float b) { // refined rcp, ONE correction, v_div_fixup: 1 trans + 6
	float r = rcp_refined(b);
	float q = a * r;
	float m = __builtin_fmaf(-b, q, a);
	return __builtin_amdgcn_div_fixupf(__builtin_fmaf(m, r, q), b, a);
}
__device__ __forceinline__ uint32_t mix32(uint32_t &h) {
	h = h * 747796405u + 2891336453u;
	uint32_t r = ((h >> ((h >> 28) + 4u)) ^ h) * 277803737u;
	return (r >> 22) ^ r;
}
__device__ __forceinline__ float rand_float_exp(uint32_t &h, uint32_t lo, uint32_t span) {
	const uint32_t m = mix32(h), e = lo + mix32(h) % span;
	return dm_u2f((m & 0x807fffffu) | (e << 23));
}
// cos_2pi of the kernel and a variant that flips the sign with integer ops
__device__ __forceinline__ float cos_cur(float x, bool alt) {
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
	if (!alt) return (((k + 1) >> 1) & 1) ? -res : res;
	return dm_u2f(dm_f2u(res) ^ ((((uint32_t)k << 30) + 0x40000000u) & 0x80000000u));
}

enum { SQ_S1, SQ_S2, SQ_S3, SQ_S4, SQ_S1_BM, SQ_S5_BM, SQ_S3_BM, LOG_D1, LOG_D2, LOG_D4, GEN_D2, GEN_FIX_BOX, GEN_FIX_WIDE, GEN_FIX1_WIDE, CAM_HOSTRCP, COS_ALT, SQ_S1_GLASS, N_OUT };

__global__ __launch_bounds__(256) void probe(unsigned long long *out) {
	unsigned long long bad[N_OUT] = {0};
	const float widths[8] = {1920.f, 1080.f, 256.f, 3840.f, 2160.f, 960.f, 37.f, 41.f};
	for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < 0x100000000ull; i += (unsigned long long)gridDim.x * blockDim.x) {
		const uint32_t r = (uint32_t)i;
		// (1) sqrt on every bit pattern that is +-0, NaN, inf, negative or of magnitude >= 2^-96 (what sqrt_core is handed)
		const float x = dm_u2f(r);
		const uint32_t mag = r & 0x7fffffffu;
		if (!((mag - 1u) < 0x0f7fffffu)) {
			const float want = __builtin_sqrtf(x);
			const bool normal_pos = (r >> 31) == 0 && mag >= 0x0f800000u && mag < 0x7f800000u; // the range a shortcut has to get right by itself
			if (normal_pos) {
				bad[SQ_S1] += same(sqrt_s1(x), want) ? 0 : 1;
				bad[SQ_S2] += same(sqrt_s2(x), want) ? 0 : 1;
				bad[SQ_S3] += same(sqrt_s3(x), want) ? 0 : 1;
				bad[SQ_S4] += same(sqrt_s4(x), want) ? 0 : 1;
			}
		}
		// (2) Box-Muller: rho = sqrt(-2 log u), every u the RNG can return
		const float u = (float)r * 2.3283064365386963e-10f;
		const float arg = -2.0f * dm_logf(u);
		const float rho = __builtin_sqrtf(arg);
		bad[SQ_S1_BM] += same(sqrt_s1(arg), rho) ? 0 : 1; // fails at u = 0 (inf) and u = 1 (-0) at least
		bad[SQ_S5_BM] += (u == 0.0f || same(sqrt_s5(arg), rho)) ? 0 : 1; // u = 0 is selected away by the caller
		bad[SQ_S3_BM] += same(sqrt_s3(arg), rho) ? 0 : 1;
		// (3) the logarithm's division on every u (f from the mantissa normalisation of dm_logf)
		{
			uint32_t ix = dm_f2u(u) & 0x007fffffu;
			const uint32_t ii = (ix + 0x4afb20u) & 0x00800000u;
			const float xm = dm_u2f(ix | (ii ^ 0x3f800000u));
			const float f = xm - 1.0f, den = 2.0f + f;
			const float want = f / den;
			bad[LOG_D1] += same(div_d1(f, den), want) ? 0 : 1;
			bad[LOG_D2] += same(div_d2(f, den), want) ? 0 : 1;
			bad[LOG_D4] += same(div_d4(f, den), want) ? 0 : 1;
		}
		// (4) general quotients: inside div3's box, and over (nearly) the whole exponent range with v_div_fixup
		{
			uint32_t h = r ^ 0x9e3779b9u;
			float a = rand_float_exp(h, 127 - 60, 110), b = rand_float_exp(h, 127 - 40, 80);
			bad[GEN_D2] += same(div_d2(a, b), a / b) ? 0 : 1;
			if ((r & 15u) == 3u) a = (r & 16u) ? 0.0f : -0.0f;
			bad[GEN_FIX_BOX] += same(div_fix(a, b), a / b) ? 0 : 1;
			float aw = rand_float_exp(h, 1, 253), bw = rand_float_exp(h, 1, 253);
			if ((r & 31u) == 5u) aw = dm_u2f(mix32(h));
			if ((r & 63u) == 9u) bw = dm_u2f(mix32(h));
			bad[GEN_FIX_WIDE] += same(div_fix(aw, bw), aw / bw) ? 0 : 1;
			bad[GEN_FIX1_WIDE] += same(div_fix1(aw, bw), aw / bw) ? 0 : 1;
		}
		// (5) camera: (px + u) / W with the host's correctly rounded 1 / W
		{
			const float W = widths[r & 7u];
			const float px = (float)((r >> 3) % (uint32_t)W);
			const float a = px + u;
			bad[CAM_HOSTRCP] += same(div_core(a, W, 1.0f / W), a / W) ? 0 : 1;
		}
		// (6) cosine sign by integer ops
		{
			const float th = 6.28318548f * u;
			bad[COS_ALT] += same(cos_cur(th, true), cos_cur(th, false)) ? 0 : 1;
		}
		// (7) glass: sqrt(1 - c*c), sqrt(|1 - l|) style arguments: 1 - v for v = u*u (0 or >= 2^-25 in magnitude)
		{
			const float a1 = 1.0f - u * u;
			bad[SQ_S1_GLASS] += (a1 == 0.0f || same(sqrt_s1(a1), __builtin_sqrtf(a1))) ? 0 : 1;
		}
	}
	for (int k = 0; k < N_OUT; k++) atomicAdd(&out[k], bad[k]);
}

int main() {
	unsigned long long *d = nullptr, h[N_OUT];
	hipMalloc(&d, sizeof h);
	hipMemset(d, 0, sizeof h);
	hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, d);
	hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
	const char *names[N_OUT] = {"sqrt S1 (rsq+1 Markstein) normal x>=2^-96", "sqrt S2 (rsq+Goldschmidt+Markstein)", "sqrt S3 (v_sqrt + rsq correction)", "sqrt S4 (rsq + 2 Markstein)",
	                            "Box-Muller sqrt S1 (all u)", "Box-Muller sqrt S5 (clamped rsq, u != 0)", "Box-Muller sqrt S3 (all u)", "log div D1 (raw rcp, 1 corr)", "log div D2 (refined rcp, 1 corr)",
	                            "log div D4 (raw rcp, 2 corr)", "general div D2 in the box", "general div fixup (box, zeros)", "general div fixup (wide exponents)", "general div fixup 1 corr (wide)",
	                            "camera div with host 1/W", "cos sign by integer ops", "glass sqrt S1 (1 - u^2, nonzero)"};
	for (int k = 0; k < N_OUT; k++) printf("%-48s mismatches %llu\n", names[k], h[k]);
	return 0;
}
