// valu_rates.hip — issue cost of a few VALU instructions on gfx950, one wave per SIMD and 4 waves per SIMD.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 131072
template <int OP>
__global__ void k(uint32_t *out, uint64_t *cyc, uint32_t a, uint32_t b) {
	uint32_t x0 = threadIdx.x + a, x1 = x0 * 3u, x2 = x0 * 5u, x3 = x0 * 7u;
	float f0 = (float)x0, f1 = f0 + 1.f, f2 = f0 + 2.f, f3 = f0 + 3.f;
	const uint64_t t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < N; i++) {
		// four independent chains per iteration
		if (OP == 0) { asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(b)); }
		if (OP == 1) { asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(b)); }
		if (OP == 2) { asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(b)); }
		if (OP == 3) { asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(b)); }
		if (OP == 4) { asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %0" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)); }
		if (OP == 5) { asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)); }
		if (OP == 6) { asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)); }
		if (OP == 7) { asm volatile("v_cvt_f32_u32 %0, %4\n v_cvt_f32_u32 %1, %5\n v_cvt_f32_u32 %2, %6\n v_cvt_f32_u32 %3, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3)); }
		if (OP == 8) { asm volatile("v_lshrrev_b32 %0, %1, %0\n v_lshrrev_b32 %1, %2, %1\n v_lshrrev_b32 %2, %3, %2\n v_lshrrev_b32 %3, %0, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
		if (OP == 9) { asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : : "vcc"); }
		if (OP == 10) { asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 4\n v_readlane_b32 s22, %2, 5\n v_readlane_b32 s23, %3, 6" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : "s20", "s21", "s22", "s23"); }
		if (OP == 11) { asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0"); }
		if (OP == 12) { asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1" : : : "s20", "s21", "s22", "s23", "scc"); }
		if (OP == 13) { double d0 = f0, d1 = f1; asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %0\n v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %0" : "+v"(d0), "+v"(d1)); f0 = (float)d0; f1 = (float)d1; }
		if (OP == 14) { asm volatile("v_mad_u64_u32 %0, vcc, %2, %4, %0\n v_mad_u64_u32 %1, vcc, %3, %4, %1" : "+v"(*(uint64_t*)&x0), "+v"(*(uint64_t*)&x2) : "v"(x1), "v"(x3), "s"(b) : "vcc"); }
	}
	const uint64_t t1 = __builtin_amdgcn_s_memtime();
	out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ (uint32_t)(f0 + f1 + f2 + f3);
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP>
void run(const char *name, int per_iter, uint32_t *out, uint64_t *cyc) {
	for (int waves_per_simd : {1, 2, 4, 8}) {
		// one CU-filling launch: 256 CUs x 4 SIMDs x waves
		const int blocks = 256 * 4 * waves_per_simd;
		hipEvent_t e0, e1;
		hipEventCreate(&e0), hipEventCreate(&e1);
		hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, cyc, 1u, 12345u);
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, cyc, 1u, 12345u);
		hipEventRecord(e1, 0);
		hipDeviceSynchronize();
		float ms = 0.f;
		hipEventElapsedTime(&ms, e0, e1);
		// wall-clock view: wave-instructions per SIMD per nanosecond (at 2.4 GHz, 0.5 per cycle = 1.2 per ns is the VALU peak)
		const double per_ns = (double)N * per_iter * waves_per_simd / (ms * 1e6);
		uint64_t h[64];
		hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
		double avg = 0;
		for (int i = 0; i < 64; i++) avg += (double)h[i];
		avg /= 64;
		printf("%-16s waves/SIMD %d: %.2f s_memtime ticks per instruction per wave  (%.2f per SIMD); wall %.3f ms = %.3f wave-instructions per SIMD per ns\n", name, waves_per_simd, avg / N / per_iter, avg / N / per_iter / waves_per_simd, ms, per_ns);
	}
}
int main() {
	uint32_t *out; uint64_t *cyc;
	hipMalloc(&out, 256 * 4 * 8 * 64 * 4); hipMalloc(&cyc, 256 * 4 * 8 * 8);
	run<1>("v_add_u32", 4, out, cyc);
	run<0>("v_mul_lo_u32", 4, out, cyc);
	run<3>("v_mul_hi_u32", 4, out, cyc);
	run<2>("v_mul_u32_u24", 4, out, cyc);
	run<14>("v_mad_u64_u32", 2, out, cyc);
	run<4>("v_fma_f32", 4, out, cyc);
	run<5>("v_sqrt_f32", 4, out, cyc);
	run<6>("v_rcp_f32", 4, out, cyc);
	run<7>("v_cvt_f32_u32", 4, out, cyc);
	run<8>("v_lshrrev(vgpr)", 4, out, cyc);
	run<9>("cmp+cndmask", 4, out, cyc);
	run<10>("v_readlane", 4, out, cyc);
	run<11>("s_nop 0", 4, out, cyc);
	run<12>("s_add_u32", 4, out, cyc);
	run<13>("v_mul_f64", 4, out, cyc);
	return 0;
}
