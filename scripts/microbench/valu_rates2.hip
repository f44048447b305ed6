// valu_rates2.hip — saturated throughput (8 waves per SIMD, wall clock) of the VALU instructions the trace kernel is made of.
// gfx950 does not run every VALU instruction at the rate of v_fma_f32: this prints wave-instructions per SIMD per ns for each.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 valu_rates2.hip -o bin/valu_rates2 && bin/valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 65536
#define OPS(X) \
	X(0, "v_fma_f32", "v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %0") \
	X(1, "v_add_f32", "v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0") \
	X(2, "v_mul_f32", "v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %0") \
	X(3, "v_fmac_f32", "v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_fmac_f32 %3, %0, %1") \
	X(4, "v_max_f32", "v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %0") \
	X(5, "v_min3_f32", "v_min3_f32 %0, %0, %1, %2\n v_min3_f32 %1, %1, %2, %3\n v_min3_f32 %2, %2, %3, %0\n v_min3_f32 %3, %3, %0, %1") \
	X(6, "v_add_u32(vgpr)", "v_add_u32 %4, %4, %5\n v_add_u32 %5, %5, %6\n v_add_u32 %6, %6, %7\n v_add_u32 %7, %7, %4") \
	X(7, "v_sub_u32", "v_sub_u32 %4, %4, %5\n v_sub_u32 %5, %5, %6\n v_sub_u32 %6, %6, %7\n v_sub_u32 %7, %7, %4") \
	X(8, "v_xor_b32", "v_xor_b32 %4, %4, %5\n v_xor_b32 %5, %5, %6\n v_xor_b32 %6, %6, %7\n v_xor_b32 %7, %7, %4") \
	X(9, "v_and_b32", "v_and_b32 %4, %4, %5\n v_and_b32 %5, %5, %6\n v_and_b32 %6, %6, %7\n v_and_b32 %7, %7, %4") \
	X(10, "v_or_b32", "v_or_b32 %4, %4, %5\n v_or_b32 %5, %5, %6\n v_or_b32 %6, %6, %7\n v_or_b32 %7, %7, %4") \
	X(11, "v_lshrrev_b32(imm)", "v_lshrrev_b32 %4, 3, %4\n v_lshrrev_b32 %5, 5, %5\n v_lshrrev_b32 %6, 7, %6\n v_lshrrev_b32 %7, 9, %7") \
	X(12, "v_lshlrev_b32(imm)", "v_lshlrev_b32 %4, 3, %4\n v_lshlrev_b32 %5, 5, %5\n v_lshlrev_b32 %6, 7, %6\n v_lshlrev_b32 %7, 9, %7") \
	X(13, "v_bfe_u32", "v_bfe_u32 %4, %4, 3, 20\n v_bfe_u32 %5, %5, 5, 20\n v_bfe_u32 %6, %6, 7, 20\n v_bfe_u32 %7, %7, 9, 20") \
	X(14, "v_bfi_b32", "v_bfi_b32 %4, %4, %5, %6\n v_bfi_b32 %5, %5, %6, %7\n v_bfi_b32 %6, %6, %7, %4\n v_bfi_b32 %7, %7, %4, %5") \
	X(15, "v_lshl_add_u32", "v_lshl_add_u32 %4, %4, 3, %5\n v_lshl_add_u32 %5, %5, 3, %6\n v_lshl_add_u32 %6, %6, 3, %7\n v_lshl_add_u32 %7, %7, 3, %4") \
	X(16, "v_add3_u32", "v_add3_u32 %4, %4, %5, %6\n v_add3_u32 %5, %5, %6, %7\n v_add3_u32 %6, %6, %7, %4\n v_add3_u32 %7, %7, %4, %5") \
	X(17, "v_xad_u32", "v_xad_u32 %4, %4, %5, %6\n v_xad_u32 %5, %5, %6, %7\n v_xad_u32 %6, %6, %7, %4\n v_xad_u32 %7, %7, %4, %5") \
	X(18, "v_mul_lo_u32", "v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %5, %5, %6\n v_mul_lo_u32 %6, %6, %7\n v_mul_lo_u32 %7, %7, %4") \
	X(19, "v_mad_u32_u24", "v_mad_u32_u24 %4, %4, %5, %6\n v_mad_u32_u24 %5, %5, %6, %7\n v_mad_u32_u24 %6, %6, %7, %4\n v_mad_u32_u24 %7, %7, %4, %5") \
	X(20, "v_min_u32", "v_min_u32 %4, %4, %5\n v_min_u32 %5, %5, %6\n v_min_u32 %6, %6, %7\n v_min_u32 %7, %7, %4") \
	X(21, "v_mov_b32", "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %4") \
	X(22, "v_cndmask_b32(vcc const)", "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %4, vcc") \
	X(23, "v_cmp_lt_f32", "v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0") \
	X(24, "v_cmp_lt_u32", "v_cmp_lt_u32 vcc, %4, %5\n v_cmp_lt_u32 vcc, %5, %6\n v_cmp_lt_u32 vcc, %6, %7\n v_cmp_lt_u32 vcc, %7, %4") \
	X(25, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %4\n v_cvt_f32_u32 %1, %5\n v_cvt_f32_u32 %2, %6\n v_cvt_f32_u32 %3, %7") \
	X(26, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7") \
	X(27, "v_cvt_i32_f32", "v_cvt_i32_f32 %4, %0\n v_cvt_i32_f32 %5, %1\n v_cvt_i32_f32 %6, %2\n v_cvt_i32_f32 %7, %3") \
	X(28, "v_rsq_f32", "v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3") \
	X(29, "v_med3_f32", "v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_med3_f32 %2, %2, %3, %0\n v_med3_f32 %3, %3, %0, %1") \
	X(30, "v_fmaak_f32", "v_fmaak_f32 %0, %0, %1, 0x3f317180\n v_fmaak_f32 %1, %1, %2, 0x3f317180\n v_fmaak_f32 %2, %2, %3, 0x3f317180\n v_fmaak_f32 %3, %3, %0, 0x3f317180") \
	X(31, "v_mul_f32(literal)", "v_mul_f32 %0, 0x3a83126f, %0\n v_mul_f32 %1, 0x3a83126f, %1\n v_mul_f32 %2, 0x3a83126f, %2\n v_mul_f32 %3, 0x3a83126f, %3") \
	X(32, "v_mbcnt_lo", "v_mbcnt_lo_u32_b32 %4, %5, %4\n v_mbcnt_lo_u32_b32 %5, %6, %5\n v_mbcnt_lo_u32_b32 %6, %7, %6\n v_mbcnt_lo_u32_b32 %7, %4, %7") \
	X(33, "v_mul_f64", "v_mul_f64 %8, %8, %9\n v_mul_f64 %9, %9, %8\n v_mul_f64 %8, %8, %9\n v_mul_f64 %9, %9, %8") \
	X(34, "v_lshrrev_b32(vgpr)", "v_lshrrev_b32 %4, %5, %4\n v_lshrrev_b32 %5, %6, %5\n v_lshrrev_b32 %6, %7, %6\n v_lshrrev_b32 %7, %4, %7") \
	X(35, "v_add_u32(literal)", "v_add_u32 %4, 0x4712a88e, %4\n v_add_u32 %5, 0x4712a88e, %5\n v_add_u32 %6, 0x4712a88e, %6\n v_add_u32 %7, 0x4712a88e, %7") \
	X(37, "v_cndmask x4 independent", "v_cndmask_b32 %4, %0, %1, vcc\n v_cndmask_b32 %5, %1, %2, vcc\n v_cndmask_b32 %6, %2, %3, vcc\n v_cndmask_b32 %7, %3, %0, vcc") \
	X(38, "v_cndmask + v_mul_f32 1:1", "v_cndmask_b32 %4, %4, %5, vcc\n v_mul_f32 %0, %0, %1\n v_cndmask_b32 %6, %6, %7, vcc\n v_mul_f32 %2, %2, %3") \
	X(39, "v_cndmask e64 (sgpr mask)", "v_cndmask_b32 %4, %4, %5, s[20:21]\n v_cndmask_b32 %5, %5, %6, s[20:21]\n v_cndmask_b32 %6, %6, %7, s[20:21]\n v_cndmask_b32 %7, %7, %4, s[20:21]") \
	X(40, "v_cmp_f32 -> v_cndmask pairs", "v_cmp_lt_f32 vcc, %0, %1\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %2\n v_cndmask_b32 %4, %4, %5, vcc") \
	X(41, "v_cndmask (const 0/1.0)", "v_cndmask_b32 %4, 0, 1.0, vcc\n v_cndmask_b32 %5, 0, 1.0, vcc\n v_cndmask_b32 %6, 0, 1.0, vcc\n v_cndmask_b32 %7, 0, 1.0, vcc") \
	X(42, "v_mul_f32 x4 (control)", "v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %0") \
	X(43, "v_cndmask e64 (vcc)", "v_cndmask_b32_e64 %4, %0, %1, vcc\n v_cndmask_b32_e64 %5, %1, %2, vcc\n v_cndmask_b32_e64 %6, %2, %3, vcc\n v_cndmask_b32_e64 %7, %3, %0, vcc") \
	X(44, "2 cndmask e32 + 2 v_mul", "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_mul_f32 %0, %0, %1\n v_mul_f32 %2, %2, %3") \
	X(45, "3 cndmask e32 + 1 v_mul", "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %5, %5, %7, vcc\n v_mul_f32 %2, %2, %3") \
	X(46, "cndmask e32, s_nop 0 between", "v_cndmask_b32 %4, %0, %1, vcc\n s_nop 0\n v_cndmask_b32 %5, %1, %2, vcc\n s_nop 0\n v_cndmask_b32 %6, %2, %3, vcc\n s_nop 0\n v_cndmask_b32 %7, %3, %0, vcc\n s_nop 0") \
	X(47, "cndmask e32 sdwa-free src0 const", "v_cndmask_b32 %4, 1.0, %1, vcc\n v_cndmask_b32 %5, 1.0, %2, vcc\n v_cndmask_b32 %6, 1.0, %3, vcc\n v_cndmask_b32 %7, 1.0, %0, vcc") \
	X(48, "v_pk_fma_f32", "v_pk_fma_f32 %8, %8, %9, %8\n v_pk_fma_f32 %9, %9, %8, %9\n v_pk_fma_f32 %8, %8, %9, %8\n v_pk_fma_f32 %9, %9, %8, %9") \
	X(49, "v_pk_mul_f32", "v_pk_mul_f32 %8, %8, %9\n v_pk_mul_f32 %9, %9, %8\n v_pk_mul_f32 %8, %8, %9\n v_pk_mul_f32 %9, %9, %8") \
	X(50, "v_pk_add_f32", "v_pk_add_f32 %8, %8, %9\n v_pk_add_f32 %9, %9, %8\n v_pk_add_f32 %8, %8, %9\n v_pk_add_f32 %9, %9, %8") \
	X(51, "v_pk_fma_f32 (sgpr pair src)", "v_pk_fma_f32 %8, s[20:21], %9, %8\n v_pk_fma_f32 %9, s[20:21], %8, %9\n v_pk_fma_f32 %8, s[20:21], %9, %8\n v_pk_fma_f32 %9, s[20:21], %8, %9") \
	X(52, "v_pk_mul_f32 (sgpr pair, neg)", "v_pk_mul_f32 %8, %9, s[20:21] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_mul_f32 %9, %8, s[20:21] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_mul_f32 %8, %9, s[20:21] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_mul_f32 %9, %8, s[20:21] neg_lo:[0,1] neg_hi:[0,1]") \
	X(53, "v_pk_mul_f32 (op_sel_hi splat)", "v_pk_mul_f32 %8, %8, %9 op_sel_hi:[1,0]\n v_pk_mul_f32 %9, %9, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %8, %8, %9 op_sel_hi:[1,0]\n v_pk_mul_f32 %9, %9, %8 op_sel_hi:[1,0]") \
	X(54, "v_div_scale_f32", "v_div_scale_f32 %0, vcc, %0, %1, %0\n v_div_scale_f32 %1, vcc, %1, %2, %1\n v_div_scale_f32 %2, vcc, %2, %3, %2\n v_div_scale_f32 %3, vcc, %3, %0, %3") \
	X(55, "v_div_fmas_f32", "v_div_fmas_f32 %0, %0, %1, %2\n v_div_fmas_f32 %1, %1, %2, %3\n v_div_fmas_f32 %2, %2, %3, %0\n v_div_fmas_f32 %3, %3, %0, %1") \
	X(56, "v_div_fixup_f32", "v_div_fixup_f32 %0, %0, %1, %2\n v_div_fixup_f32 %1, %1, %2, %3\n v_div_fixup_f32 %2, %2, %3, %0\n v_div_fixup_f32 %3, %3, %0, %1") \
	X(57, "v_rcp_f32", "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3") \
	X(58, "v_fma_f32 (2 vgpr + sgpr)", "v_fma_f32 %0, %0, s20, %1\n v_fma_f32 %1, %1, s20, %2\n v_fma_f32 %2, %2, s20, %3\n v_fma_f32 %3, %3, s20, %0") \
	X(59, "v_fma_f32 (neg modifier)", "v_fma_f32 %0, -%0, %1, %1\n v_fma_f32 %1, -%1, %2, %2\n v_fma_f32 %2, -%2, %3, %3\n v_fma_f32 %3, -%3, %0, %0") \
	X(60, "v_cmp_class_f32", "v_cmp_class_f32 vcc, %0, %4\n v_cmp_class_f32 vcc, %1, %5\n v_cmp_class_f32 vcc, %2, %6\n v_cmp_class_f32 vcc, %3, %7") \
	X(36, "v_add_u32(inline 4)", "v_add_u32 %4, 4, %4\n v_add_u32 %5, 4, %5\n v_add_u32 %6, 4, %6\n v_add_u32 %7, 4, %7")

template <int OP>
__global__ void k(uint32_t *out, uint32_t a) {
	uint32_t x0 = threadIdx.x + a, x1 = x0 * 3u, x2 = x0 * 5u, x3 = x0 * 7u;
	float f0 = (float)x0 * 1e-3f, f1 = f0 + 1.f, f2 = f0 + 2.f, f3 = f0 + 3.f;
	double d0 = f0, d1 = f1;
	for (int i = 0; i < N; i++) {
#define CASE(id, name, text) if (OP == id) asm volatile(text : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(d0), "+v"(d1) : : "vcc", "s20", "s21");
		OPS(CASE)
#undef CASE
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ (uint32_t)(f0 + f1 + f2 + f3) ^ (uint32_t)(d0 + d1);
}
template <int OP>
void run(const char *name, uint32_t *out) {
	const int waves = 8, blocks = 256 * 4 * waves;
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, 1u);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, 1u);
	hipEventRecord(e1, 0);
	hipDeviceSynchronize();
	float ms = 0.f;
	hipEventElapsedTime(&ms, e0, e1);
	const double per_ns = (double)N * 4 * waves / (ms * 1e6);
	printf("%-26s %.3f wave-instructions per SIMD per ns  (%.2f x the time of v_fma_f32 at 1.0)\n", name, per_ns, 1.0 / per_ns);
}
int main() {
	uint32_t *out;
	hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
#define RUN(id, name, text) run<id>(name, out);
	OPS(RUN)
	return 0;
}
