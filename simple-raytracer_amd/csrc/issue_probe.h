// issue_probe.h -- included INSIDE srt_trace_kernel's main loop by -DSRT_DUMMY_KIND=k builds only (scripts/r04_issue_cost.sh).
// 100 extra instructions of ONE kind per loop iteration that compute nothing: what an instruction of each kind costs the launch
// tells what the kernel's time is made of (profiles/r04_valu_issue.json). Uses the loop's `org` and `dir`.
		// (regime probe, development builds only; scripts/r04_issue_cost.sh) 100 extra instructions of one kind per loop iteration
		// that compute nothing: what an instruction of each kind costs the launch tells what the kernel's time is made of
		{
			float d0 = org.x, d1 = org.y, d2 = org.z, d3 = dir.x;
			uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#define SRT_DUMMY4(I) asm volatile(".rept 25\n " I(0, 4) "\n " I(1, 5) "\n " I(2, 6) "\n " I(3, 7) "\n .endr" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(dir.y), "v"(dir.z) : "scc", "vcc")
#if SRT_DUMMY_KIND == 1
#define SRT_DI(k, q) "v_add_f32 %" #k ", 1.0, %" #k
#elif SRT_DUMMY_KIND == 2
#define SRT_DI(k, q) "v_max_f32 %" #k ", 1.0, %" #k
#elif SRT_DUMMY_KIND == 3
#define SRT_DI(k, q) "s_add_u32 %" #q ", %" #q ", 1"
#elif SRT_DUMMY_KIND == 4
#define SRT_DI(k, q) "v_fma_f32 %" #k ", %8, %9, %" #k
#elif SRT_DUMMY_KIND == 5
#define SRT_DI(k, q) "v_mul_lo_u32 %" #k ", %" #k ", %8"
#elif SRT_DUMMY_KIND == 6
#define SRT_DI(k, q) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc"
#elif SRT_DUMMY_KIND == 7
#define SRT_DI(k, q) "v_cmp_lt_f32 vcc, %" #k ", %8"
#elif SRT_DUMMY_KIND == 8
#define SRT_DI(k, q) "v_rcp_f32 %" #k ", %" #k
#elif SRT_DUMMY_KIND == 9
#define SRT_DI(k, q) "v_mov_b32 %" #k ", %8"
#elif SRT_DUMMY_KIND == 10
#define SRT_DI(k, q) "s_mov_b32 %" #q ", 0x12345678"
#elif SRT_DUMMY_KIND == 11
#define SRT_DI(k, q) "s_nop 0"
#elif SRT_DUMMY_KIND == 12
#define SRT_DI(k, q) "v_add_f32 %" #k ", 0x40490fdb, %" #k
#elif SRT_DUMMY_KIND == 13
#define SRT_DI(k, q) "v_xor_b32 %" #k ", %8, %" #k
#elif SRT_DUMMY_KIND == 14
#define SRT_DI(k, q) "v_add_u32 %" #k ", %8, %" #k
#elif SRT_DUMMY_KIND == 15
#define SRT_DI(k, q) "v_cvt_f32_u32 %" #k ", %" #k
#elif SRT_DUMMY_KIND == 16
#define SRT_DI(k, q) "s_and_b64 vcc, vcc, exec"
#elif SRT_DUMMY_KIND == 17
#define SRT_DI(k, q) "v_readfirstlane_b32 %" #q ", %" #k
#elif SRT_DUMMY_KIND == 18
#define SRT_DI(k, q) "s_cmp_eq_u32 %" #q ", 77\n s_cbranch_scc1 1f\n 1:" // compare + branch not taken
#elif SRT_DUMMY_KIND == 21
#define SRT_DI(k, q) "s_cmp_lg_u32 %" #q ", 77\n s_cbranch_scc1 1f\n 1:" // compare + branch taken (to the next instruction)
#elif SRT_DUMMY_KIND == 22
#define SRT_DI(k, q) "v_sub_f32 %" #k ", %" #k ", %8\n v_mul_f32 %" #k ", %" #k ", %9" // two dependent full-rate instructions
#elif SRT_DUMMY_KIND == 19
#define SRT_DI(k, q) "v_mul_f32 %" #k ", %8, %" #k
#elif SRT_DUMMY_KIND == 20
#define SRT_DI(k, q) "v_lshlrev_b32 %" #k ", 1, %" #k
#endif
			SRT_DUMMY4(SRT_DI);
		}
