// ta_rates.hip -- what a vector memory LOAD costs a CU's address / data-return path on gfx950, by width, by the lanes
// that execute it and by the cache lines it touches (the question behind profiles/r04_bvh_vmem_probe.md).
// 20 one-wave workgroups per CU (the BVH kernel's residency), every wave issues clauses of 7 independent loads, waits,
// folds the data into the next clause's addresses (a walk's dependency) and repeats. Addresses stay inside an 8 MB
// buffer (the 10^5-triangle hierarchy's size). Reported: nanoseconds per load instruction and CU (wall time x CUs /
// instructions), which at ~2.2 GHz is ~2.2 cycles per ns.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 ta_rates.hip -o ta_rates && ./ta_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define ITERS 3000
#define LOADS 7
#define BUF_BYTES (8u << 20)

// LANES: how many lanes execute the loads (the lowest ones; the others are masked off by a branch)
// LINES: 0 = every lane its own random 128-byte line; 1 = all lanes the same line (same 16 bytes); k > 1 = lanes spread over k random lines
// W: dwords per lane and load (1, 2, 4)
template <int LANES, int LINES, int W>
__global__ __launch_bounds__(64) void k(const uint32_t *__restrict__ buf, uint32_t *out, uint32_t seed) {
	const uint32_t lane = threadIdx.x;
	uint32_t h = (blockIdx.x * 64u + lane) * 2654435761u + seed;
	uint32_t acc = 0;
	if ((int)lane < LANES) {
		for (int it = 0; it < ITERS; it++) {
			h = h * 747796405u + 2891336453u;
			uint32_t pick = h;
			if (LINES == 1) pick = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
			else if (LINES > 1) pick = (uint32_t)__builtin_amdgcn_readfirstlane((int)h) + (lane % LINES) * 0x9e3779b9u;
			const uint32_t line = (pick >> 7) % (BUF_BYTES / 128u); // a 128-byte line of the buffer
			const char *base = reinterpret_cast<const char *>(buf) + (size_t)line * 128u;
			uint32_t v[LOADS];
#pragma unroll
			for (int q = 0; q < LOADS; q++) { // seven loads into the same line, as a step of the walk issues them
				if (W == 4) {
					const uint4 x = *reinterpret_cast<const uint4 *>(base + 16 * q);
					v[q] = x.x ^ x.y ^ x.z ^ x.w;
				} else if (W == 2) {
					const uint2 x = *reinterpret_cast<const uint2 *>(base + 16 * q);
					v[q] = x.x ^ x.y;
				} else {
					v[q] = *reinterpret_cast<const uint32_t *>(base + 16 * q);
				}
			}
#pragma unroll
			for (int q = 0; q < LOADS; q++) acc ^= v[q];
			h ^= acc; // the next addresses depend on what came back
		}
	}
	out[blockIdx.x * 64u + lane] = acc ^ h;
}

template <int LANES, int LINES, int W>
void run(const char *name, const uint32_t *buf, uint32_t *out) {
	const int blocks = 256 * 20;
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	hipLaunchKernelGGL((k<LANES, LINES, W>), dim3(blocks), dim3(64), 0, 0, buf, out, 1u);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL((k<LANES, LINES, W>), dim3(blocks), dim3(64), 0, 0, buf, out, 2u);
	hipEventRecord(e1, 0);
	hipDeviceSynchronize();
	float ms = 0.f;
	hipEventElapsedTime(&ms, e0, e1);
	const double per_cu = (double)ITERS * LOADS * 20; // load instructions per CU
	printf("%-62s %8.3f ms  %7.2f ns per load instruction and CU\n", name, ms, ms * 1e6 / per_cu);
	fflush(stdout);
}

int main() {
	uint32_t *buf, *out;
	hipMalloc(&buf, BUF_BYTES);
	hipMalloc(&out, 256 * 20 * 64 * 4);
	std::vector<uint32_t> hbuf(BUF_BYTES / 4);
	uint32_t s = 12345u;
	for (auto &x : hbuf) x = (s = s * 1664525u + 1013904223u);
	hipMemcpy(buf, hbuf.data(), BUF_BYTES, hipMemcpyHostToDevice);
	printf("# 256 CUs x 20 waves, %d clauses of %d loads per wave, 8 MB buffer\n", ITERS, LOADS);
	run<64, 0, 4>("64 lanes, 64 lines, 16 B (a fully divergent wave-step)", buf, out);
	run<26, 0, 4>("26 lanes, 26 lines, 16 B (the walk's average step)", buf, out);
	run<8, 0, 4>(" 8 lanes,  8 lines, 16 B", buf, out);
	run<1, 0, 4>(" 1 lane,   1 line,  16 B", buf, out);
	run<64, 1, 4>("64 lanes,  1 line,  16 B (all lanes the same address)", buf, out);
	run<64, 8, 4>("64 lanes,  8 lines, 16 B", buf, out);
	run<26, 8, 4>("26 lanes,  8 lines, 16 B", buf, out);
	run<26, 1, 4>("26 lanes,  1 line,  16 B", buf, out);
	run<64, 0, 2>("64 lanes, 64 lines,  8 B", buf, out);
	run<64, 0, 1>("64 lanes, 64 lines,  4 B", buf, out);
	run<26, 0, 2>("26 lanes, 26 lines,  8 B", buf, out);
	run<26, 0, 1>("26 lanes, 26 lines,  4 B", buf, out);
	return 0;
}
