// magic_check.cpp — srt_magic_u31 (csrc/device_types.h): n / d == (mulhi(n, magic) + n) >> shift for every n < 2^31 the test
// tries (all n up to 2^16 around multiples of d, powers of two, the top of the range, and a pseudo-random sweep) and for the
// divisors an image width / block height can take. Also the 16-bit reciprocal of the sample-batch size. Prints "ok" or the
// first counter-example. Build: g++ -std=c++17 -O2 -I simple-raytracer_amd/csrc tests/csrc/magic_check.cpp -o magic_check
#include <cstdint>
#include <cstdio>

#include "device_types.h"

static bool check(uint32_t d) {
	uint32_t m, s;
	srt_magic_u31(d, &m, &s);
	auto q = [&](uint32_t n) { return (uint32_t)(((uint32_t)(((uint64_t)n * m) >> 32) + n) >> s); };
	auto bad = [&](uint32_t n) {
		if (q(n) != n / d) {
			printf("srt_magic_u31(%u): n = %u gives %u, want %u\n", d, n, q(n), n / d);
			return true;
		}
		return false;
	};
	for (uint32_t n = 0; n < 70000u; n++)
		if (bad(n)) return false;
	for (uint64_t k = 1; k * d < 0x80000000ull; k = k * 3 + 1)
		for (int o = -2; o <= 2; o++) {
			const uint64_t n = k * d + (uint64_t)(int64_t)o;
			if (n < 0x80000000ull && bad((uint32_t)n)) return false;
		}
	for (uint32_t n = 0x7fffff00u; n < 0x80000000u; n++)
		if (bad(n)) return false;
	uint32_t x = 12345u;
	for (int i = 0; i < 2000000; i++) {
		x = x * 747796405u + 2891336453u;
		if (bad(x >> 1)) return false;
	}
	return true;
}

int main() {
	const uint32_t ds[] = {1, 2, 3, 5, 7, 8, 16, 37, 41, 64, 96, 135, 256, 480, 540, 960, 1000, 1080, 1920, 2160, 3840, 4096, 65535, 65536, 65537, 1u << 24, (1u << 24) + 1, 0x7fffffffu};
	for (uint32_t d : ds)
		if (!check(d)) return 1;
	for (uint32_t d = 1; d < 3000; d++) {
		uint32_t m, s;
		srt_magic_u31(d, &m, &s);
		for (uint32_t n = 0; n < 0x80000000u; n += 0x00fffff1u)
			if ((uint32_t)(((uint32_t)(((uint64_t)n * m) >> 32) + n) >> s) != n / d) {
				printf("srt_magic_u31(%u) fails at %u\n", d, n);
				return 1;
			}
	}
	// n / nbs = (n * ceil(2^16 / nbs)) >> 16 for n < 256 and nbs < 128 (kernels.hip CAMERA)
	for (uint32_t nbs = 1; nbs < 128; nbs++) {
		const uint32_t m16 = (65536u + nbs - 1u) / nbs;
		for (uint32_t n = 0; n < 256; n++)
			if (((n * m16) >> 16) != n / nbs) {
				printf("16-bit reciprocal of %u fails at %u\n", nbs, n);
				return 1;
			}
	}
	printf("ok\n");
	return 0;
}
