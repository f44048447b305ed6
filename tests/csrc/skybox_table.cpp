// prints the 256 colour texels and the 256 alpha texels of host/skybox.hpp as hexadecimal float bit patterns, then the
// RGBA32F image it makes of a small RGB / RGBA / grey test picture read from stdin (tests/test_skybox_prep.py)
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../simple-raytracer_amd/host/skybox.hpp"

static unsigned bits(float f) {
	unsigned u;
	std::memcpy(&u, &f, 4);
	return u;
}
int main(int argc, char **argv) {
	for (int i = 0; i < 256; i++) std::printf("%08x %08x\n", bits(srt_skybox_l2h((uint8_t)i)), bits(srt_skybox_alpha((uint8_t)i)));
	int w, h, c;
	if (std::scanf("%d %d %d", &w, &h, &c) != 3) return 0;
	std::vector<uint8_t> img((size_t)w * h * c);
	for (auto &b : img) {
		int v;
		if (std::scanf("%d", &v) != 1) return 2;
		b = (uint8_t)v;
	}
	std::vector<float> out((size_t)w * h * 4);
	if (!srt_skybox_from_rgb8(img.data(), w, h, c, out.data())) return 3;
	std::printf("image\n");
	for (float f : out) std::printf("%08x\n", bits(f));
	return 0;
}
