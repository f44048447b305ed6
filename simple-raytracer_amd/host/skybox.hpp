// skybox.hpp — the texel preparation behind the reference's skybox upload, host only (no device, no decoder).
//
// The reference decodes assets/skybox.png with stb_image's FLOAT loader, four channels, flipped
// (/root/reference/src/tracer.cpp:42-46: stbi_set_flip_vertically_on_load(1); stbi_loadf_from_file(..., 4)) and uploads the
// result as the CL_RGBA / CL_FLOAT image `sky_box` samples (:48-55). For an 8-bit image stb does, in this order
// (/root/reference/lib/stb_image.h): convert to four channels (stbi__convert_format :1760-1790: grey -> r = g = b, missing alpha ->
// 255), flip the rows (the last row of the file becomes row 0, i.e. t = 0 at the bottom of the picture), then stbi__ldr_to_hdr
// (:1857-1878) with stbi__l2h_gamma = 2.2f and stbi__l2h_scale = 1.0f (:1572):
//     colour channels   (float)(pow(byte / 255.0f, 2.2f) * 1.0f)     -- the quotient in float, pow in double on the promoted operands
//     alpha             byte / 255.0f
// The C ABI takes the decoded RGBA32F image (srt_set_skybox); a front-end without stb calls this on the bytes its own decoder
// gives it and gets the reference's texels. PNG decoding itself stays out of scope.
//
// pow() is the platform's libm: glibc's is correctly rounded in all but astronomically rare cases and the result is then
// rounded to float, so the 256 values are the same on every glibc; tests/golden/skybox_l2h_table.json pins them
// (tests/test_skybox_prep.py, CPU).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>

// the colour-channel map for one byte (stbi__ldr_to_hdr, stb_image.h:1868)
inline float srt_skybox_l2h(uint8_t byte) { return (float)(std::pow((double)((float)byte / 255.0f), (double)2.2f) * (double)1.0f); }
// the alpha map (stb_image.h:1873)
inline float srt_skybox_alpha(uint8_t byte) { return (float)byte / 255.0f; }

// rgb: h rows of w pixels of `channels` bytes (1 grey, 2 grey + alpha, 3 RGB, 4 RGBA), first row = TOP of the picture, as a PNG
// decoder hands them out. rgba_out: w * h * 4 floats, row 0 = BOTTOM of the picture -- what srt_set_skybox / Tracer::set_skybox
// take. Returns false for arguments it cannot use.
inline bool srt_skybox_from_rgb8(const uint8_t *rgb, int w, int h, int channels, float *rgba_out) {
	if (!rgb || !rgba_out || w <= 0 || h <= 0 || channels < 1 || channels > 4) return false;
	float lut[256];
	for (int i = 0; i < 256; i++) lut[i] = srt_skybox_l2h((uint8_t)i);
	for (int y = 0; y < h; y++) {
		const uint8_t *src = rgb + (size_t)(h - 1 - y) * (size_t)w * (size_t)channels; // vertical flip
		float *dst = rgba_out + (size_t)y * (size_t)w * 4;
		for (int x = 0; x < w; x++, src += channels, dst += 4) {
			const bool grey = channels < 3;
			dst[0] = lut[src[0]];
			dst[1] = lut[grey ? src[0] : src[1]];
			dst[2] = lut[grey ? src[0] : src[2]];
			dst[3] = (channels == 2 || channels == 4) ? srt_skybox_alpha(src[channels - 1]) : 1.0f; // alpha 255 / 255.0f
		}
	}
	return true;
}
