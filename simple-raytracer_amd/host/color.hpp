// color.hpp — same interface as the reference's include/color.hpp.
#pragma once

#include <cstdint>

#include "srt_compat.hpp"

typedef glm::vec3 Color;

namespace color {
const Color white(1.0f);
const Color black(0.0f);
const Color gray(0.5f);

/// 0xRRGGBB -> linear [0,1] components (byte / 255)
inline Color from_hex(const uint32_t value) {
	return Color(float((value >> 16) & 0xFF) / 255.f, float((value >> 8) & 0xFF) / 255.f, float(value & 0xFF) / 255.f);
}

inline Color from_RGB(const uint8_t r, const uint8_t g, const uint8_t b) { return Color(r / 255.f, g / 255.f, b / 255.f); }
} // namespace color
