// material.hpp — same interface and byte layout as the reference's include/material.hpp
// (64 bytes: six floats, then color@32 and emission@48 as 16-byte float3 slots).
#pragma once

#include <cstddef>

#include "../../include/srt_types.h"
#include "color.hpp"

struct Material {
	float smoothness;
	float metallic;          ///< tinted reflection
	float specular;          ///< untinted reflection
	float emission_strength;
	float transmittance;
	float refraction_index;

	alignas(cl_float3) Color color;
	alignas(cl_float3) Color emission;

	Material(
		const Color &color = color::white, float smoothness = 0.0f, float metallic = 0.0f, float specular = 0.0f,
		float transmittance = 0.0f, float refraction_index = 1.0f, const Color &emission = color::black,
		float emission_strength = 0.0f
	)
		: smoothness(smoothness), metallic(metallic), specular(specular), emission_strength(emission_strength),
		  transmittance(transmittance), refraction_index(refraction_index), color(color), emission(emission) {}
};

static_assert(sizeof(Material) == sizeof(srt_material), "Material must match srt_material");
static_assert(offsetof(Material, color) == offsetof(srt_material, color), "Material.color");
static_assert(offsetof(Material, emission) == offsetof(srt_material, emission), "Material.emission");
static_assert(offsetof(Material, transmittance) == offsetof(srt_material, transmittance), "Material.transmittance");
