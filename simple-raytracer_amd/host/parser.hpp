// parser.hpp — same interface as the reference's include/parser.hpp / src/parser.cpp:
// the on-disk formats either side of the hot path.
//
//   save_ppm        ARGB8 frame -> binary PPM ("P6 W H 255\n", alpha dropped)
//   load_stl_model  binary STL  -> flat-shaded Triangles (facet normal on all 3 corners)
//   load_obj_model  Wavefront OBJ (v / vn / f, triangulated) -> smooth-shaded Triangles
//
// Both loaders APPEND to `triangles` and return {first index, count}, or nullopt when
// the file cannot be opened — as the reference does. Deliberate differences (the
// reference's behaviour there is undefined, not a format rule):
//   * OBJ negative indices follow the OBJ spec (-1 = last element so far at the time
//     the whole file is read); the reference computes `len - index + 1` (out of range).
//   * A face without `vn` indices gets the geometric face normal on its three corners;
//     the reference leaves those normals uninitialised.
//   * A truncated STL stops at the last complete record instead of pushing garbage.
//   * Index values outside the vertex / normal lists make the loader return nullopt.
#pragma once

#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <optional>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "shape.hpp"

namespace fs = std::filesystem;

using ModelPair = std::pair<unsigned int, unsigned int>;

inline void save_ppm(const fs::path &filename, const std::vector<uint8_t> &pixels, int width, int height) {
	std::ofstream file(filename, std::ios::binary | std::ios::out);
	file << "P6 " << width << ' ' << height << " 255\n";
	std::vector<char> rgb;
	rgb.reserve(pixels.size() / 4 * 3);
	for (size_t i = 0; i + 3 < pixels.size(); i += 4) { // bytes are A,R,G,B
		rgb.push_back((char)pixels[i + 1]);
		rgb.push_back((char)pixels[i + 2]);
		rgb.push_back((char)pixels[i + 3]);
	}
	file.write(rgb.data(), (std::streamsize)rgb.size());
}

inline std::optional<ModelPair> load_stl_model(const fs::path &filename, std::vector<Triangle> &triangles) {
	std::ifstream file(filename, std::ios::binary | std::ios::in);
	if (file.fail()) return std::nullopt;

	unsigned char header[84]; // 80 bytes of text + little-endian u32 facet count
	file.read(reinterpret_cast<char *>(header), sizeof header);
	if (file.gcount() != (std::streamsize)sizeof header) return ModelPair{(unsigned)triangles.size(), 0u};
	uint32_t count;
	std::memcpy(&count, header + 80, 4);

	const unsigned first = (unsigned)triangles.size();
	unsigned loaded = 0;
	for (uint32_t i = 0; i < count; i++) {
		unsigned char rec[50]; // 12 floats (normal, v1, v2, v3) + u16 attribute
		file.read(reinterpret_cast<char *>(rec), sizeof rec);
		if (file.gcount() != (std::streamsize)sizeof rec) break;
		float f[12];
		std::memcpy(f, rec, sizeof f);
		triangles.push_back(Triangle(glm::vec3(f[0], f[1], f[2]), glm::vec3(f[3], f[4], f[5]), glm::vec3(f[6], f[7], f[8]),
		                             glm::vec3(f[9], f[10], f[11])));
		loaded++;
	}
	return ModelPair{first, loaded};
}

inline std::optional<ModelPair> load_obj_model(const fs::path filename, std::vector<Triangle> &triangles) {
	std::ifstream file(filename, std::ios::in);
	if (file.fail()) return std::nullopt;

	struct Corner {
		long v = 0, n = 0; // 1-based / negative as written; n == 0: no normal given
	};
	std::vector<glm::vec3> positions, normals;
	std::vector<Corner> corners; // 3 per face

	std::string line;
	while (std::getline(file, line)) {
		std::istringstream in(line);
		std::string tag;
		in >> tag;
		if (tag == "v") {
			float x = 0, y = 0, z = 0;
			in >> x >> y >> z;
			positions.push_back(glm::vec3(x, y, z));
		} else if (tag == "vn") {
			float x = 0, y = 0, z = 0;
			in >> x >> y >> z;
			normals.push_back(glm::normalize(glm::vec3(x, y, z)));
		} else if (tag == "f") {
			std::string tok;
			int got = 0;
			Corner c[3];
			while (got < 3 && (in >> tok)) {
				// forms: v | v/vt | v//vn | v/vt/vn
				long v = 0, vt = 0, vn = 0;
				if (std::sscanf(tok.c_str(), "%ld/%ld/%ld", &v, &vt, &vn) == 3) {
				} else if (std::sscanf(tok.c_str(), "%ld//%ld", &v, &vn) == 2) {
				} else if (std::sscanf(tok.c_str(), "%ld/%ld", &v, &vt) == 2) {
					vn = 0;
				} else if (std::sscanf(tok.c_str(), "%ld", &v) == 1) {
					vn = 0;
				} else {
					break;
				}
				c[got].v = v;
				c[got].n = vn;
				got++;
			}
			if (got == 3)
				for (auto &k : c) corners.push_back(k);
		}
		// '#', 's', 'o', 'g', 'vt', 'usemtl', ... are ignored
	}

	auto resolve = [](long index, size_t len) -> long { // -> 0-based, or -1 when invalid
		if (index > 0) return index <= (long)len ? index - 1 : -1;
		if (index < 0) return (long)len + index >= 0 ? (long)len + index : -1;
		return -1;
	};

	const unsigned first = (unsigned)triangles.size();
	for (size_t f = 0; f + 2 < corners.size(); f += 3) {
		Triangle t;
		bool have_normals = true;
		for (int i = 0; i < 3; i++) {
			long vi = resolve(corners[f + i].v, positions.size());
			if (vi < 0) {
				triangles.resize(first);
				return std::nullopt;
			}
			t.vertices[i].pos = positions[(size_t)vi];
			if (corners[f + i].n == 0) {
				have_normals = false;
			} else {
				long ni = resolve(corners[f + i].n, normals.size());
				if (ni < 0) {
					triangles.resize(first);
					return std::nullopt;
				}
				t.vertices[i].normal = normals[(size_t)ni];
			}
		}
		if (!have_normals) {
			glm::vec3 n = glm::cross(t.vertices[1].pos - t.vertices[0].pos, t.vertices[2].pos - t.vertices[0].pos);
			float len2 = glm::dot(n, n);
			n = len2 > 0.0f ? glm::normalize(n) : glm::vec3(0.0f);
			for (auto &v : t.vertices) v.normal = n;
		}
		triangles.push_back(t);
	}
	return ModelPair{first, (unsigned)(triangles.size() - first)};
}
