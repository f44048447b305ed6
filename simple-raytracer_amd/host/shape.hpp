// shape.hpp — same interface and byte layouts as the reference's include/shape.hpp /
// src/shape.cpp (Sphere 32 B, Plane 32 B, Triangle 96 B with normal BEFORE pos, Model
// 112 B with a column-major transform, Shape 128 B tagged union). Header-only.
#pragma once

#include <cmath>
#include <cstddef>
#include <limits>
#include <stdexcept>
#include <vector>

#include "../../include/srt_types.h"
#include "srt_compat.hpp"

struct Sphere {
	alignas(cl_float3) glm::vec3 position;
	alignas(cl_float3) float radius;

	Sphere(const glm::vec3 &position, float radius) : position(position), radius(radius) {}
};

struct Plane {
	glm::vec3 position;
	alignas(cl_float3) glm::vec3 normal;

	Plane(const glm::vec3 &position, const glm::vec3 &normal) : position(position), normal(normal) {}
};

struct Triangle {
	struct Vertex {
		alignas(cl_float3) glm::vec3 normal;
		alignas(cl_float3) glm::vec3 pos;
	};

	Vertex vertices[3];

	/// every field zero
	Triangle() {
		for (auto &v : vertices) {
			v.normal = glm::vec3(0.0f);
			v.pos = glm::vec3(0.0f);
		}
	}

	/// flat shaded: one normal for the three corners
	Triangle(glm::vec3 normal, glm::vec3 pos0, glm::vec3 pos1, glm::vec3 pos2) {
		const glm::vec3 p[3] = {pos0, pos1, pos2};
		for (int i = 0; i < 3; i++) {
			vertices[i].normal = normal;
			vertices[i].pos = p[i];
		}
	}

	Triangle(Vertex v0, Vertex v1, Vertex v2) {
		vertices[0] = v0;
		vertices[1] = v1;
		vertices[2] = v2;
	}
};

/// point (w = 1) or direction (w = 0) through a mat4, glm association
inline glm::vec3 transform_vec3(const glm::mat4 &m, const glm::vec3 &v, bool translate) {
	glm::vec4 r = m * glm::vec4(v, translate ? 1.0f : 0.0f);
	return glm::vec3(r.x, r.y, r.z);
}

/// A range of the shared triangle array plus a transform; bounds are WORLD space.
struct Model {
	cl_uint triangle_index;
	cl_uint num_triangles;
	alignas(cl_float3) glm::vec3 bounding_min;
	alignas(cl_float3) glm::vec3 bounding_max;
	alignas(cl_float3) glm::mat4 transform;

	Model() {}

	/// identity transform, bounds computed from the triangles
	Model(const std::vector<Triangle> &triangles, cl_uint triangle_index, cl_uint num_triangles)
		: triangle_index(triangle_index), num_triangles(num_triangles), transform(1.0f) {
		compute_bounding_box(triangles);
	}

	/// world AABB = componentwise min/max of transform * vertex over the model's range
	void compute_bounding_box(const std::vector<Triangle> &triangles) {
		const float inf = std::numeric_limits<float>::infinity();
		bounding_min = glm::vec3(inf);
		bounding_max = glm::vec3(-inf);
		for (cl_uint i = 0; i < num_triangles; i++) {
			const Triangle &t = triangles[triangle_index + i];
			for (const auto &vertex : t.vertices) {
				const glm::vec3 w = transform_vec3(transform, vertex.pos, true);
				bounding_min = glm::min(bounding_min, w);
				bounding_max = glm::max(bounding_max, w);
			}
		}
	}
};

struct Box {
	static inline int triangle_index = -1;

	/// appends the 12 triangles of the [-1,1]^3 cube (flat normals pointing outward)
	static void create_triangle(std::vector<Triangle> &triangles) {
		static const float corner[8][3] = {{-1, -1, 1}, {-1, 1, 1}, {-1, -1, -1}, {-1, 1, -1},
		                                   {1, -1, 1},  {1, 1, 1},  {1, -1, -1},  {1, 1, -1}};
		static const int face[12][3] = {{1, 2, 0}, {3, 6, 2}, {7, 4, 6}, {5, 0, 4}, {6, 0, 2}, {3, 5, 7},
		                                {1, 3, 2}, {3, 7, 6}, {7, 5, 4}, {5, 1, 0}, {6, 4, 0}, {3, 1, 5}};
		Box::triangle_index = (int)triangles.size();
		for (const auto &f : face) {
			const glm::vec3 a(corner[f[0]][0], corner[f[0]][1], corner[f[0]][2]);
			const glm::vec3 b(corner[f[1]][0], corner[f[1]][1], corner[f[1]][2]);
			const glm::vec3 c(corner[f[2]][0], corner[f[2]][1], corner[f[2]][2]);
			glm::vec3 n = glm::cross(b - a, c - a);
			if (!(glm::dot(a, n) > 0.0f)) n = n * -1.0f; // away from the cube centre
			triangles.push_back(Triangle(glm::normalize(n), a, b, c));
		}
	}

	/// an instance of the shared cube: transform = translate(position), bounds =
	/// position -+ size/2 (consistent with the triangles only for size 2, which is what
	/// the reference's UI passes)
	static Model model(const glm::vec3 &position, const glm::vec3 &size) {
		if (Box::triangle_index == -1) throw std::runtime_error("uninitialized box model, you forgot to call Box::create_triangle");
		Model m;
		m.triangle_index = (cl_uint)Box::triangle_index;
		m.num_triangles = 12;
		m.bounding_min = position - size * 0.5f;
		m.bounding_max = position + size * 0.5f;
		m.transform = glm::translate(position);
		return m;
	}
};

enum ShapeType { SHAPE_SPHERE, SHAPE_PLANE, SHAPE_MODEL };

struct Shape {
	ShapeType type;
	cl_int material;
	union U {
		Sphere sphere;
		Plane plane;
		Model model;
		U() {}
	} shape;

	Shape(cl_int material_index, const Sphere &s) : type(SHAPE_SPHERE), material(material_index) { shape.sphere = s; }
	Shape(cl_int material_index, const Plane &p) : type(SHAPE_PLANE), material(material_index) { shape.plane = p; }
	Shape(cl_int material_index, const Model &m) : type(SHAPE_MODEL), material(material_index) { shape.model = m; }
};

static_assert(sizeof(Sphere) == sizeof(srt_sphere) && offsetof(Sphere, radius) == offsetof(srt_sphere, radius), "Sphere layout");
static_assert(sizeof(Plane) == sizeof(srt_plane) && offsetof(Plane, normal) == offsetof(srt_plane, normal), "Plane layout");
static_assert(sizeof(Triangle) == sizeof(srt_triangle) && offsetof(Triangle::Vertex, pos) == offsetof(srt_vertex, pos), "Triangle layout");
static_assert(sizeof(Model) == sizeof(srt_model) && offsetof(Model, transform) == offsetof(srt_model, transform) &&
                  offsetof(Model, bounding_min) == offsetof(srt_model, bounding_min) &&
                  offsetof(Model, bounding_max) == offsetof(srt_model, bounding_max),
              "Model layout");
static_assert(sizeof(Shape) == sizeof(srt_shape) && offsetof(Shape, shape) == offsetof(srt_shape, shape) &&
                  offsetof(Shape, material) == offsetof(srt_shape, material),
              "Shape layout");
