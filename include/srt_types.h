/* srt_types.h — plain-C scene/record layouts crossing the C ABI.
 *
 * These are byte-for-byte the device structs of the reference kernel
 * (/root/reference/src/render.cl:5-105) and of their host mirrors
 * (/root/reference/include/material.hpp:10-38, include/shape.hpp:15-111,
 * include/tracer.hpp:48-80). OpenCL float3 occupies 16 bytes, so every
 * 3-vector below carries one pad float. No glm, boost or OpenCL headers needed.
 */
#ifndef SRT_TYPES_H
#define SRT_TYPES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct srt_float3 {
	float x, y, z, _pad;
} srt_float3; /* cl_float3 */

typedef struct srt_float4 {
	float x, y, z, w;
} srt_float4; /* cl_float4 */

/* render.cl:17-27 / material.hpp:10-21 */
typedef struct srt_material {
	float smoothness;
	float metallic;
	float specular;
	float emission_strength;
	float transmittance;
	float refraction_index;
	float _pad[2];
	srt_float3 color;
	srt_float3 emission;
} srt_material;

/* render.cl:29-32 / shape.hpp:15-20 */
typedef struct srt_sphere {
	srt_float3 position;
	float radius;
	float _pad[3];
} srt_sphere;

/* render.cl:34-37 / shape.hpp:22-27 */
typedef struct srt_plane {
	srt_float3 position;
	srt_float3 normal;
} srt_plane;

/* render.cl:39-42 / shape.hpp:30-33 — note: normal FIRST, then position */
typedef struct srt_vertex {
	srt_float3 normal;
	srt_float3 pos;
} srt_vertex;

/* render.cl:44-53 / shape.hpp:29-44 */
typedef struct srt_triangle {
	srt_vertex vertices[3];
} srt_triangle;

/* render.cl:55-61 / shape.hpp:47-68; transform = 4 COLUMNS (glm::mat4) */
typedef struct srt_model {
	uint32_t triangle_index;
	uint32_t num_triangles;
	uint32_t _pad[2];
	srt_float3 bounding_min;
	srt_float3 bounding_max;
	srt_float4 transform[4];
} srt_model;

/* render.cl:63-67 / shape.hpp:78-82 */
enum { SRT_SHAPE_SPHERE = 0, SRT_SHAPE_PLANE = 1, SRT_SHAPE_MODEL = 2 };

/* render.cl:69-77 / shape.hpp:84-111 */
typedef struct srt_shape {
	int32_t type;
	int32_t material;
	int32_t _pad[2];
	union {
		srt_sphere sphere;
		srt_plane plane;
		srt_model model;
	} shape;
} srt_shape;

/* One node of the binary hierarchy the host builds over a model's triangles (srt_set_acceleration
 * in srt_abi.h; new, no counterpart in the reference) and srt_bvh_build_host hands out. Nodes of a
 * model are stored in depth-first order: an inner node's first child is node + 1, its second child
 * that child's `skip`. The device walks a four-wide folding of it (srt_bvh_wide_host). */
typedef struct srt_bvh_node {
	float lo[3];
	uint32_t skip; /* first node after this node's subtree, or SRT_BVH_END */
	float hi[3];
	uint32_t leaf; /* 0: inner node; else (count << 28) | first triangle record */
} srt_bvh_node;
#define SRT_BVH_END 0xffffffffu

/* render.cl:79-92 / tracer.hpp:48-67 */
typedef struct srt_render_data {
	int32_t width, height;
	int32_t num_samples;
	int32_t num_bounces;
	float aspect_ratio;
	float fov_scale;
	uint8_t show_normals;
	uint8_t _pad0[7];
	srt_float4 camera_to_world[4]; /* columns; column 3 = camera position */
	uint32_t time;
	uint32_t tick; /* never read by the kernel */
	uint32_t _pad1[2];
} srt_render_data;

/* render.cl:94-105 / tracer.hpp:69-80 */
typedef struct srt_scene_data {
	int32_t num_shapes;
	float sun_focus;
	float sun_intensity;
	float _pad0;
	srt_float3 horizon_color; /* unused by the kernel (render.cl:381-389) */
	srt_float3 zenith_color;  /* unused */
	srt_float3 ground_color;  /* unused */
	srt_float3 sun_color;
	srt_float3 sun_direction;
} srt_scene_data;

#ifdef __cplusplus
}
#endif

#if defined(__cplusplus)
#define SRT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define SRT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

SRT_STATIC_ASSERT(sizeof(srt_float3) == 16, "float3 is 16 B");
SRT_STATIC_ASSERT(sizeof(srt_material) == 64, "Material 64 B");
SRT_STATIC_ASSERT(offsetof(srt_material, transmittance) == 16, "Material.transmittance@16");
SRT_STATIC_ASSERT(offsetof(srt_material, refraction_index) == 20, "Material.refraction_index@20");
SRT_STATIC_ASSERT(offsetof(srt_material, color) == 32, "Material.color@32");
SRT_STATIC_ASSERT(offsetof(srt_material, emission) == 48, "Material.emission@48");
SRT_STATIC_ASSERT(sizeof(srt_sphere) == 32, "Sphere 32 B");
SRT_STATIC_ASSERT(offsetof(srt_sphere, radius) == 16, "Sphere.radius@16");
SRT_STATIC_ASSERT(sizeof(srt_plane) == 32, "Plane 32 B");
SRT_STATIC_ASSERT(offsetof(srt_plane, normal) == 16, "Plane.normal@16");
SRT_STATIC_ASSERT(sizeof(srt_vertex) == 32, "Vertex 32 B");
SRT_STATIC_ASSERT(offsetof(srt_vertex, pos) == 16, "Vertex.pos@16");
SRT_STATIC_ASSERT(sizeof(srt_triangle) == 96, "Triangle 96 B");
SRT_STATIC_ASSERT(sizeof(srt_model) == 112, "Model 112 B");
SRT_STATIC_ASSERT(offsetof(srt_model, num_triangles) == 4, "Model.num_triangles@4");
SRT_STATIC_ASSERT(offsetof(srt_model, bounding_min) == 16, "Model.bounding_min@16");
SRT_STATIC_ASSERT(offsetof(srt_model, bounding_max) == 32, "Model.bounding_max@32");
SRT_STATIC_ASSERT(offsetof(srt_model, transform) == 48, "Model.transform@48");
SRT_STATIC_ASSERT(sizeof(srt_shape) == 128, "Shape 128 B");
SRT_STATIC_ASSERT(offsetof(srt_shape, material) == 4, "Shape.material@4");
SRT_STATIC_ASSERT(offsetof(srt_shape, shape) == 16, "Shape.union@16");
SRT_STATIC_ASSERT(sizeof(srt_bvh_node) == 32, "BVH node 32 B");
SRT_STATIC_ASSERT(sizeof(srt_render_data) == 112, "RenderData 112 B");
SRT_STATIC_ASSERT(offsetof(srt_render_data, aspect_ratio) == 16, "RenderData.aspect_ratio@16");
SRT_STATIC_ASSERT(offsetof(srt_render_data, show_normals) == 24, "RenderData.show_normals@24");
SRT_STATIC_ASSERT(offsetof(srt_render_data, camera_to_world) == 32, "RenderData.camera_to_world@32");
SRT_STATIC_ASSERT(offsetof(srt_render_data, time) == 96, "RenderData.time@96");
SRT_STATIC_ASSERT(offsetof(srt_render_data, tick) == 100, "RenderData.tick@100");
SRT_STATIC_ASSERT(sizeof(srt_scene_data) == 96, "SceneData 96 B");
SRT_STATIC_ASSERT(offsetof(srt_scene_data, sun_focus) == 4, "SceneData.sun_focus@4");
SRT_STATIC_ASSERT(offsetof(srt_scene_data, horizon_color) == 16, "SceneData.horizon@16");
SRT_STATIC_ASSERT(offsetof(srt_scene_data, sun_color) == 64, "SceneData.sun_color@64");
SRT_STATIC_ASSERT(offsetof(srt_scene_data, sun_direction) == 80, "SceneData.sun_direction@80");

#endif /* SRT_TYPES_H */
