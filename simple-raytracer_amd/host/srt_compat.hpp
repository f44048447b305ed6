// srt_compat.hpp — what the mirrored host headers need from glm and from
// boost/compute/types.hpp, without requiring either.
//
// * With glm installed (the reference's front-end needs it anyway) the real glm types
//   are used, so src/main.cpp / src/interface.cpp compile against these headers as is.
// * Without glm a minimal stand-in provides exactly the subset the mirrored headers
//   use: vec3/vec4/mat4 (column-major, m[col][row]), translate, scale, eulerAngleYXZ,
//   normalize, cross, dot, min, max, mat*vec, mat*mat. It is enough to build scenes and
//   drive the Tracer headlessly (tests, tools); it is not a glm replacement.
// * cl_int / cl_uint / cl_float / cl_float3 / cl_float4: layout-compatible stand-ins
//   (cl_float3 is 16 bytes with .x .y .z and .s[4], brace-initialisable as
//   `cl_float3({{x, y, z}})` like VEC3TOCL does, include/tracer.hpp:23).
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__has_include)
#if __has_include(<glm/glm.hpp>) && !defined(SRT_FORCE_MINI_GLM)
#define SRT_HAVE_GLM 1
#endif
#endif

#ifdef SRT_HAVE_GLM
#define GLM_FORCE_SWIZZLE
#define GLM_ENABLE_EXPERIMENTAL
#include <glm/glm.hpp>
#include <glm/gtc/type_ptr.hpp>
#include <glm/gtx/euler_angles.hpp>
#include <glm/gtx/transform.hpp>
#else
namespace glm {
struct vec3 {
	float x, y, z;
	vec3() : x(0), y(0), z(0) {}
	explicit vec3(float s) : x(s), y(s), z(s) {}
	vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
	float &operator[](int i) { return (&x)[i]; }
	const float &operator[](int i) const { return (&x)[i]; }
};
struct vec4 {
	float x, y, z, w;
	vec4() : x(0), y(0), z(0), w(0) {}
	vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
	vec4(const vec3 &v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
	float &operator[](int i) { return (&x)[i]; }
	const float &operator[](int i) const { return (&x)[i]; }
};
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 &operator*=(vec3 &a, float s) { return a = a * s; }
inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline vec3 normalize(vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline vec3 min(vec3 a, vec3 b) { return {b.x < a.x ? b.x : a.x, b.y < a.y ? b.y : a.y, b.z < a.z ? b.z : a.z}; }
inline vec3 max(vec3 a, vec3 b) { return {a.x < b.x ? b.x : a.x, a.y < b.y ? b.y : a.y, a.z < b.z ? b.z : a.z}; }
struct mat4 {
	vec4 c[4];
	mat4() : mat4(1.0f) {}
	explicit mat4(float d) {
		c[0] = {d, 0, 0, 0};
		c[1] = {0, d, 0, 0};
		c[2] = {0, 0, d, 0};
		c[3] = {0, 0, 0, d};
	}
	vec4 &operator[](int i) { return c[i]; }
	const vec4 &operator[](int i) const { return c[i]; }
};
// glm's own association: (m0*x + m1*y) + (m2*z + m3*w)
inline vec4 operator*(const mat4 &m, const vec4 &v) { return (m[0] * v.x + m[1] * v.y) + (m[2] * v.z + m[3] * v.w); }
inline mat4 operator*(const mat4 &a, const mat4 &b) {
	mat4 r(0.0f);
	for (int j = 0; j < 4; j++) r[j] = ((a[0] * b[j].x + a[1] * b[j].y) + a[2] * b[j].z) + a[3] * b[j].w;
	return r;
}
inline mat4 &operator*=(mat4 &a, const mat4 &b) { return a = a * b; }
inline mat4 translate(const mat4 &m, const vec3 &v) {
	mat4 r = m;
	r[3] = ((m[0] * v.x + m[1] * v.y) + m[2] * v.z) + m[3];
	return r;
}
inline mat4 translate(const vec3 &v) { return translate(mat4(1.0f), v); }
inline mat4 scale(const vec3 &v) {
	mat4 r(1.0f);
	r[0].x = v.x;
	r[1].y = v.y;
	r[2].z = v.z;
	return r;
}
inline mat4 eulerAngleYXZ(float yaw, float pitch, float roll) {
	float ch = std::cos(yaw), sh = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch), cb = std::cos(roll), sb = std::sin(roll);
	mat4 r(1.0f);
	r[0] = {ch * cb + sh * sp * sb, sb * cp, -sh * cb + ch * sp * sb, 0};
	r[1] = {-ch * sb + sh * sp * cb, cb * cp, sb * sh + ch * sp * cb, 0};
	r[2] = {sh * cp, -sp, ch * cp, 0};
	r[3] = {0, 0, 0, 1};
	return r;
}
inline const float *value_ptr(const mat4 &m) { return &m.c[0].x; }
inline float *value_ptr(mat4 &m) { return &m.c[0].x; }
} // namespace glm
#endif

#if defined(__has_include)
#if __has_include(<boost/compute/types.hpp>) && !defined(SRT_FORCE_MINI_CL)
#define SRT_HAVE_BOOST_COMPUTE_TYPES 1
#endif
#endif

#ifdef SRT_HAVE_BOOST_COMPUTE_TYPES
#define CL_TARGET_OPENCL_VERSION 200
#include <boost/compute/types.hpp>
#else
typedef int32_t cl_int;
typedef uint32_t cl_uint;
typedef float cl_float;
typedef uint8_t cl_uchar;
union alignas(16) cl_float4 {
	float s[4];
	struct {
		float x, y, z, w;
	};
};
typedef cl_float4 cl_float3; // as in CL/cl_platform.h: float3 is a float4
union alignas(4) cl_uchar4 {
	cl_uchar s[4];
	struct {
		cl_uchar x, y, z, w;
	};
};
#endif

static_assert(sizeof(cl_float3) == 16 && alignof(cl_float3) == 16, "cl_float3 is 16 bytes");
