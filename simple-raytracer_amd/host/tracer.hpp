// tracer.hpp — `class Tracer` with the reference's public interface
// (/root/reference/include/tracer.hpp:26-88), implemented over the C ABI of
// libsrt_hip.so instead of boost.compute/OpenCL. The front-end's call sequence
// (src/main.cpp:114-126, 277-290) works unchanged:
//
//     Tracer tracer(w, h);  tracer.options.num_samples = 2; tracer.scene_data.sun_focus = ...
//     tracer.clear_canvas(); tracer.update_scene(shapes, triangles, materials.materials);
//     tracer.options.camera_to_world = camera_mat; ...; tracer.render(ticks, pixels);
//
// Differences a maintainer must know: no run-time files are read
// (render.cl is gone; the skybox is handed over decoded via set_skybox), and failures
// throw std::runtime_error carrying srt_last_error() where boost.compute threw.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/srt_abi.h"
#include "color.hpp"
#include "material.hpp"
#include "shape.hpp"

#define VEC3TOCL(v) (cl_float3({{(v).x, (v).y, (v).z}}))
#define VEC4TOCL(v) (cl_float4({{(v).x, (v).y, (v).z, (v).w}}))

class Tracer {
  private:
	srt_tracer *handle = nullptr; // one device ...
	srt_group *group = nullptr;   // ... or a group of devices (n_devices > 1): rows dealt in interleaved blocks, one RCCL gather per frame

	void check(int rc) const {
		if (rc != SRT_OK) throw std::runtime_error(std::string("srt: ") + (group ? srt_group_last_error(group) : srt_last_error(handle)));
	}

  public:
	struct RenderData {
		cl_int width, height;
		cl_int num_samples;
		cl_int num_bounces;
		cl_float aspect_ratio;
		cl_float fov_scale;
		bool show_normals;

		alignas(cl_float4) glm::mat4 camera_to_world;

		cl_uint time;
		cl_uint tick;

		RenderData(int width, int height)
			: width(width), height(height), num_samples(4), num_bounces(10), aspect_ratio(float(width) / float(height)),
			  fov_scale(1.0f), show_normals(false), camera_to_world(1.0f), time(1), tick(0) {}
	} options;

	struct SceneData {
		cl_int num_shapes;
		cl_float sun_focus;
		cl_float sun_intensity;

		alignas(cl_float3) Color horizon_color;
		alignas(cl_float3) Color zenith_color;
		alignas(cl_float3) Color ground_color;
		alignas(cl_float3) Color sun_color;

		cl_float3 sun_direction;
	} scene_data;

	/// n_devices == 1: the reference's single-device Tracer on HIP device `device`. n_devices > 1 (new): the same
	/// interface over devices 0 .. n_devices-1 of this node; every call below fans out, render() collects the
	/// frame with ONE ncclGather on device 0 and returns the same bytes a single device would.
	Tracer(const int width, const int height, const int device = 0, const int n_devices = 1) : options(width, height), scene_data() {
		int rc = n_devices > 1 ? srt_group_create(width, height, n_devices, nullptr, 8, &group) : srt_create(width, height, device, &handle);
		if (rc != SRT_OK) throw std::runtime_error(std::string("srt_create: ") + srt_last_error(nullptr));
	}
	~Tracer() {
		srt_group_destroy(group);
		srt_destroy(handle);
	}
	Tracer(const Tracer &) = delete;
	Tracer &operator=(const Tracer &) = delete;

	/// replaces the stbi_loadf + enqueue_write_image of the reference's constructor:
	/// `rgba` = width*height RGBA32F texels, row 0 = bottom of the picture
	void set_skybox(const float *rgba, int width, int height) {
		check(group ? srt_group_set_skybox(group, rgba, width, height) : srt_set_skybox(handle, rgba, width, height));
	}

	void update_scene(
		const std::vector<Shape> &shapes, const std::vector<Triangle> &triangles, const std::vector<Material> &materials
	) {
		static_assert(sizeof(SceneData) == sizeof(srt_scene_data), "SceneData layout");
		const srt_shape *sh = reinterpret_cast<const srt_shape *>(shapes.data());
		const srt_triangle *tr = reinterpret_cast<const srt_triangle *>(triangles.data());
		const srt_material *ma = reinterpret_cast<const srt_material *>(materials.data());
		const srt_scene_data *sd = reinterpret_cast<const srt_scene_data *>(&scene_data);
		check(group ? srt_group_update_scene(group, sh, shapes.size(), tr, triangles.size(), ma, materials.size(), sd)
		            : srt_update_scene(handle, sh, shapes.size(), tr, triangles.size(), ma, materials.size(), sd));
		scene_data.num_shapes = (cl_int)shapes.size();
	}

	void clear_canvas() { check(group ? srt_group_clear_canvas(group) : srt_clear_canvas(handle)); }

	/// trace + resolve + blocking read-back of width*height*4 bytes (A,R,G,B)
	void render(cl_uint ticks_stopped, std::vector<uint8_t> &output) {
		static_assert(sizeof(RenderData) == sizeof(srt_render_data), "RenderData layout");
		if (output.size() < size_t(options.width) * size_t(options.height) * 4)
			throw std::runtime_error("Tracer::render: output must hold width*height*4 bytes");
		const srt_render_data *rd = reinterpret_cast<const srt_render_data *>(&options);
		check(group ? srt_group_render(group, rd, ticks_stopped, output.data()) : srt_render(handle, rd, ticks_stopped, output.data()));
	}

	/// render() for a loop that can show a frame one call late (src/main.cpp:277-337 blits `output` right after the
	/// call): enqueues this frame and fills `output` with the PREVIOUS one, so frame N's read-back runs under frame
	/// N+1's trace. Returns the index of the frame delivered, -1 on the first call (output untouched); finish()
	/// hands out the last one. Same bytes as render(), one call later. Single device only.
	long long render_pipelined(cl_uint ticks_stopped, std::vector<uint8_t> &output) {
		if (group) throw std::runtime_error("Tracer::render_pipelined: single-device tracers only");
		if (output.size() < size_t(options.width) * size_t(options.height) * 4)
			throw std::runtime_error("Tracer::render_pipelined: output must hold width*height*4 bytes");
		long long delivered = -1;
		check(srt_render_pipelined(handle, reinterpret_cast<const srt_render_data *>(&options), ticks_stopped, output.data(), &delivered));
		return delivered;
	}
	long long finish(std::vector<uint8_t> &output) {
		long long delivered = -1;
		if (!group) check(srt_pipeline_flush(handle, output.data(), &delivered));
		return delivered;
	}

	// -- extras beyond the reference's interface --
	/// SRT_ACCEL_BVH: models get a bounding-volume hierarchy at the next update_scene (the
	/// reference's README.md:41 "future plan"); SRT_ACCEL_NONE (default) keeps the array-order scan
	void set_acceleration(int mode) { check(group ? srt_group_set_acceleration(group, mode) : srt_set_acceleration(handle, mode)); }
	/// render() / render_pipelined() record the kernel timers' events too (off by default: they cost 10-17 us of a 150 us
	/// frame); single-device tracers only
	void set_kernel_timers(bool enable) {
		if (!group) check(srt_set_kernel_timers(handle, enable ? 1 : 0));
	}
	void read_canvas(std::vector<float> &rgba) {
		rgba.resize(size_t(options.width) * size_t(options.height) * 4);
		check(group ? srt_group_read_canvas(group, rgba.data()) : srt_read_canvas(handle, rgba.data()));
	}
	srt_counters counters() {
		srt_counters c;
		check(group ? srt_group_get_counters(group, &c) : srt_get_counters(handle, &c));
		return c;
	}
	srt_tracer *native_handle() { return group ? srt_group_tracer(group, 0) : handle; }
};

static_assert(offsetof(Tracer::RenderData, camera_to_world) == offsetof(srt_render_data, camera_to_world), "RenderData.camera_to_world");
static_assert(offsetof(Tracer::RenderData, time) == offsetof(srt_render_data, time), "RenderData.time");
static_assert(offsetof(Tracer::RenderData, show_normals) == offsetof(srt_render_data, show_normals), "RenderData.show_normals");
static_assert(offsetof(Tracer::SceneData, sun_color) == offsetof(srt_scene_data, sun_color), "SceneData.sun_color");
static_assert(offsetof(Tracer::SceneData, sun_direction) == offsetof(srt_scene_data, sun_direction), "SceneData.sun_direction");
static_assert(offsetof(Tracer::SceneData, horizon_color) == offsetof(srt_scene_data, horizon_color), "SceneData.horizon_color");
