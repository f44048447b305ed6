// srt_headless — drives the mirrored C++ host API (host/tracer.hpp) the way the
// reference's src/main.cpp drives its Tracer, without SDL/ImGui: build a scene with
// Shape/Material/Box/Model, optionally load OBJ/STL models, render N progressive
// frames, write a PPM (save_ppm) and, for tests, raw dumps of inputs and canvas.
//
//   srt_headless [--scene spheres|meshes|empty] [--obj f.obj]... [--stl f.stl]...
//                [--width W --height H --spp S --bounces B --frames N --time T]
//                [--out frame.ppm] [--dump prefix] [--parse-only] [--bvh] [--gpus N] [--pipelined] [--skybox sky.ppm]
//
// --skybox sky.ppm: an 8-bit binary PPM (P6) as the sky, prepared the way the reference prepares assets/skybox.png
//                   (host/skybox.hpp: four channels, rows flipped, pow(byte / 255, 2.2)); default: the synthetic sky.
//
// --dump prefix writes prefix.{shapes,tris,mats,rd,sd,canvas,argb}.bin (raw records).
// --parse-only skips everything that needs a GPU (loaders + scene construction only).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../host/parser.hpp"
#include "../host/skybox.hpp"
#include "../host/tracer.hpp"

// binary PPM (P6, maxval 255; comments allowed in the header) -> RGB bytes, first row = top of the picture
static bool load_ppm(const std::string &path, std::vector<uint8_t> &rgb, int &w, int &h) {
	std::ifstream f(path, std::ios::binary);
	if (!f) return false;
	std::string magic;
	f >> magic;
	if (magic != "P6") return false;
	int vals[3], got = 0;
	while (got < 3 && f) {
		f >> std::ws;
		if (f.peek() == '#') {
			std::string line;
			std::getline(f, line);
			continue;
		}
		f >> vals[got++];
	}
	if (got < 3 || vals[0] <= 0 || vals[1] <= 0 || vals[2] != 255) return false;
	f.get(); // the single whitespace byte behind maxval
	w = vals[0], h = vals[1];
	rgb.resize((size_t)w * h * 3);
	f.read(reinterpret_cast<char *>(rgb.data()), (std::streamsize)rgb.size());
	return (size_t)f.gcount() == rgb.size();
}

// same integer formula as simple-raytracer_amd/scenes.py synthetic_sky()
static std::vector<float> synthetic_sky(int w, int h) {
	std::vector<float> out((size_t)w * h * 4);
	const long horizon = h / 2;
	auto clampi = [](long v) { return v < 0 ? 0L : (v > 255 ? 255L : v); };
	for (long y = 0; y < h; y++)
		for (long x = 0; x < w; x++) {
			long up = y - horizon > 0 ? y - horizon : 0;
			long cloud = (((x * 5 + y * 3) >> 5) ^ ((x * 3 - y * 7) >> 6)) & 31;
			long base = y < horizon ? 70 + (y * 40) / (horizon > 1 ? horizon : 1) : 230 - (up * 120) / ((h - horizon) > 1 ? (h - horizon) : 1);
			long r = clampi(base - 40 + cloud), g = clampi(base - 10 + cloud);
			long b = y < horizon ? clampi(base - 20 + cloud) : clampi(base + 20 + cloud / 2);
			float *t = &out[((size_t)y * w + x) * 4];
			t[0] = (float)(r * r) / 65025.0f;
			t[1] = (float)(g * g) / 65025.0f;
			t[2] = (float)(b * b) / 65025.0f;
			t[3] = 1.0f;
		}
	return out;
}

// The front-end's camera convention: camera_to_world carries the eye position in
// column 3 and the yaw-then-pitch rotation in the upper 3x3.
static glm::mat4 eye_matrix(const glm::vec3 &eye, float yaw, float pitch) {
	return glm::translate(eye) * glm::eulerAngleYXZ(yaw, pitch, 0.0f);
}

// the scene's material table; the names only matter to a GUI, so the tool keeps none
struct MaterialTable {
	std::vector<Material> list;
	int add(const Material &m) {
		list.push_back(m);
		return (int)list.size() - 1;
	}
	int size() const { return (int)list.size(); }
};

template <class T>
static void dump(const std::string &path, const T *data, size_t count) {
	std::ofstream f(path, std::ios::binary);
	f.write(reinterpret_cast<const char *>(data), (std::streamsize)(count * sizeof(T)));
}

int main(int argc, char **argv) {
	std::string scene = "spheres", out, dump_prefix, skybox_path;
	std::vector<std::string> objs, stls;
	int width = 256, height = 256, spp = 16, bounces = 10, frames = 1;
	unsigned time_seed = 12345;
	bool parse_only = false, bvh = false, pipelined = false;
	int gpus = 1;
	for (int i = 1; i < argc; i++) {
		std::string a = argv[i];
		auto next = [&]() -> const char * {
			if (i + 1 >= argc) {
				std::cerr << "missing value for " << a << "\n";
				std::exit(2);
			}
			return argv[++i];
		};
		if (a == "--scene") scene = next();
		else if (a == "--obj") objs.push_back(next());
		else if (a == "--stl") stls.push_back(next());
		else if (a == "--width") width = std::atoi(next());
		else if (a == "--height") height = std::atoi(next());
		else if (a == "--spp") spp = std::atoi(next());
		else if (a == "--bounces") bounces = std::atoi(next());
		else if (a == "--frames") frames = std::atoi(next());
		else if (a == "--time") time_seed = (unsigned)std::strtoul(next(), nullptr, 10);
		else if (a == "--out") out = next();
		else if (a == "--dump") dump_prefix = next();
		else if (a == "--parse-only") parse_only = true;
		else if (a == "--bvh") bvh = true;
		else if (a == "--gpus") gpus = std::atoi(next());
		else if (a == "--pipelined") pipelined = true;
		else if (a == "--skybox") skybox_path = next();
		else {
			std::cerr << "usage: srt_headless [--scene spheres|meshes|empty] [--obj f]... [--stl f]... [--width W --height H --spp S "
			             "--bounces B --frames N --time T] [--out f.ppm] [--dump prefix] [--parse-only] [--bvh] [--gpus N] [--pipelined] [--skybox sky.ppm]\n";
			return 2;
		}
	}

	// ---- scene construction, as src/main.cpp:95-126 does it ----
	std::vector<Shape> shapes;
	std::vector<Triangle> triangles;
	MaterialTable materials;
	Box::create_triangle(triangles);

	if (scene == "spheres") {
		materials.add(Material(Color(0.8f, 0.8f, 0.9f)));
		materials.add(Material(Color(0.9f, 0.3f, 0.3f)));
		materials.add(Material(Color(0.3f, 0.9f, 0.4f)));
		materials.add(Material(Color(0.9f, 0.95f, 1.0f)));
		materials.add(Material(color::white, 1.0f, 0.0f, 0.0f, 1.0f, 1.5f));
		materials.add(Material(Color(0.2f, 0.3f, 0.9f), 1.0f, 1.0f));
		materials.add(Material(color::white, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, Color(1.0f, 0.2f, 0.2f), 5.0f));
		shapes.push_back(Shape(0, Plane({0, -1, 0}, {0, 1, 0})));
		shapes.push_back(Shape(1, Plane({-4, 0, 0}, {1, 0, 0})));
		shapes.push_back(Shape(2, Plane({0, 0, -6}, {0, 0, 1})));
		shapes.push_back(Shape(3, Sphere({-2, 0, -1}, 1.5f)));
		shapes.push_back(Shape(4, Sphere({0.5f, 0.8f, -3}, 1.0f)));
		shapes.push_back(Shape(5, Sphere({2.5f, 0.5f, -1.5f}, 1.0f)));
		shapes.push_back(Shape(6, Sphere({0.6f, -1, 0}, 0.6f)));
	} else if (scene == "meshes") {
		materials.add(Material(Color(0.8f, 0.8f, 0.8f)));
		materials.add(Material(Color(0.4f, 0.9f, 0.5f), 1.0f, 0.0f, 0.0f, 1.0f, 1.3f));
		materials.add(Material(Color(0.9f, 0.7f, 0.3f), 0.7f, 0.6f));
		shapes.push_back(Shape(0, Plane({0, -1.2f, 0}, {0, 1, 0})));
		shapes.push_back(Shape(2, Box::model({2.5f, -0.2f, -3.0f}, {2, 2, 2})));
	} else {
		materials.add(Material()); // the app's start-up state (main.cpp:100)
	}

	int slot = 0;
	auto add_model = [&](const std::optional<ModelPair> &pair, const std::string &path) {
		if (!pair) {
			std::cerr << "Inexistant file: " << path << "\n";
			std::exit(3);
		}
		Model m(triangles, pair->first, pair->second);
		// place successive models side by side, like dragging them apart in the GUI
		m.transform = glm::translate(glm::vec3(-1.3f + 2.7f * (float)slot, 0.1f, -1.0f - 0.6f * (float)slot)) *
		              glm::eulerAngleYXZ(0.6f - 1.5f * (float)slot, 0.2f, 0.0f);
		m.compute_bounding_box(triangles);
		shapes.push_back(Shape(materials.size() > 1 ? 1 + slot % (materials.size() - 1) : 0, m));
		slot++;
		std::cout << path << ": " << pair->second << " triangles at " << pair->first << "\n";
	};
	for (auto &p : objs) add_model(load_obj_model(p, triangles), p);
	for (auto &p : stls) add_model(load_stl_model(p, triangles), p);

	const glm::mat4 camera_to_world = eye_matrix(glm::vec3(0.0f, 0.5f, 5.0f), 0.0f, 0.0f);

	if (!dump_prefix.empty()) {
		dump(dump_prefix + ".shapes.bin", shapes.data(), shapes.size());
		dump(dump_prefix + ".tris.bin", triangles.data(), triangles.size());
		dump(dump_prefix + ".mats.bin", materials.list.data(), materials.list.size());
	}
	if (parse_only) return 0;

	// ---- tracer set-up, as src/main.cpp:114-126 ----
	Tracer tracer(width, height, 0, gpus); // --gpus N: one Tracer over N devices (rows split, one RCCL gather per frame)
	if (bvh) tracer.set_acceleration(SRT_ACCEL_BVH);
	tracer.options.num_samples = spp;
	tracer.options.num_bounces = bounces;
	tracer.options.show_normals = false;
	tracer.scene_data.horizon_color = color::from_hex(0x374F62);
	tracer.scene_data.zenith_color = color::from_hex(0x11334A);
	tracer.scene_data.ground_color = color::from_hex(0x777777);
	tracer.scene_data.sun_focus = 25.0f;
	tracer.scene_data.sun_color = color::from_hex(0xffffd3);
	tracer.scene_data.sun_intensity = 1.0f;
	tracer.scene_data.sun_direction = VEC3TOCL(glm::normalize(glm::vec3(1.0f, -1.0f, 0.0f)));
	int sky_w = 2048, sky_h = 1024;
	std::vector<float> sky;
	if (!skybox_path.empty()) {
		std::vector<uint8_t> rgb;
		if (!load_ppm(skybox_path, rgb, sky_w, sky_h)) {
			std::cerr << "cannot read " << skybox_path << " (binary PPM, P6, maxval 255)\n";
			return 3;
		}
		sky.resize((size_t)sky_w * sky_h * 4);
		srt_skybox_from_rgb8(rgb.data(), sky_w, sky_h, 3, sky.data()); // as stbi_loadf + flip (src/tracer.cpp:42-46)
	} else {
		sky = synthetic_sky(sky_w, sky_h);
	}
	tracer.set_skybox(sky.data(), sky_w, sky_h);
	if (!dump_prefix.empty()) dump(dump_prefix + ".sky.bin", sky.data(), sky.size());

	std::vector<uint8_t> pixels((size_t)width * height * 4);

	// ---- frame loop, as src/main.cpp:277-290,337 ----
	unsigned time_not_moved = 1;
	auto t0 = std::chrono::steady_clock::now();
	for (int frame = 0; frame < frames; frame++) {
		if (time_not_moved == 1) {
			tracer.clear_canvas();
			tracer.update_scene(shapes, triangles, materials.list);
		}
		auto &options = tracer.options;
		options.aspect_ratio = (float)width / (float)height;
		options.fov_scale = 1.0f; // tan(90deg / 2)
		options.camera_to_world = camera_to_world;
		options.time = time_seed + 7919u * (unsigned)frame;
		options.tick = (unsigned)frame;
		if (pipelined) tracer.render_pipelined(time_not_moved, pixels); // delivers the previous frame
		else tracer.render(time_not_moved, pixels);
		time_not_moved++;
	}
	if (pipelined) tracer.finish(pixels);
	double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	srt_counters c = tracer.counters();
	std::printf("%d frame(s) %dx%d x %d spp: %.3f s, %.1f Mray/s, %llu NaN pixels\n", frames, width, height, spp, secs,
	            (double)c.rays / secs / 1e6, (unsigned long long)c.nan_pixels);

	if (!out.empty()) save_ppm(out, pixels, width, height);
	if (!dump_prefix.empty()) {
		std::vector<float> canvas;
		tracer.read_canvas(canvas);
		dump(dump_prefix + ".canvas.bin", canvas.data(), canvas.size());
		dump(dump_prefix + ".argb.bin", pixels.data(), pixels.size());
		dump(dump_prefix + ".rd.bin", &tracer.options, 1);
		dump(dump_prefix + ".sd.bin", &tracer.scene_data, 1);
	}
	return 0;
}
