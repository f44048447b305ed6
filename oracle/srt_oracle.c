/* srt_oracle.c — CPU restatement of the reference path-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product path (simple-raytracer_amd/,
 * include/) may include, link, import or execute this file or anything else under
 * oracle/. Allowed users: tests/, __graft_entry__.smoke() and the `cpu_baseline`
 * leg of bench.py — as the checker, never as the thing measured or shipped.
 *
 * What it restates: the OpenCL kernels `render` and `average` of
 * /root/reference/src/render.cl (lines cited at every function), scalar C, one
 * pixel at a time, same operation order, same RNG stream. The OpenCL built-ins
 * (implementation-defined within ULP bounds) are taken from
 * simple-raytracer_amd/csrc/detmath.h — the same definitions the HIP kernels and
 * the reference-kernel shim (oracle/cl_builtins_shim.cpp) use.
 *
 * PARITY UNPINNED at the built-in boundary: the reference has no tests, fixtures or golden
 * vectors, and its program cannot be built in this image. What this file IS checked against:
 * `make -C oracle ref` compiles the reference's own render.cl for x86-64
 * (oracle/_ref/libsrt_ref.so) with the OpenCL built-ins supplied by cl_builtins_shim.cpp from
 * detmath.h; tests/test_oracle_vs_ref.py requires this file to be BIT-IDENTICAL to that build
 * (control flow, RNG stream, draw and operation order are then the reference's own), and
 * tests/golden/ holds vectors generated from it (tests/golden/make_golden.py) so the same check
 * runs where /root/reference is absent. The built-ins themselves are held to the OpenCL ULP
 * bounds against glibc (tests/test_detmath.py) and to statistical agreement with a glibc-backed
 * build of the same object (tests/test_oracle_statistics.py). See DESIGN.md "Oracle and parity".
 *
 * Build: gcc -O2 -ffp-contract=off [-mfma] -fopenmp -fPIC -shared (oracle/Makefile).
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#include "../include/srt_types.h"
#include "../simple-raytracer_amd/csrc/detmath.h"

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
	float x, y, z;
} v3;

/* deterministic work counters: schedule independent */
enum {
	ORC_C_PATHS = 0,
	ORC_C_RAYS,        /* closest_intersection calls = path segments */
	ORC_C_SPHERE,      /* intersect_sphere calls */
	ORC_C_PLANE,       /* intersect_plane calls */
	ORC_C_AABB,        /* intersection_aabb calls */
	ORC_C_TRI,         /* intersect_triangle calls */
	ORC_C_TRI_PASS_U,  /* triangle tests that got past the u check */
	ORC_C_BOUNCE,      /* shaded hits that drew a new direction */
	ORC_C_SKY,         /* sky_box evaluations */
	ORC_C_NAN_PIXELS,  /* pixels whose per-dispatch colour is NaN in any channel */
	ORC_C_COUNT
};

/* ---- vector helpers with the pinned operation order ------------------------ */
static inline v3 V(float x, float y, float z) {
	v3 r = {x, y, z};
	return r;
}
static inline v3 f3(const srt_float3 *p) { return V(p->x, p->y, p->z); }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* the built-ins dot, cross, normalize, mix as detmath.h pins them (FMA forms) */
static inline float vdot(v3 a, v3 b) { return dm_dot3(a.x, a.y, a.z, b.x, b.y, b.z); }
static inline v3 vcross(v3 a, v3 b) {
	return V(dm_cross1(a.y, b.z, a.z, b.y), dm_cross1(a.z, b.x, a.x, b.z), dm_cross1(a.x, b.y, a.y, b.x));
}
static inline v3 vnormalize(v3 a) {
	float v[3] = {a.x, a.y, a.z};
	dm_normalize3(v, v);
	return V(v[0], v[1], v[2]);
}
static inline v3 vmix(v3 x, v3 y, float a) {
	return V(dm_mix(x.x, y.x, a), dm_mix(x.y, y.y, a), dm_mix(x.z, y.z, a));
}

/* render.cl:114-120 — column-major 4x4 times (v, w); xyz of the result.
 * Each row is ((m0*v.x + m1*v.y) + m2*v.z) + m3*w, left associated. */
static inline v3 mat_by_vec(const srt_float4 *m, v3 v, float w) {
	return V(((m[0].x * v.x + m[1].x * v.y) + m[2].x * v.z) + m[3].x * w,
	         ((m[0].y * v.x + m[1].y * v.y) + m[2].y * v.z) + m[3].y * w,
	         ((m[0].z * v.x + m[1].z * v.y) + m[2].z * v.z) + m[3].z * w);
}

/* render.cl:139-141 */
static inline v3 reflect3(v3 v, v3 n) { return vsub(v, vscale(n, 2.0f * vdot(v, n))); }

/* render.cl:143-148 — PCG-RXS-M-XS-32; (float)UINT_MAX == 2^32 */
float orc_random_float(uint32_t *seed) {
	*seed = *seed * 747796405u + 2891336453u;
	uint32_t result = ((*seed >> ((*seed >> 28) + 4)) ^ *seed) * 277803737u;
	result = (result >> 22) ^ result;
	return (float)result / 4294967296.0f;
}

/* render.cl:150-154 */
static inline float random_float_normal(uint32_t *seed) {
	float theta = 6.28318548f * orc_random_float(seed); /* 2 * M_PI_F */
	float rho = dm_sqrtf(-2.0f * dm_logf(orc_random_float(seed)));
	return rho * dm_cosf(theta);
}

/* render.cl:156-158 — components drawn x, then y, then z */
static inline v3 random_direction(uint32_t *seed) {
	float x = random_float_normal(seed);
	float y = random_float_normal(seed);
	float z = random_float_normal(seed);
	return vnormalize(V(x, y, z));
}

/* render.cl:160-163 */
static inline v3 random_direction_hemisphere(v3 normal, uint32_t *seed) {
	v3 dir = random_direction(seed);
	return vscale(dir, dm_sign(vdot(normal, dir)));
}

/* render.cl:173-178 — the literals are double, so this runs in fp64 */
float orc_shlick_reflectance(float mu, float cos_theta) {
	float r0 = (float)((1.0 - (double)mu) / (1.0 + (double)mu));
	r0 = r0 * r0;
	return (float)((double)r0 + (1.0 - (double)r0) * dm_pown_d(1.0 - (double)cos_theta, 5));
}

/* render.cl:180-204 */
int orc_intersect_sphere(const srt_sphere *sphere, const float *o, const float *d, float *t) {
	v3 org = V(o[0], o[1], o[2]), dir = V(d[0], d[1], d[2]);
	v3 rayToCenter = vsub(f3(&sphere->position), org);
	float b = vdot(rayToCenter, dir);
	float c = vdot(rayToCenter, rayToCenter) - sphere->radius * sphere->radius;
	float disc = b * b - c;
	if (disc < 0.0f) return 0;
	*t = b - dm_sqrtf(disc);
	if (*t < 0.0f) {
		*t = b + dm_sqrtf(disc);
		if (*t < 0.0f) return 0;
	}
	return 1;
}

/* render.cl:206-221 */
int orc_intersect_plane(const srt_plane *plane, const float *o, const float *d, float *t) {
	v3 org = V(o[0], o[1], o[2]), dir = V(d[0], d[1], d[2]);
	float denom = vdot(f3(&plane->normal), dir);
	if (dm_fabs(denom) == 0.0f) return 0;
	float tmp = vdot(f3(&plane->normal), vsub(f3(&plane->position), org)) / denom;
	if (tmp < 0.0f) return 0;
	*t = tmp;
	return 1;
}

/* render.cl:223-241 — returns the "shifted" weights (w2, w0, w1) */
static inline v3 barycentric_weights(v3 p0, v3 p1, v3 p2, v3 p) {
	v3 v0 = vsub(p1, p0);
	v3 v1 = vsub(p2, p0);
	v3 v2 = vsub(p, p0);
	float d00 = vdot(v0, v0);
	float d01 = vdot(v0, v1);
	float d11 = vdot(v1, v1);
	float d20 = vdot(v2, v0);
	float d21 = vdot(v2, v1);
	float denom = d00 * d11 - d01 * d01;
	float w0 = (d11 * d20 - d01 * d21) / denom;
	float w1 = (d00 * d21 - d01 * d20) / denom;
	float w2 = 1.0f - w0 - w1;
	return V(w2, w0, w1);
}

/* render.cl:243-275 — Moller-Trumbore, no epsilon, two-sided.
 * pass_u (optional) reports whether the u-range check was passed. */
static inline int intersect_triangle(v3 p0, v3 p1, v3 p2, v3 org, v3 dir, float *t, int *pass_u) {
	v3 edge1 = vsub(p1, p0);
	v3 edge2 = vsub(p2, p0);
	v3 h = vcross(dir, edge2);
	float a = vdot(edge1, h);
	if (a == 0.0f) return 0;
	float f = 1.0f / a;
	v3 s = vsub(org, p0);
	float u = f * vdot(s, h);
	if (u < 0.0f || u > 1.0f) return 0;
	if (pass_u) *pass_u = 1;
	v3 q = vcross(s, edge1);
	float v = f * vdot(dir, q);
	if (v < 0.0f || u + v > 1.0f) return 0;
	*t = f * vdot(edge2, q);
	if (*t > 0.0f) return 1;
	return 0;
}

int orc_intersect_triangle(const float *p0, const float *p1, const float *p2, const float *o, const float *d, float *t) {
	return intersect_triangle(V(p0[0], p0[1], p0[2]), V(p1[0], p1[1], p1[2]), V(p2[0], p2[1], p2[2]),
	                          V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), t, NULL);
}

/* render.cl:279-290 */
int orc_intersection_aabb(const float *bmin, const float *bmax, const float *o, const float *inv_dir, float tmax) {
	float tmin = 0.0f;
	for (int d = 0; d < 3; d++) {
		float t1 = (bmin[d] - o[d]) * inv_dir[d];
		float t2 = (bmax[d] - o[d]) * inv_dir[d];
		tmin = dm_max(tmin, dm_min(t1, t2));
		tmax = dm_min(tmax, dm_max(t1, t2));
	}
	return tmin < tmax;
}

typedef struct {
	const srt_scene_data *data;
	const srt_shape *shapes;
	const srt_triangle *triangles;
	const srt_material *materials;
	const float *sky; /* RGBA32F, row 0 first */
	int sky_w, sky_h;
} scene_t;

typedef struct {
	v3 position, normal;
	int front;
} hit_t;

/* render.cl:293-378 */
static int closest_intersection(const scene_t *scene, v3 org, v3 dir, hit_t *rayhit, uint64_t *ctr) {
	int closest = -1;
	float tmin = DM_INF_F;
	float o[3] = {org.x, org.y, org.z}, d[3] = {dir.x, dir.y, dir.z};
	float inv_dir[3] = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
	ctr[ORC_C_RAYS]++;

	for (int i = 0; i < scene->data->num_shapes; i++) {
		const srt_shape *shape = &scene->shapes[i];
		if (shape->type == SRT_SHAPE_SPHERE) {
			const srt_sphere *sphere = &shape->shape.sphere;
			float t_i;
			ctr[ORC_C_SPHERE]++;
			if (orc_intersect_sphere(sphere, o, d, &t_i)) {
				if (t_i < tmin) {
					tmin = t_i;
					closest = shape->material;
					rayhit->position = vadd(org, vscale(dir, tmin));
					rayhit->normal = vdivs(vsub(rayhit->position, f3(&sphere->position)), sphere->radius);
				}
			}
		} else if (shape->type == SRT_SHAPE_MODEL) {
			const srt_model *model = &shape->shape.model;
			float bmin[3] = {model->bounding_min.x, model->bounding_min.y, model->bounding_min.z};
			float bmax[3] = {model->bounding_max.x, model->bounding_max.y, model->bounding_max.z};
			ctr[ORC_C_AABB]++;
			if (!orc_intersection_aabb(bmin, bmax, o, inv_dir, tmin)) continue;
			for (size_t j = 0; j < model->num_triangles; j++) {
				const srt_triangle *tri = &scene->triangles[model->triangle_index + j];
				/* render.cl:325-328: positions to world space, per ray */
				v3 p0 = mat_by_vec(model->transform, f3(&tri->vertices[0].pos), 1.0f);
				v3 p1 = mat_by_vec(model->transform, f3(&tri->vertices[1].pos), 1.0f);
				v3 p2 = mat_by_vec(model->transform, f3(&tri->vertices[2].pos), 1.0f);
				float t_i;
				int pass_u = 0;
				ctr[ORC_C_TRI]++;
				int hit = intersect_triangle(p0, p1, p2, org, dir, &t_i, &pass_u);
				ctr[ORC_C_TRI_PASS_U] += (uint64_t)pass_u;
				if (hit) {
					if (t_i < tmin) {
						tmin = t_i;
						closest = shape->material;
						rayhit->position = vadd(org, vscale(dir, tmin));
						/* render.cl:340-343 smooth shading; forward matrix, w = 0 */
						v3 w = barycentric_weights(p0, p1, p2, rayhit->position);
						v3 n = vadd(vadd(vscale(f3(&tri->vertices[0].normal), w.x),
						                 vscale(f3(&tri->vertices[1].normal), w.y)),
						            vscale(f3(&tri->vertices[2].normal), w.z));
						n = mat_by_vec(model->transform, n, 0.0f);
						rayhit->normal = vnormalize(n);
					}
				}
			}
		} else if (shape->type == SRT_SHAPE_PLANE) {
			const srt_plane *plane = &shape->shape.plane;
			float t_i;
			ctr[ORC_C_PLANE]++;
			if (orc_intersect_plane(plane, o, d, &t_i)) {
				if (t_i < tmin) {
					tmin = t_i;
					closest = shape->material;
					rayhit->normal = f3(&plane->normal);
					rayhit->position = vadd(org, vscale(dir, tmin));
				}
			}
		}
	}
	/* render.cl:369 `tmin == FLT_MAX` never fires (tmin starts at INFINITY) */
	rayhit->front = vdot(rayhit->normal, dir) < 0.0f;
	rayhit->normal = vscale(rayhit->normal, rayhit->front ? 1.0f : -1.0f);
	return closest;
}

/* OpenCL 3.0 §8.2 linear filter, normalized coords, CLAMP_TO_EDGE
 * (sampler: /root/reference/src/tracer.cpp:47-48). Manual float bilinear. */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static v3 sample_sky(const scene_t *scene, float s, float t) {
	int W = scene->sky_w, H = scene->sky_h;
	float fu = s * (float)W - 0.5f;
	float fv = t * (float)H - 0.5f;
	/* pin float->int for NaN / out-of-range so x86 and gfx950 agree */
	float cu = dm_clamp(fu, -1.0f, (float)W);
	float cv = dm_clamp(fv, -1.0f, (float)H);
	if (!(cu == cu)) cu = 0.0f;
	if (!(cv == cv)) cv = 0.0f;
	float x0f = __builtin_floorf(cu), y0f = __builtin_floorf(cv);
	float a = fu - x0f, b = fv - y0f;
	int x0 = (int)x0f, y0 = (int)y0f;
	int i0 = clampi(x0, 0, W - 1), i1 = clampi(x0 + 1, 0, W - 1);
	int j0 = clampi(y0, 0, H - 1), j1 = clampi(y0 + 1, 0, H - 1);
	const float *T00 = scene->sky + 4 * ((size_t)j0 * W + i0);
	const float *T10 = scene->sky + 4 * ((size_t)j0 * W + i1);
	const float *T01 = scene->sky + 4 * ((size_t)j1 * W + i0);
	const float *T11 = scene->sky + 4 * ((size_t)j1 * W + i1);
	float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
	v3 r;
	r.x = dm_bilinear(w00, T00[0], w10, T10[0], w01, T01[0], w11, T11[0]);
	r.y = dm_bilinear(w00, T00[1], w10, T10[1], w01, T01[1], w11, T11[1]);
	r.z = dm_bilinear(w00, T00[2], w10, T10[2], w01, T01[2], w11, T11[2]);
	return r;
}

/* render.cl:380-394 */
static v3 sky_box(const scene_t *scene, v3 dir) {
	const srt_scene_data *sd = scene->data;
	float lobe = dm_powf(dm_max(vdot(dir, vneg(f3(&sd->sun_direction))), 0.0f), sd->sun_focus);
	v3 sun = vscale(vscale(f3(&sd->sun_color), lobe), sd->sun_intensity);
	float u = dm_atan2pif(dir.z, dir.x) * 0.5f + 0.5f;
	float v = dir.y * 0.5f + 0.5f;
	return vadd(sample_sky(scene, u, v), sun);
}

/* render.cl:396-471 */
static v3 trace(const srt_render_data *render, const scene_t *scene, v3 org, v3 dir, uint32_t seed, uint64_t *ctr) {
	v3 color = V(0.f, 0.f, 0.f);
	v3 mask = V(1.f, 1.f, 1.f);
	hit_t rayhit;
	memset(&rayhit, 0, sizeof rayhit);

	for (int i = 0; i < render->num_bounces; i++) {
		int material_index = closest_intersection(scene, org, dir, &rayhit, ctr);
		if (material_index >= 0) {
			if (render->show_normals) {
				color = V(rayhit.normal.x * 0.5f + 0.5f, rayhit.normal.y * 0.5f + 0.5f, rayhit.normal.z * 0.5f + 0.5f);
				break;
			}
			const srt_material *material = &scene->materials[material_index];
			color = vadd(color, vscale(vmul(mask, f3(&material->emission)), material->emission_strength));
			if (i == render->num_bounces - 1) break;
			ctr[ORC_C_BOUNCE]++;

			org = rayhit.position;
			/* cosine weighted: 6 draws */
			v3 random_dir = vnormalize(vadd(rayhit.normal, random_direction_hemisphere(rayhit.normal, &seed)));
			v3 reflected_dir = reflect3(dir, rayhit.normal);
			int is_metallic = material->metallic > orc_random_float(&seed);
			int is_specular = material->specular > orc_random_float(&seed);
			v3 rough_dir = vmix(random_dir, reflected_dir, material->smoothness);
			int is_transparent = material->transmittance > orc_random_float(&seed);

			if (!is_transparent) {
				dir = vmix(random_dir, rough_dir, (is_metallic || is_specular) ? 1.0f : 0.0f);
				mask = vmul(mask, vmix(f3(&material->color), V(1.0f, 1.0f, 1.0f), is_specular ? 1.0f : 0.0f));
			} else {
				v3 in_dir = reflect3(rough_dir, rayhit.normal);
				float mu = rayhit.front ? 1.0f / material->refraction_index : material->refraction_index;
				float cos_theta = dm_min(1.0f, vdot(in_dir, vneg(rayhit.normal)));
				float sin_theta = dm_sqrtf(1.0f - cos_theta * cos_theta);
				int reflected = mu * sin_theta > 1.0f;
				if (!reflected) reflected = orc_shlick_reflectance(mu, cos_theta) > orc_random_float(&seed);
				if (reflected) {
					dir = rough_dir;
				} else {
					v3 out_perp = vscale(vadd(in_dir, vscale(rayhit.normal, cos_theta)), mu);
					float lsq = out_perp.x * out_perp.x + out_perp.y * out_perp.y + out_perp.z * out_perp.z;
					v3 out_parallel = vscale(rayhit.normal, -dm_sqrtf(dm_fabs(1.0f - lsq)));
					dir = vadd(out_perp, out_parallel);
					mask = vmul(mask, f3(&material->color));
				}
			}
			dir = vnormalize(dir);
			org = vadd(org, vscale(vscale(rayhit.normal, dm_sign(vdot(rayhit.normal, dir))), 0.001f));
		} else {
			ctr[ORC_C_SKY]++;
			mask = vmul(mask, sky_box(scene, dir));
			color = vadd(color, mask);
			break;
		}
	}
	return color;
}

/* render.cl:483-523 for one pixel; returns the per-dispatch colour (already /num_samples) */
static v3 render_pixel(const srt_render_data *data, const scene_t *scene, int px, int py, uint64_t *ctr) {
	uint32_t id = (uint32_t)px + (uint32_t)py * (uint32_t)data->width;
	float wx = (float)px, wy = (float)py;
	v3 color = V(0.f, 0.f, 0.f);
	for (int sample = 0; sample < data->num_samples; sample++) {
		uint32_t seed = ((uint32_t)sample + id * (uint32_t)data->num_samples) * data->time * 5304u;
		float ndc_x = (wx + orc_random_float(&seed)) / (float)data->width;
		float ndc_y = (wy + orc_random_float(&seed)) / (float)data->height;
		float sx = ((2.f * ndc_x - 1.f) * data->aspect_ratio) * data->fov_scale;
		float sy = (1.f - 2.f * ndc_y) * data->fov_scale;
		v3 org = V(data->camera_to_world[3].x, data->camera_to_world[3].y, data->camera_to_world[3].z);
		v3 dir = vnormalize(mat_by_vec(data->camera_to_world, V(sx, sy, -1.0f), 0.0f));
		ctr[ORC_C_PATHS]++;
		color = vadd(color, trace(data, scene, org, dir, seed, ctr));
	}
	return vdivs(color, (float)data->num_samples);
}

/* The `render` kernel over rows [y0, y1): canvas[id] += colour. canvas is the full
 * width*height image of 16-byte float3 (x,y,z,pad). counters may be NULL. */
void orc_render_strided(const srt_render_data *data, const srt_scene_data *scene_data, float *canvas,
                        const srt_shape *shapes, const srt_triangle *triangles, const srt_material *materials,
                        const float *sky_rgba, int sky_w, int sky_h, int y0, int y1, int ystride, int nthreads,
                        uint64_t *counters);

void orc_render(const srt_render_data *data, const srt_scene_data *scene_data, float *canvas,
                const srt_shape *shapes, const srt_triangle *triangles, const srt_material *materials,
                const float *sky_rgba, int sky_w, int sky_h, int y0, int y1, int nthreads, uint64_t *counters) {
	orc_render_strided(data, scene_data, canvas, shapes, triangles, materials, sky_rgba, sky_w, sky_h, y0, y1, 1, nthreads,
	                   counters);
}

/* rows y0, y0+ystride, ... < y1 (a strided sample of the frame for the CPU baseline) */
void orc_render_strided(const srt_render_data *data, const srt_scene_data *scene_data, float *canvas,
                        const srt_shape *shapes, const srt_triangle *triangles, const srt_material *materials,
                        const float *sky_rgba, int sky_w, int sky_h, int y0, int y1, int ystride, int nthreads,
                        uint64_t *counters) {
	scene_t scene = {scene_data, shapes, triangles, materials, sky_rgba, sky_w, sky_h};
	uint64_t total[ORC_C_COUNT];
	memset(total, 0, sizeof total);
#ifdef _OPENMP
	if (nthreads > 0) omp_set_num_threads(nthreads);
#else
	(void)nthreads;
#endif
#pragma omp parallel
	{
		uint64_t ctr[ORC_C_COUNT];
		memset(ctr, 0, sizeof ctr);
		if (ystride < 1) ystride = 1;
		const int nrows = y1 > y0 ? (y1 - y0 + ystride - 1) / ystride : 0;
		/* pixels, not rows, are the unit of work: a sample of a few rows still fills every core */
#pragma omp for collapse(2) schedule(dynamic, 16)
		for (int r = 0; r < nrows; r++) {
			for (int x = 0; x < data->width; x++) {
				const int y = y0 + r * ystride;
				v3 c = render_pixel(data, &scene, x, y, ctr);
				float *out = canvas + 4 * ((size_t)y * data->width + x);
				out[0] += c.x;
				out[1] += c.y;
				out[2] += c.z;
				if (c.x != c.x || c.y != c.y || c.z != c.z) ctr[ORC_C_NAN_PIXELS]++;
			}
		}
#pragma omp critical
		for (int k = 0; k < ORC_C_COUNT; k++) total[k] += ctr[k];
	}
	if (counters)
		for (int k = 0; k < ORC_C_COUNT; k++) counters[k] += total[k];
}

/* radiance of single (pixel, sample) paths — for trace-level golden vectors */
void orc_trace_paths(const srt_render_data *data, const srt_scene_data *scene_data, const srt_shape *shapes,
                     const srt_triangle *triangles, const srt_material *materials, const float *sky_rgba, int sky_w,
                     int sky_h, const int32_t *pixel_ids, const int32_t *samples, int n, float *out_rgb) {
	scene_t scene = {scene_data, shapes, triangles, materials, sky_rgba, sky_w, sky_h};
	uint64_t ctr[ORC_C_COUNT];
	memset(ctr, 0, sizeof ctr);
	for (int k = 0; k < n; k++) {
		uint32_t id = (uint32_t)pixel_ids[k];
		int px = (int)(id % (uint32_t)data->width), py = (int)(id / (uint32_t)data->width);
		uint32_t seed = ((uint32_t)samples[k] + id * (uint32_t)data->num_samples) * data->time * 5304u;
		float ndc_x = ((float)px + orc_random_float(&seed)) / (float)data->width;
		float ndc_y = ((float)py + orc_random_float(&seed)) / (float)data->height;
		float sx = ((2.f * ndc_x - 1.f) * data->aspect_ratio) * data->fov_scale;
		float sy = (1.f - 2.f * ndc_y) * data->fov_scale;
		v3 org = V(data->camera_to_world[3].x, data->camera_to_world[3].y, data->camera_to_world[3].z);
		v3 dir = vnormalize(mat_by_vec(data->camera_to_world, V(sx, sy, -1.0f), 0.0f));
		v3 c = trace(data, &scene, org, dir, seed, ctr);
		out_rgb[3 * k + 0] = c.x;
		out_rgb[3 * k + 1] = c.y;
		out_rgb[3 * k + 2] = c.z;
	}
}

/* render.cl:473-481 */
static inline float aces1(float x) {
	const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
	return dm_clamp((x * (x * a + b)) / (x * (x * c + d) + e), 0.0f, 1.0f);
}

/* float -> uchar as x86 cvttss2si + truncation does for in-range values; NaN -> 0 */
static inline uint8_t to_uchar(float v) {
	if (!(v == v)) return 0;
	return (uint8_t)(int)v;
}

/* render.cl:525-535 — memory byte order A,R,G,B */
void orc_average(uint32_t num_steps, const float *canvas, uint8_t *output, size_t num_pixels) {
	for (size_t id = 0; id < num_pixels; id++) {
		float n = (float)num_steps;
		float r = dm_sqrtf(aces1(canvas[4 * id + 0] / n));
		float g = dm_sqrtf(aces1(canvas[4 * id + 1] / n));
		float b = dm_sqrtf(aces1(canvas[4 * id + 2] / n));
		output[4 * id + 0] = 255;
		output[4 * id + 1] = to_uchar(r * 255.0f);
		output[4 * id + 2] = to_uchar(g * 255.0f);
		output[4 * id + 3] = to_uchar(b * 255.0f);
	}
}

/* ---- small KAT entry points ------------------------------------------------ */
void orc_matrix_by_vector(const float *m16, const float *v4, float *out4) {
	srt_float4 m[4];
	memcpy(m, m16, sizeof m);
	out4[0] = ((m[0].x * v4[0] + m[1].x * v4[1]) + m[2].x * v4[2]) + m[3].x * v4[3];
	out4[1] = ((m[0].y * v4[0] + m[1].y * v4[1]) + m[2].y * v4[2]) + m[3].y * v4[3];
	out4[2] = ((m[0].z * v4[0] + m[1].z * v4[1]) + m[2].z * v4[2]) + m[3].z * v4[3];
	out4[3] = ((m[0].w * v4[0] + m[1].w * v4[1]) + m[2].w * v4[2]) + m[3].w * v4[3];
}
void orc_barycentric_weights(const float *p0, const float *p1, const float *p2, const float *p, float *out3) {
	v3 w = barycentric_weights(V(p0[0], p0[1], p0[2]), V(p1[0], p1[1], p1[2]), V(p2[0], p2[1], p2[2]), V(p[0], p[1], p[2]));
	out3[0] = w.x;
	out3[1] = w.y;
	out3[2] = w.z;
}
void orc_sky_box(const srt_scene_data *sd, const float *sky, int w, int h, const float *dir, float *out3) {
	scene_t scene = {sd, NULL, NULL, NULL, sky, w, h};
	v3 c = sky_box(&scene, V(dir[0], dir[1], dir[2]));
	out3[0] = c.x;
	out3[1] = c.y;
	out3[2] = c.z;
}
void orc_aces(const float *in3, float *out3) {
	for (int k = 0; k < 3; k++) out3[k] = aces1(in3[k]);
}
float orc_cos(float x) { return dm_cosf(x); }
float orc_log(float x) { return dm_logf(x); }
float orc_pow(float x, float y) { return dm_powf(x, y); }
float orc_atan2pi(float y, float x) { return dm_atan2pif(y, x); }
int orc_num_counters(void) { return ORC_C_COUNT; }

/* Host-side twin of the device self-test (srt_selftest_math): sums of result bit
 * patterns of detmath over r = 0, stride, ... ; out[3..7] as in include/srt_abi.h. */
static inline uint64_t canon_bits(float a) { return (a != a) ? 0x7fc00000ull : (uint64_t)dm_f2u(a); }
void orc_math_checksums(uint32_t stride, uint64_t *out8) {
	uint64_t s_log = 0, s_cos = 0, s_sqrt = 0, s_atan = 0, s_pow = 0, s_norm = 0;
	const uint64_t total = (0x100000000ull + stride - 1) / stride;
#pragma omp parallel for reduction(+ : s_log, s_cos, s_sqrt, s_atan, s_pow, s_norm) schedule(static)
	for (uint64_t i = 0; i < total; i++) {
		const uint32_t r = (uint32_t)(i * stride);
		const float u = (float)r / 4294967296.0f;
		const float th = 6.28318548f * u;
		s_log += canon_bits(dm_logf(u));
		s_cos += canon_bits(dm_cosf(th));
		s_sqrt += canon_bits(dm_sqrtf(u));
		s_atan += canon_bits(dm_atan2pif(u - 0.5f, 0.37f - u));
		s_pow += canon_bits(dm_powf(u, 25.0f));
		const v3 nv = vnormalize(V(u - 0.5f, 0.37f - u, (float)(r & 0xffffu) * 1e-3f - 30.0f));
		s_norm += canon_bits(nv.x) + canon_bits(nv.y) + canon_bits(nv.z);
	}
	for (int k = 0; k < 8; k++) out8[k] = 0;
	out8[0] = s_norm; /* compared with out[9] of the device self-test */
	out8[3] = s_log;
	out8[4] = s_cos;
	out8[5] = s_sqrt;
	out8[6] = s_atan;
	out8[7] = s_pow;
}
