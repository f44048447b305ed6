// srt_internal.h — the handle behind `srt_tracer *` and the small helpers the translation units of
// libsrt_hip.so share (srt_abi.hip: life cycle, scene, trace; srt_collect.hip: multi-GPU collection,
// frame pipeline). Not part of the public ABI.
#ifndef SRT_INTERNAL_H
#define SRT_INTERNAL_H

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/srt_abi.h"
#include "device_types.h"

template <class T>
struct DevBuf {
	T *ptr = nullptr;
	size_t cap = 0; // elements
	// "buffers only ever grow" (src/tracer.cpp:5-9)
	hipError_t reserve(size_t n) {
		if (n < 1) n = 1;
		if (n <= cap) return hipSuccess;
		if (ptr) (void)hipFree(ptr);
		ptr = nullptr;
		cap = 0;
		hipError_t e = hipMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T));
		if (e == hipSuccess) cap = n;
		return e;
	}
	void release() {
		if (ptr) (void)hipFree(ptr);
		ptr = nullptr;
		cap = 0;
	}
};

struct SrtCollect; // srt_collect.hip: RCCL communicator, gathered canvases, frame pipeline

struct srt_tracer {
	int width = 0, height = 0, device = 0;
	SrtCollect *collect = nullptr;
	hipStream_t own_stream = nullptr, stream = nullptr;
	DevBuf<float> canvas_own;
	float *canvas = nullptr;
	size_t canvas_bytes = 0; // of the buffer in use
	DevBuf<uint8_t> argb;
	DevBuf<srt_shape> shapes;
	DevBuf<BlockGroup> runs; // group headers of the packed shape blocks
	DevBuf<float> run_data;
	DevBuf<WinnerRec> winners;
	int num_runs = 0;
	size_t num_materials = 0;
	DevBuf<srt_triangle> triangles;
	DevBuf<srt_material> materials;
	DevBuf<float> wtris;
	DevBuf<uint32_t> wtri_offset;
	DevBuf<float> sky;
	DevBuf<uint32_t> bvh_blocks; // wide hierarchy, 32 dwords per block (device_types.h)
	DevBuf<uint32_t> bvh_order, bvh_dest;
	int accel_mode = SRT_ACCEL_NONE; // what the next srt_update_scene builds
	bool bvh_active = false;         // the current scene's models carry BVH roots
	uint64_t bvh_info[7] = {0, 0, 0, 0, 0, 0, 0};
	struct BvhCache *bvh_cache = nullptr; // hierarchies of the previous srt_update_scene (see BvhCacheEntry)
	DevBuf<unsigned long long> counters;
	DevBuf<unsigned long long> wave_counters; // per persistent wave, summed in srt_get_counters
	DevBuf<float> scan_queue;                 // array scan: per persistent wave (two sets, as the counters), allocated when first needed
	DevBuf<float> radiance;  // 3 floats per (pixel, sample) of the current batch
	DevBuf<float> running;   // float4 per pixel, carries the ordered sum across batches
	size_t radiance_budget = 0; // bytes; 0 = pick from free HBM at first use
	size_t radiance_budget_env = 0; // bytes from SRT_RADIANCE_BUDGET_MB as srt_create found it; 0 = not set
	int num_cus = 0;
	int last_waves_per_cu = 0, last_grid = 0;
	std::vector<hipEvent_t> ev_k; // one pair per sample batch, around srt_trace_kernel alone (reduce excluded)
	// dispatches of several sample batches: even / odd batches trace on two streams of their own so that one batch's tail
	// (its last long paths, a few lanes per wave) runs under the next batch's start; the reductions stay in order on `stream`
	hipStream_t batch_stream[2] = {nullptr, nullptr};
	bool timers_in_render = false; // srt_render / srt_render_async / srt_render_pipelined record the kernel timers' events too (srt_set_kernel_timers)
	bool queue_dirty[2] = {false, false}; // a trace launch used the work cursor and no reduction has reset it since (srt_trace)
	hipEvent_t ev_batch_traced[2] = {nullptr, nullptr}, ev_batch_reduced[2] = {nullptr, nullptr}, ev_batch_fork = nullptr;
	bool batches_overlapped = false; // the last srt_trace ran that way (srt_last_trace_kernel_ms: a span, not a sum)
	size_t ev_k_used = 0;         // events of the last srt_trace
	float last_trace_kernel_ms = 0.f, last_reduce_ms = 0.f;
	int sky_w = 0, sky_h = 0;
	srt_scene_data sd{};
	int num_models = 0;
	bool all_materials_ok = false; // no shape of the scene has a negative material index
	bool unit_materials = false;   // the device material table holds bernoulli() thresholds (srt_update_scene)
	int material_flags = 0;        // SRT_MF_* (device_types.h)
	uint64_t scan_tris = 0; // array scan: triangles of the models a ray can be made to scan (all of them), for the launch-length bound
	bool scene_set = false;
	bool count_tris = false;
	int rank = 0, world = 1, rows_per_block = 8, owned_rows = 0;
	hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr, ev_r0 = nullptr, ev_r1 = nullptr;
	bool have_trace_ev = false, have_resolve_ev = false, have_kernel_ev = false;
	std::string err;
};

int srt_fail(srt_tracer *t, int code, const std::string &msg); // records the text for srt_last_error (t == NULL: srt_create's)
static inline int fail(srt_tracer *t, int code, const std::string &msg) { return srt_fail(t, code, msg); }

#define SRT_HIP(t, call)                                                                              \
	do {                                                                                              \
		hipError_t e_ = (call);                                                                       \
		if (e_ != hipSuccess)                                                                         \
			return fail((t), SRT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));         \
	} while (0)


void srt_collect_release(srt_tracer *t);
/* srt_trace whose last reduction also resolves into fused_argb (device, owned pixels x 4 bytes; NULL: plain srt_trace) */
extern "C" int srt_trace_fused(srt_tracer *t, const srt_render_data *options, uint8_t *fused_argb, uint32_t ticks_stopped);
/* one scene for several handles: the host pass once (members[0]'s acceleration mode and hierarchy cache), the uploads of all
 * members enqueued before the first is waited for (srt_abi.hip; srt_group_update_scene). *failed_member = the member an error came from */
extern "C" int srt_update_scene_many(srt_tracer *const *members, size_t n_members, const srt_shape *shapes, size_t n_shapes, const srt_triangle *triangles,
                          size_t n_triangles, const srt_material *materials, size_t n_materials, const srt_scene_data *scene, size_t *failed_member);

#endif
