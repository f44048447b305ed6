// srt_collect.hip — what happens to a frame AFTER the trace kernel, beyond one blocking device:
//
//  * collecting the row partition of a multi-GPU frame on one GPU: ONE ncclGather (RCCL over xGMI) of the
//    packed per-rank canvases, then an unpermute kernel on the root -- for one process per GPU
//    (srt_comm_*, srt_gather) and for one process driving several GPUs (srt_group_*, what a C++ front-end
//    that keeps the reference's `Tracer` class needs). New work: the reference is single-device
//    (/root/reference/src/tracer.cpp:13).
//  * a two-deep frame pipeline for the interactive loop (/root/reference/src/main.cpp:277-337): frame N's
//    read-back overlaps frame N+1's trace (srt_render_pipelined).
//
// RCCL is loaded with dlopen at first use, so libsrt_hip.so has no link-time dependency on it and a
// single-GPU user never touches it.
#include <dlfcn.h>

#include <cstring>
#include <new>

#include "srt_internal.h"

// ---------------------------------------------------------------------------------
// RCCL entry points (signatures of /opt/rocm/include/rccl/rccl.h)
// ---------------------------------------------------------------------------------
namespace {
typedef struct ncclComm *ncclComm_t;
struct ncclUniqueId {
	char internal[SRT_COMM_ID_BYTES];
};
static_assert(SRT_COMM_ID_BYTES == 128, "NCCL_UNIQUE_ID_BYTES");
constexpr int kNcclSuccess = 0, kNcclFloat = 7;

struct Rccl {
	void *lib = nullptr;
	int (*GetUniqueId)(ncclUniqueId *) = nullptr;
	int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
	int (*CommDestroy)(ncclComm_t) = nullptr;
	int (*Gather)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
	std::string error;
};

Rccl &rccl() {
	static Rccl r;
	return r;
}

bool rccl_load() {
	Rccl &r = rccl();
	if (r.lib) return true;
	for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
		r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
		if (r.lib) break;
	}
	if (!r.lib) {
		r.error = std::string("RCCL not found (dlopen librccl.so.1): ") + (dlerror() ? dlerror() : "");
		return false;
	}
	auto sym = [&](const char *n) { return dlsym(r.lib, n); };
	r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
	r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
	r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
	r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
	r.Gather = reinterpret_cast<decltype(r.Gather)>(sym("ncclGather"));
	r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
	r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
	r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
	if (!r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.Gather || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
		r.error = "librccl.so lacks an entry point this library needs (ncclGather and friends)";
		dlclose(r.lib);
		r.lib = nullptr;
		return false;
	}
	return true;
}

#define SRT_NCCL(t, call)                                                                                        \
	do {                                                                                                         \
		int r_ = (call);                                                                                         \
		if (r_ != kNcclSuccess) return fail((t), SRT_ERR_HIP, std::string(#call) + ": " + rccl().GetErrorString(r_)); \
	} while (0)
} // namespace

// per-handle state of this module
struct SrtCollect {
	ncclComm_t comm = nullptr;
	bool comm_owned = false; // created by srt_comm_init (else by a group's ncclCommInitAll)
	int comm_rank = -1, comm_world = 0;
	DevBuf<float> gathered; // root: world x padded_rows x width float4, rank-major
	DevBuf<float> full;     // root: height x width float4, the unpermuted image
	DevBuf<uint8_t> full_argb;
	bool have_full = false;
	// frame pipeline
	hipStream_t copy_stream = nullptr;
	uint8_t *pinned[2] = {nullptr, nullptr};
	uint8_t *dev_argb[2] = {nullptr, nullptr};
	size_t frame_bytes = 0;
	hipEvent_t resolved[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};
	long long frames_enqueued = 0, frames_delivered = 0;
};

namespace {
SrtCollect *collect_of(srt_tracer *t) {
	if (!t->collect) t->collect = new (std::nothrow) SrtCollect();
	return t->collect;
}

// image row y of the frame <- packed row of the rank that owns it (include/srt_abi.h srt_set_partition)
__global__ __launch_bounds__(256) void srt_unpermute_kernel(const float4 *__restrict__ gathered, float4 *__restrict__ image, int width, int height,
                                                            int world, int rpb, int padded_rows) {
	const size_t n = (size_t)width * height;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const int y = (int)(i / width), x = (int)(i - (size_t)y * width);
		const int blk = y / rpb, r = blk % world, lb = blk / world;
		const int lr = lb * rpb + (y - blk * rpb);
		image[i] = gathered[((size_t)r * padded_rows + lr) * width + x];
	}
}

int unpermute_on_root(srt_tracer *t, SrtCollect *c) {
	const int padded = srt_partition_padded_rows(t->height, t->world, t->rows_per_block);
	const size_t n = (size_t)t->width * t->height;
	SRT_HIP(t, c->full.reserve(n * 4));
	unsigned blocks = (unsigned)((n + 255) / 256);
	if (blocks > 8192) blocks = 8192;
	hipLaunchKernelGGL(srt_unpermute_kernel, dim3(blocks), dim3(256), 0, t->stream, reinterpret_cast<const float4 *>(c->gathered.ptr),
	                   reinterpret_cast<float4 *>(c->full.ptr), t->width, t->height, t->world, t->rows_per_block, padded);
	SRT_HIP(t, hipGetLastError());
	c->have_full = true;
	return SRT_OK;
}
} // namespace

void srt_collect_release(srt_tracer *t) {
	SrtCollect *c = t->collect;
	if (!c) return;
	if (c->comm && c->comm_owned && rccl().lib) (void)rccl().CommDestroy(c->comm);
	c->gathered.release();
	c->full.release();
	c->full_argb.release();
	if (c->copy_stream) {
		(void)hipStreamSynchronize(c->copy_stream);
		(void)hipStreamDestroy(c->copy_stream);
	}
	for (int i = 0; i < 2; i++) {
		if (c->pinned[i]) (void)hipHostFree(c->pinned[i]);
		if (c->dev_argb[i]) (void)hipFree(c->dev_argb[i]);
		if (c->resolved[i]) (void)hipEventDestroy(c->resolved[i]);
		if (c->copied[i]) (void)hipEventDestroy(c->copied[i]);
	}
	delete c;
	t->collect = nullptr;
}

extern "C" {

// ---------------------------------------------------------------------------------
// one process per GPU
// ---------------------------------------------------------------------------------
int srt_comm_unique_id(void *id_out) {
	if (!id_out) return SRT_ERR_INVALID;
	if (!rccl_load()) return fail(nullptr, SRT_ERR_HIP, rccl().error);
	ncclUniqueId id;
	SRT_NCCL(nullptr, rccl().GetUniqueId(&id));
	memcpy(id_out, id.internal, SRT_COMM_ID_BYTES);
	return SRT_OK;
}

int srt_comm_init(srt_tracer *t, const void *id, int rank, int world) {
	if (!t) return SRT_ERR_INVALID;
	if (!id || world < 1 || rank < 0 || rank >= world) return fail(t, SRT_ERR_INVALID, "srt_comm_init: need an id and 0 <= rank < world");
	if (!rccl_load()) return fail(t, SRT_ERR_HIP, rccl().error);
	SrtCollect *c = collect_of(t);
	if (!c) return fail(t, SRT_ERR_INVALID, "out of host memory");
	if (c->comm) return fail(t, SRT_ERR_STATE, "srt_comm_init: the handle already has a communicator");
	SRT_HIP(t, hipSetDevice(t->device));
	ncclUniqueId uid;
	memcpy(uid.internal, id, SRT_COMM_ID_BYTES);
	SRT_NCCL(t, rccl().CommInitRank(&c->comm, world, uid, rank));
	c->comm_owned = true;
	c->comm_rank = rank;
	c->comm_world = world;
	return SRT_OK;
}

int srt_gather(srt_tracer *t, int root) {
	if (!t) return SRT_ERR_INVALID;
	SrtCollect *c = t->collect;
	if (!c || !c->comm) return fail(t, SRT_ERR_STATE, "srt_gather: no communicator (srt_comm_init)");
	if (c->comm_world != t->world || c->comm_rank != t->rank)
		return fail(t, SRT_ERR_STATE, "srt_gather: the communicator's rank / world differ from srt_set_partition's");
	if (root < 0 || root >= t->world) return fail(t, SRT_ERR_INVALID, "srt_gather: root out of range");
	SRT_HIP(t, hipSetDevice(t->device));
	const int padded = srt_partition_padded_rows(t->height, t->world, t->rows_per_block);
	const size_t count = (size_t)padded * t->width * 4; // floats every rank sends (its canvas buffer holds padded rows)
	if (t->canvas_bytes < count * sizeof(float)) return fail(t, SRT_ERR_STATE, "srt_gather: canvas smaller than padded_rows * width * 16 (bind a larger one)");
	if (t->rank == root) SRT_HIP(t, c->gathered.reserve(count * (size_t)t->world));
	// the ONE collective of the path: every peer's packed canvas over its own xGMI link to the root
	SRT_NCCL(t, rccl().Gather(t->canvas, t->rank == root ? c->gathered.ptr : nullptr, count, kNcclFloat, root, c->comm, t->stream));
	if (t->rank == root) return unpermute_on_root(t, c);
	return SRT_OK;
}

/* test hook: the unpermute kernel alone on caller-owned device buffers (gathered: world x padded_rows x width float4,
 * image: height x width float4), on the NULL stream, synchronous */
int srt_unpermute_device(const void *gathered, void *image, int width, int height, int world, int rows_per_block) {
	if (!gathered || !image || width <= 0 || height <= 0 || world < 1 || rows_per_block < 1) return SRT_ERR_INVALID;
	const int padded = srt_partition_padded_rows(height, world, rows_per_block);
	const size_t n = (size_t)width * height;
	unsigned blocks = (unsigned)((n + 255) / 256);
	if (blocks > 8192) blocks = 8192;
	hipLaunchKernelGGL(srt_unpermute_kernel, dim3(blocks), dim3(256), 0, nullptr, static_cast<const float4 *>(gathered), static_cast<float4 *>(image),
	                   width, height, world, rows_per_block, padded);
	if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return SRT_ERR_HIP;
	return SRT_OK;
}

int srt_resolve_gathered(srt_tracer *t, uint32_t ticks_stopped) {
	if (!t) return SRT_ERR_INVALID;
	SrtCollect *c = t->collect;
	if (!c || !c->have_full) return fail(t, SRT_ERR_STATE, "srt_resolve_gathered: nothing gathered on this handle (srt_gather as root)");
	const size_t n = (size_t)t->width * t->height;
	SRT_HIP(t, c->full_argb.reserve(n * 4));
	return srt_resolve_external(t, c->full.ptr, (uint32_t)n, ticks_stopped, c->full_argb.ptr);
}

int srt_gathered_buffers(srt_tracer *t, void **canvas, void **argb) {
	if (!t) return SRT_ERR_INVALID;
	SrtCollect *c = t->collect;
	if (!c || !c->have_full) return fail(t, SRT_ERR_STATE, "srt_gathered_buffers: nothing gathered on this handle");
	if (canvas) *canvas = c->full.ptr;
	if (argb) *argb = c->full_argb.ptr;
	return SRT_OK;
}

int srt_read_gathered(srt_tracer *t, float *canvas_out, uint8_t *argb_out) {
	if (!t) return SRT_ERR_INVALID;
	SrtCollect *c = t->collect;
	if (!c || !c->have_full) return fail(t, SRT_ERR_STATE, "srt_read_gathered: nothing gathered on this handle");
	SRT_HIP(t, hipSetDevice(t->device));
	const size_t n = (size_t)t->width * t->height;
	if (canvas_out) SRT_HIP(t, hipMemcpyAsync(canvas_out, c->full.ptr, n * 16, hipMemcpyDeviceToHost, t->stream));
	if (argb_out) {
		if (c->full_argb.cap < n * 4) return fail(t, SRT_ERR_STATE, "srt_read_gathered: no resolved image (srt_resolve_gathered)");
		SRT_HIP(t, hipMemcpyAsync(argb_out, c->full_argb.ptr, n * 4, hipMemcpyDeviceToHost, t->stream));
	}
	SRT_HIP(t, hipStreamSynchronize(t->stream));
	return SRT_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------
// one process, several GPUs: what `Tracer(width, height, n_devices)` of host/tracer.hpp drives
// ---------------------------------------------------------------------------------
struct srt_group {
	std::vector<srt_tracer *> t;
	int width = 0, height = 0, rpb = 8;
	// The device list names a device more than once ("virtual devices": N members on fewer GPUs, down to one). RCCL refuses such a
	// list, and there is no link to cross: the collection is then a device-to-device copy of every member's packed canvas into its
	// slot of the root's buffer, enqueued on the member's own stream where ncclGather would be, and the root's stream waits for all
	// of them before it unpermutes. Partition, packed layout, unpermute, resolve and every srt_group_* call are the real ones --
	// what a one-GPU box can run of the N > 1 path (tests/test_gpu_collect.py).
	bool loopback = false;
	std::vector<hipEvent_t> copied; // loopback: member i's canvas has arrived in the root's buffer
	std::string err;
};

namespace {
int gfail(srt_group *g, int code, const std::string &msg) {
	if (g) g->err = msg;
	return code;
}
} // namespace

extern "C" {

void srt_group_destroy(srt_group *g) {
	if (!g) return;
	for (hipEvent_t e : g->copied)
		if (e) (void)hipEventDestroy(e);
	for (srt_tracer *t : g->t) {
		if (t && t->collect && t->collect->comm && rccl().lib) {
			(void)hipSetDevice(t->device);
			(void)hipStreamSynchronize(t->stream);
			(void)rccl().CommDestroy(t->collect->comm);
			t->collect->comm = nullptr;
		}
	}
	for (srt_tracer *t : g->t) srt_destroy(t);
	delete g;
}

const char *srt_group_last_error(const srt_group *g) { return g ? g->err.c_str() : "srt_group: NULL"; }

int srt_group_create(int width, int height, int n_devices, const int *devices, int rows_per_block, srt_group **out) {
	if (!out) return SRT_ERR_INVALID;
	*out = nullptr;
	if (n_devices < 1 || rows_per_block < 1) return fail(nullptr, SRT_ERR_INVALID, "srt_group_create: need n_devices >= 1 and rows_per_block >= 1");
	srt_group *g = new (std::nothrow) srt_group();
	if (!g) return fail(nullptr, SRT_ERR_INVALID, "out of host memory");
	g->width = width, g->height = height, g->rpb = rows_per_block;
	std::vector<int> devs(n_devices);
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		(void)hipGetLastError();
		delete g;
		return fail(nullptr, SRT_ERR_HIP, "srt_group_create: no HIP device");
	}
	// no list: devices 0 .. n-1; more members than the node has devices: they wrap around (virtual devices, see srt_group::loopback)
	for (int i = 0; i < n_devices; i++) devs[i] = devices ? devices[i] : i % ndev;
	for (int i = 0; i < n_devices && !g->loopback; i++)
		for (int k = 0; k < i; k++)
			if (devs[k] == devs[i]) g->loopback = true;
	for (int i = 0; i < n_devices; i++) {
		srt_tracer *t = nullptr;
		int rc = srt_create(width, height, devs[i], &t);
		if (rc == SRT_OK) rc = srt_set_partition(t, i, n_devices, rows_per_block);
		if (rc != SRT_OK) {
			const std::string msg = std::string("srt_group_create: device ") + std::to_string(devs[i]) + ": " + srt_last_error(rc && t ? t : nullptr);
			if (t) srt_destroy(t);
			srt_group_destroy(g);
			return fail(nullptr, rc, msg);
		}
		g->t.push_back(t);
	}
	if (g->loopback) {
		for (int i = 0; i < n_devices; i++) {
			SrtCollect *c = collect_of(g->t[i]);
			hipEvent_t ev = nullptr;
			if (!c || hipSetDevice(devs[i]) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
				(void)hipGetLastError();
				srt_group_destroy(g);
				return fail(nullptr, SRT_ERR_HIP, "srt_group_create: loopback set-up");
			}
			g->copied.push_back(ev);
			c->comm_rank = i, c->comm_world = n_devices;
		}
		*out = g;
		return SRT_OK;
	}
	if (!rccl_load()) {
		srt_group_destroy(g);
		return fail(nullptr, SRT_ERR_HIP, rccl().error);
	}
	std::vector<ncclComm_t> comms(n_devices, nullptr);
	const int r = rccl().CommInitAll(comms.data(), n_devices, devs.data());
	if (r != kNcclSuccess) {
		const std::string msg = std::string("srt_group_create: ncclCommInitAll: ") + rccl().GetErrorString(r);
		srt_group_destroy(g);
		return fail(nullptr, SRT_ERR_HIP, msg);
	}
	for (int i = 0; i < n_devices; i++) {
		SrtCollect *c = collect_of(g->t[i]);
		if (!c) {
			// the communicators not yet attached to a handle (i and above) belong to nobody: destroy them here, the attached ones
			// go with their handles in srt_group_destroy
			for (int k = i; k < n_devices; k++)
				if (comms[k]) (void)rccl().CommDestroy(comms[k]);
			srt_group_destroy(g);
			return fail(nullptr, SRT_ERR_INVALID, "out of host memory");
		}
		c->comm = comms[i];
		c->comm_owned = false; // destroyed by srt_group_destroy
		c->comm_rank = i, c->comm_world = n_devices;
	}
	*out = g;
	return SRT_OK;
}

int srt_group_size(const srt_group *g) { return g ? (int)g->t.size() : 0; }
srt_tracer *srt_group_tracer(srt_group *g, int i) { return (g && i >= 0 && i < (int)g->t.size()) ? g->t[i] : nullptr; }

#define SRT_EACH(g, expr)                                                                  \
	do {                                                                                   \
		for (srt_tracer * t_ : (g)->t) {                                                   \
			const int rc_ = (expr);                                                        \
			if (rc_ != SRT_OK) return gfail((g), rc_, srt_last_error(t_));                 \
		}                                                                                  \
	} while (0)

int srt_group_set_skybox(srt_group *g, const float *rgba, int width, int height) {
	if (!g) return SRT_ERR_INVALID;
	SRT_EACH(g, srt_set_skybox(t_, rgba, width, height));
	return SRT_OK;
}

int srt_group_set_acceleration(srt_group *g, int mode) {
	if (!g) return SRT_ERR_INVALID;
	SRT_EACH(g, srt_set_acceleration(t_, mode));
	return SRT_OK;
}

int srt_group_update_scene(srt_group *g, const srt_shape *shapes, size_t n_shapes, const srt_triangle *triangles, size_t n_triangles,
                           const srt_material *materials, size_t n_materials, const srt_scene_data *scene) {
	if (!g) return SRT_ERR_INVALID;
	// the scene is replicated: prepared on the host once (hierarchy build, shape blocks, thresholds), uploaded to every member
	size_t who = 0;
	const int rc = srt_update_scene_many(g->t.data(), g->t.size(), shapes, n_shapes, triangles, n_triangles, materials, n_materials, scene, &who);
	if (rc != SRT_OK) return gfail(g, rc, srt_last_error(g->t[who < g->t.size() ? who : 0]));
	return SRT_OK;
}

int srt_group_clear_canvas(srt_group *g) {
	if (!g) return SRT_ERR_INVALID;
	SRT_EACH(g, srt_clear_canvas(t_));
	return SRT_OK;
}

// trace on every device, ONE gather to device 0 of the group, unpermute there; everything asynchronous
int srt_group_trace_and_gather(srt_group *g, const srt_render_data *options) {
	if (!g) return SRT_ERR_INVALID;
	SRT_EACH(g, srt_trace(t_, options));
	srt_tracer *root = g->t[0];
	const int world = (int)g->t.size();
	const int padded = srt_partition_padded_rows(g->height, world, g->rpb);
	const size_t count = (size_t)padded * g->width * 4;
	SrtCollect *rc = root->collect;
	if (hipSetDevice(root->device) != hipSuccess || rc->gathered.reserve(count * (size_t)world) != hipSuccess)
		return gfail(g, SRT_ERR_HIP, "srt_group: gather buffer on the root device");
	if (g->loopback) {
		for (int i = 0; i < world; i++) {
			srt_tracer *t = g->t[i];
			if (hipSetDevice(t->device) != hipSuccess ||
			    hipMemcpyAsync(rc->gathered.ptr + (size_t)i * count, t->canvas, count * sizeof(float), hipMemcpyDeviceToDevice, t->stream) != hipSuccess ||
			    hipEventRecord(g->copied[i], t->stream) != hipSuccess)
				return gfail(g, SRT_ERR_HIP, "srt_group: loopback copy of a member's canvas");
		}
		(void)hipSetDevice(root->device);
		for (int i = 1; i < world; i++)
			if (hipStreamWaitEvent(root->stream, g->copied[i], 0) != hipSuccess) return gfail(g, SRT_ERR_HIP, "srt_group: loopback wait");
		const int u = unpermute_on_root(root, rc);
		if (u != SRT_OK) return gfail(g, u, srt_last_error(root));
		return SRT_OK;
	}
	int r = rccl().GroupStart();
	for (int i = 0; i < world && r == kNcclSuccess; i++) {
		srt_tracer *t = g->t[i];
		(void)hipSetDevice(t->device);
		r = rccl().Gather(t->canvas, i == 0 ? rc->gathered.ptr : nullptr, count, kNcclFloat, 0, t->collect->comm, t->stream);
	}
	const int r2 = rccl().GroupEnd();
	if (r != kNcclSuccess || r2 != kNcclSuccess) return gfail(g, SRT_ERR_HIP, std::string("srt_group: ncclGather: ") + rccl().GetErrorString(r != kNcclSuccess ? r : r2));
	(void)hipSetDevice(root->device);
	const int u = unpermute_on_root(root, rc);
	if (u != SRT_OK) return gfail(g, u, srt_last_error(root));
	return SRT_OK;
}

// Tracer::render for the group: trace everywhere, gather, resolve on device 0, blocking read-back of width*height*4 bytes
int srt_group_render(srt_group *g, const srt_render_data *options, uint32_t ticks_stopped, uint8_t *argb_out) {
	if (!g) return SRT_ERR_INVALID;
	if (!argb_out) return gfail(g, SRT_ERR_INVALID, "srt_group_render: argb_out is NULL");
	int rc = srt_group_trace_and_gather(g, options);
	if (rc != SRT_OK) return rc;
	srt_tracer *root = g->t[0];
	rc = srt_resolve_gathered(root, ticks_stopped);
	if (rc == SRT_OK) rc = srt_read_gathered(root, nullptr, argb_out);
	if (rc != SRT_OK) return gfail(g, rc, srt_last_error(root));
	for (size_t i = 1; i < g->t.size(); i++) { // every device has finished its part before the call returns
		rc = srt_synchronize(g->t[i]);
		if (rc != SRT_OK) return gfail(g, rc, srt_last_error(g->t[i]));
	}
	return SRT_OK;
}

int srt_group_read_canvas(srt_group *g, float *rgba_out) {
	if (!g) return SRT_ERR_INVALID;
	const int rc = srt_read_gathered(g->t[0], rgba_out, nullptr);
	return rc == SRT_OK ? SRT_OK : gfail(g, rc, srt_last_error(g->t[0]));
}

int srt_group_get_counters(srt_group *g, srt_counters *out) {
	if (!g || !out) return SRT_ERR_INVALID;
	memset(out, 0, sizeof *out);
	for (srt_tracer *t : g->t) {
		srt_counters c;
		const int rc = srt_get_counters(t, &c);
		if (rc != SRT_OK) return gfail(g, rc, srt_last_error(t));
		out->paths += c.paths, out->rays += c.rays, out->sky += c.sky, out->tri_tests += c.tri_tests, out->tri_pass_u += c.tri_pass_u;
		out->nan_pixels += c.nan_pixels, out->watchdog += c.watchdog;
	}
	return SRT_OK;
}

// ---------------------------------------------------------------------------------
// frame pipeline: the interactive loop's render() with frame N's read-back under frame N+1's trace
// ---------------------------------------------------------------------------------
int srt_render_pipelined(srt_tracer *t, const srt_render_data *options, uint32_t ticks_stopped, uint8_t *argb_out, long long *frame_delivered) {
	if (!t) return SRT_ERR_INVALID;
	if (!argb_out) return fail(t, SRT_ERR_INVALID, "srt_render_pipelined: argb_out is NULL");
	SrtCollect *c = collect_of(t);
	if (!c) return fail(t, SRT_ERR_INVALID, "out of host memory");
	SRT_HIP(t, hipSetDevice(t->device));
	const size_t bytes = (size_t)t->owned_rows * t->width * 4;
	if (!c->copy_stream || c->frame_bytes != bytes) {
		if (c->frames_enqueued != c->frames_delivered) return fail(t, SRT_ERR_STATE, "srt_render_pipelined: partition changed with a frame in flight (srt_pipeline_flush first)");
		if (!c->copy_stream) SRT_HIP(t, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
		for (int i = 0; i < 2; i++) {
			if (c->pinned[i]) (void)hipHostFree(c->pinned[i]);
			if (c->dev_argb[i]) (void)hipFree(c->dev_argb[i]);
			c->pinned[i] = c->dev_argb[i] = nullptr;
			SRT_HIP(t, hipHostMalloc(reinterpret_cast<void **>(&c->pinned[i]), bytes ? bytes : 4, hipHostMallocDefault));
			SRT_HIP(t, hipMalloc(reinterpret_cast<void **>(&c->dev_argb[i]), bytes ? bytes : 4));
			if (!c->resolved[i]) SRT_HIP(t, hipEventCreateWithFlags(&c->resolved[i], hipEventDisableTiming));
			if (!c->copied[i]) SRT_HIP(t, hipEventCreateWithFlags(&c->copied[i], hipEventDisableTiming));
		}
		c->frame_bytes = bytes;
	}
	// enqueue frame N: trace + resolve into this frame's own ARGB buffer on the handle's stream, its copy on the copy stream
	const int slot = (int)(c->frames_enqueued & 1);
	int rc = srt_trace_fused(t, options, c->dev_argb[slot], ticks_stopped); // the last reduction resolves into this frame's own image
	if (rc != SRT_OK) return rc;
	SRT_HIP(t, hipEventRecord(c->resolved[slot], t->stream));
	SRT_HIP(t, hipStreamWaitEvent(c->copy_stream, c->resolved[slot], 0));
	SRT_HIP(t, hipMemcpyAsync(c->pinned[slot], c->dev_argb[slot], bytes, hipMemcpyDeviceToHost, c->copy_stream));
	SRT_HIP(t, hipEventRecord(c->copied[slot], c->copy_stream));
	c->frames_enqueued++;
	// deliver frame N-1 (its trace, resolve and copy ran while the host prepared frame N)
	if (frame_delivered) *frame_delivered = -1;
	if (c->frames_enqueued - c->frames_delivered == 2) {
		const int prev = (int)(c->frames_delivered & 1);
		SRT_HIP(t, hipEventSynchronize(c->copied[prev]));
		memcpy(argb_out, c->pinned[prev], bytes);
		if (frame_delivered) *frame_delivered = c->frames_delivered;
		c->frames_delivered++;
	}
	return SRT_OK;
}

int srt_pipeline_flush(srt_tracer *t, uint8_t *argb_out, long long *frame_delivered) {
	if (!t) return SRT_ERR_INVALID;
	SrtCollect *c = t->collect;
	if (frame_delivered) *frame_delivered = -1;
	if (!c || c->frames_enqueued == c->frames_delivered) return SRT_OK;
	SRT_HIP(t, hipSetDevice(t->device));
	while (c->frames_enqueued != c->frames_delivered) { // at most two; the caller gets the newest
		const int prev = (int)(c->frames_delivered & 1);
		SRT_HIP(t, hipEventSynchronize(c->copied[prev]));
		if (argb_out) memcpy(argb_out, c->pinned[prev], c->frame_bytes);
		if (frame_delivered) *frame_delivered = c->frames_delivered;
		c->frames_delivered++;
	}
	return SRT_OK;
}

} // extern "C"
