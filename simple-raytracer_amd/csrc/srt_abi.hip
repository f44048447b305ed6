// srt_abi.hip — host side of libsrt_hip.so: the C ABI of include/srt_abi.h over the
// kernels in kernels.hip. This is what replaces the boost.compute/OpenCL dispatch of
// the reference's Tracer (/root/reference/src/tracer.cpp:11-116). HIP runtime only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
#include <new>
#include <string>
#include <system_error>
#include <vector>

#include "../../include/srt_abi.h"
#include "detmath.h"
#include "device_types.h"

#include "srt_internal.h"

#ifndef SRT_OVERLAP_BATCHES
#define SRT_OVERLAP_BATCHES 1 // srt_trace: consecutive sample batches on two streams (0: one after the other, A/B only)
#endif
#ifndef SRT_SCAN_PAIRS_PER_LAUNCH
#define SRT_SCAN_PAIRS_PER_LAUNCH 8e12 // srt_trace: ray-triangle pairs one array-scan launch may come to (sample batches)
#endif

namespace {

thread_local std::string g_create_error;

int num_blocks(int height, int rpb) { return (height + rpb - 1) / rpb; }

// ---- BVH builder (host; layout in device_types.h) --------------------
// Top-down, binned surface-area heuristic (16 bins, all three axes), leaves of at most
// SRT_BVH_LEAF_MAX triangles, nodes emitted in depth-first order with skip links. Runs once
// per srt_update_scene and model instance; ~35 ms for 10^5 triangles on one host core.
struct BvhBuilder {
	struct Tri {
		float lo[3], hi[3], c[3];
		uint32_t j;
	};
	std::vector<Tri> tris;
	std::vector<BvhNode> &nodes;
	std::vector<uint32_t> &order;
	uint32_t rec_base = 0;
	uint32_t leaves = 0, max_depth = 0;
	uint32_t sah_depth = 48; // below this depth: median splits (0 = a balanced tree, see fold_wide's stack bound)

	BvhBuilder(std::vector<BvhNode> &n, std::vector<uint32_t> &o) : nodes(n), order(o) {}

	static float half_area(const float lo[3], const float hi[3]) {
		const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
		return dx * dy + dy * dz + dz * dx;
	}

	// World-space vertices as the pre-pass kernel computes them (render.cl:114-120 order); boxes
	// are padded by 2^-12 of the model's diagonal: a hit the float Moller-Trumbore test accepts
	// lies within rounding error of its triangle, and must still be inside every box above it.
	void load(const srt_model &m, const srt_triangle *all) {
		const uint32_t n = m.num_triangles;
		tris.resize(n);
		auto xf = [&](const srt_float3 &v, float out[3]) {
			const srt_float4 *t = m.transform;
			out[0] = ((t[0].x * v.x + t[1].x * v.y) + t[2].x * v.z) + t[3].x * 1.0f;
			out[1] = ((t[0].y * v.x + t[1].y * v.y) + t[2].y * v.z) + t[3].y * 1.0f;
			out[2] = ((t[0].z * v.x + t[1].z * v.y) + t[2].z * v.z) + t[3].z * 1.0f;
		};
		float mlo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mhi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
		for (uint32_t j = 0; j < n; j++) {
			const srt_triangle &tr = all[m.triangle_index + j];
			float p[3][3];
			for (int k = 0; k < 3; k++) xf(tr.vertices[k].pos, p[k]);
			Tri &t = tris[j];
			t.j = j;
			bool finite = true;
			for (int a = 0; a < 3; a++) {
				// the kernel's triangle is (p0, p0 + e1, p0 + e2) with e = p_k - p0 rounded: cover both
				const float q1 = p[0][a] + (p[1][a] - p[0][a]), q2 = p[0][a] + (p[2][a] - p[0][a]);
				t.lo[a] = std::min(std::min(std::min(p[0][a], p[1][a]), std::min(p[2][a], q1)), q2);
				t.hi[a] = std::max(std::max(std::max(p[0][a], p[1][a]), std::max(p[2][a], q1)), q2);
				finite = finite && std::isfinite(t.lo[a]) && std::isfinite(t.hi[a]);
			}
			if (!finite) { // hostile input: a box that every ray enters, so the triangle is always tested
				for (int a = 0; a < 3; a++) t.lo[a] = -FLT_MAX, t.hi[a] = FLT_MAX, t.c[a] = 0.0f;
				continue;
			}
			for (int a = 0; a < 3; a++) {
				t.c[a] = 0.5f * t.lo[a] + 0.5f * t.hi[a];
				mlo[a] = std::min(mlo[a], t.lo[a]);
				mhi[a] = std::max(mhi[a], t.hi[a]);
			}
		}
		double d2 = 0.0;
		for (int a = 0; a < 3; a++)
			if (mhi[a] >= mlo[a]) d2 += ((double)mhi[a] - mlo[a]) * ((double)mhi[a] - mlo[a]);
		const float pad = (float)std::min(std::sqrt(d2) * (1.0 / 4096.0), (double)FLT_MAX);
		for (Tri &t : tris)
			for (int a = 0; a < 3; a++) {
				if (t.lo[a] == -FLT_MAX) continue;
				// widen by pad plus two ulps (the slab arithmetic rounds, too); stays finite
				t.lo[a] = std::max(-FLT_MAX, std::nextafter(std::nextafter(t.lo[a] - pad, -INFINITY), -INFINITY));
				t.hi[a] = std::min(FLT_MAX, std::nextafter(std::nextafter(t.hi[a] + pad, INFINITY), INFINITY));
			}
	}

	// Subtree over tris[b, e) appended to `out` (indices inside `out`; a node's skip = the index behind its subtree, which
	// for the subtree's last nodes is out.size() at return). The two halves of a large range are built by two threads into
	// vectors of their own and appended in order -- same nodes in the same order as the one-thread build, the ranges of
	// `tris` the threads partition are disjoint -- down to `par` levels: 10^5 triangles 42 -> 13 ms on the GPU box's cores.
	struct Stats {
		uint32_t leaves = 0, max_depth = 0;
	};
	Stats build_into(std::vector<BvhNode> &out, uint32_t b, uint32_t e, uint32_t depth, int par) {
		Stats st;
		const uint32_t self = (uint32_t)out.size();
		out.emplace_back();
		st.max_depth = depth;
		float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
		float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
		for (uint32_t i = b; i < e; i++)
			for (int a = 0; a < 3; a++) {
				lo[a] = std::min(lo[a], tris[i].lo[a]), hi[a] = std::max(hi[a], tris[i].hi[a]);
				clo[a] = std::min(clo[a], tris[i].c[a]), chi[a] = std::max(chi[a], tris[i].c[a]);
			}
		for (int a = 0; a < 3; a++) out[self].lo[a] = lo[a], out[self].hi[a] = hi[a];
		const uint32_t n = e - b;
		if (n <= SRT_BVH_LEAF_MAX) {
			out[self].leaf = (n << 28) | (rec_base + b);
			out[self].skip = (uint32_t)out.size();
			st.leaves = 1;
			return st;
		}
		// binned SAH over the three axes
		constexpr int NB = 16;
		int best_axis = -1, best_bin = 0;
		float best_cost = INFINITY;
		if (depth < sah_depth) {
			for (int a = 0; a < 3; a++) {
				const float ext = chi[a] - clo[a];
				if (!(ext > 0.0f) || !std::isfinite(ext)) continue;
				const float scale = (float)NB / ext;
				uint32_t cnt[NB] = {0};
				float blo[NB][3], bhi[NB][3];
				for (int k = 0; k < NB; k++)
					for (int c = 0; c < 3; c++) blo[k][c] = FLT_MAX, bhi[k][c] = -FLT_MAX;
				for (uint32_t i = b; i < e; i++) {
					int k = (int)((tris[i].c[a] - clo[a]) * scale);
					k = k < 0 ? 0 : (k >= NB ? NB - 1 : k);
					cnt[k]++;
					for (int c = 0; c < 3; c++) blo[k][c] = std::min(blo[k][c], tris[i].lo[c]), bhi[k][c] = std::max(bhi[k][c], tris[i].hi[c]);
				}
				float rarea[NB];
				uint32_t rcnt[NB];
				float rl[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, rh[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
				uint32_t rc = 0;
				for (int k = NB - 1; k > 0; k--) {
					for (int c = 0; c < 3; c++) rl[c] = std::min(rl[c], blo[k][c]), rh[c] = std::max(rh[c], bhi[k][c]);
					rc += cnt[k];
					rarea[k] = rc ? half_area(rl, rh) : 0.0f;
					rcnt[k] = rc;
				}
				float ll[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lh[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
				uint32_t lc = 0;
				for (int k = 0; k < NB - 1; k++) { // split after bin k
					for (int c = 0; c < 3; c++) ll[c] = std::min(ll[c], blo[k][c]), lh[c] = std::max(lh[c], bhi[k][c]);
					lc += cnt[k];
					if (lc == 0 || rcnt[k + 1] == 0) continue;
					const float cost = half_area(ll, lh) * (float)lc + rarea[k + 1] * (float)rcnt[k + 1];
					if (cost < best_cost) best_cost = cost, best_axis = a, best_bin = k;
				}
			}
		}
		uint32_t mid;
		if (best_axis >= 0) {
			const int a = best_axis;
			const float scale = (float)NB / (chi[a] - clo[a]);
			const float c0 = clo[a];
			auto it = std::partition(tris.begin() + b, tris.begin() + e, [&](const Tri &t) {
				int k = (int)((t.c[a] - c0) * scale);
				k = k < 0 ? 0 : (k >= NB ? NB - 1 : k);
				return k <= best_bin;
			});
			mid = (uint32_t)(it - tris.begin());
		} else {
			mid = b; // no usable split (coincident centroids, overflow, depth cap): halve by index
		}
		if (mid == b || mid == e) {
			int a = 0;
			for (int c = 1; c < 3; c++)
				if (chi[c] - clo[c] > chi[a] - clo[a]) a = c;
			mid = b + n / 2;
			std::nth_element(tris.begin() + b, tris.begin() + mid, tris.begin() + e, [a](const Tri &x, const Tri &y) { return x.c[a] < y.c[a]; });
		}
		out[self].leaf = 0;
		Stats sl, sr;
		if (par > 0 && n >= 8192) {
			std::vector<BvhNode> left, right;
			std::future<Stats> fut;
			try {
				fut = std::async(std::launch::async, [&] { return build_into(left, b, mid, depth + 1, par - 1); });
			} catch (const std::system_error &) { // no thread to be had: this one does both halves
			}
			sr = build_into(right, mid, e, depth + 1, par - 1);
			sl = fut.valid() ? fut.get() /* (rethrows what the other thread threw) */ : build_into(left, b, mid, depth + 1, 0);
			for (std::vector<BvhNode> *sub : {&left, &right}) {
				const uint32_t off = (uint32_t)out.size();
				out.insert(out.end(), sub->begin(), sub->end());
				for (size_t i = off; i < out.size(); i++) out[i].skip += off;
			}
		} else {
			sl = build_into(out, b, mid, depth + 1, 0);
			sr = build_into(out, mid, e, depth + 1, 0);
		}
		out[self].skip = (uint32_t)out.size();
		st.leaves = sl.leaves + sr.leaves;
		st.max_depth = std::max(sl.max_depth, sr.max_depth);
		return st;
	}
	uint32_t build(uint32_t b, uint32_t e, uint32_t depth) {
		const uint32_t self = (uint32_t)nodes.size();
		const Stats st = build_into(nodes, b, e, depth, 3); // up to 8 subtrees at a time
		leaves += st.leaves;
		if (st.max_depth > max_depth) max_depth = st.max_depth;
		return self;
	}

	// New boxes for an existing topology (nodes relative to the model, `order` = triangle of each
	// record): the model moved but its triangles did not. Children follow their parent in the
	// array, so one backward sweep has every child's box ready before its parent's. O(n).
	void refit(const srt_model &m, const srt_triangle *all) {
		load(m, all);
		const uint32_t n = (uint32_t)nodes.size();
		for (uint32_t i = n; i-- > 0;) {
			BvhNode &nd = nodes[i];
			float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
			if (nd.leaf) {
				const uint32_t first = nd.leaf & 0x0fffffffu, cnt = nd.leaf >> 28;
				for (uint32_t r = first; r < first + cnt; r++) {
					const Tri &t = tris[order[r]]; // load() leaves tris in triangle order
					for (int a = 0; a < 3; a++) lo[a] = std::min(lo[a], t.lo[a]), hi[a] = std::max(hi[a], t.hi[a]);
				}
			} else {
				const uint32_t left = i + 1, right = nodes[left].skip == SRT_BVH_END ? n : nodes[left].skip;
				for (uint32_t c : {left, right})
					for (int a = 0; a < 3; a++) lo[a] = std::min(lo[a], nodes[c].lo[a]), hi[a] = std::max(hi[a], nodes[c].hi[a]);
			}
			for (int a = 0; a < 3; a++) nd.lo[a] = lo[a], nd.hi[a] = hi[a];
		}
	}

	// The wide form of a hierarchy (device_types.h): `c` is the canonical binary form build() leaves (depth-first, the left
	// child directly behind its parent, the right one at the left one's skip link; indices and records relative to the
	// model). An inner block takes the two children of a node and then, while it has room, replaces the child with the
	// largest box by that child's own two (`balanced`: the node's grandchildren, level by level); leaves become leaf
	// blocks. Blocks are appended to `out` (32 dwords each, block indices relative to `out`'s start = the model's first
	// block), `inner` lists the inner ones (their references are shifted when the model is placed in the scene), dest[r]
	// = (leaf block << 2) | slot of record r. Returns the root reference and in `need` the most entries a walk can have
	// waiting at once: every block on the way down leaves at most (children - 1) behind.
	struct Wide {
		std::vector<uint32_t> blocks;
		std::vector<uint32_t> inner;
		std::vector<uint32_t> dest;
		uint32_t root = SRT_BVH_NONE, need = 0;
		struct Job { // an inner block whose boxes are still to be quantised (fold_wide does them on several threads)
			uint32_t self, kids[4], nk, tags, first;
		};
		std::vector<Job> jobs;
	};
	// The boxes of an inner block as bytes (device_types.h): bound = fmaf(q, 2^e, origin) per axis, rounded outwards and CHECKED in
	// exactly that arithmetic, the grid coarsened until every upper bound fits a byte. Non-finite extents (hostile input in
	// all-embracing boxes) end on the coarsest grid, where upper bounds overflow to +inf: still a box that contains the child.
	template <class Fma>
	static inline __attribute__((always_inline)) void quantise_body(const std::vector<BvhNode> &c, const uint32_t *kids, uint32_t nk, uint32_t tags, uint32_t first, uint32_t *blk, Fma fmaf_any) {
		float origin[3];
		uint32_t expo[3];
		uint8_t qlo[3][4], qhi[3][4];
		auto pow2 = [](int e) { // 2^e as a float, -126 <= e <= 127
			const uint32_t bits = (uint32_t)(e + 127) << 23;
			float f;
			memcpy(&f, &bits, 4);
			return f;
		};
		for (int a = 0; a < 3; a++) {
			origin[a] = FLT_MAX;
			float top = -FLT_MAX;
			for (uint32_t k = 0; k < nk; k++) origin[a] = std::min(origin[a], c[kids[k]].lo[a]), top = std::max(top, c[kids[k]].hi[a]);
			if (!(origin[a] == origin[a])) origin[a] = -FLT_MAX; // (NaN boxes of hostile input)
			int e = -126;
			const float extent = top - origin[a];
			if (extent > 0.0f) {
				int ex = 0;
				(void)std::frexp(extent / 255.0f, &ex); // extent / 255 = m * 2^ex, m in [0.5, 1): 2^ex is the first power of two above it
				e = std::isfinite(extent) ? ex : 126;
			}
			for (;; e++) {
				if (e < -126) e = -126;
				if (e > 127) e = 127;
				const float scale = pow2(e);
				const double inv_scale = std::ldexp(1.0, -e); // (exact; in double so that 2^126 has a reciprocal)
				bool fits = true;
				for (uint32_t k = 0; k < nk && fits; k++) {
					const float lo = c[kids[k]].lo[a], hi = c[kids[k]].hi[a];
					const double fl = ((double)lo - (double)origin[a]) * inv_scale; // >= 0: origin is the smallest lo
					int ql = fl >= 255.0 ? 255 : (fl > 0.0 ? (int)fl : 0);       // (NaN: 0)
					while (ql > 0 && !(fmaf_any((float)ql, scale, origin[a]) <= lo)) ql--;
					const double fh = ((double)hi - (double)origin[a]) * inv_scale;
					int qh = fh > 255.0 ? 256 : (fh > 0.0 ? (int)fh + ((double)(int)fh < fh ? 1 : 0) : 0);
					if (!(fh == fh)) qh = 256;
					while (qh <= 255 && !(fmaf_any((float)qh, scale, origin[a]) >= hi)) qh++;
					if (qh > 255) fits = false;
					qlo[a][k] = (uint8_t)ql, qhi[a][k] = (uint8_t)(qh & 255);
				}
				if (fits || e == 127) { // (e == 127: 255 * 2^127 overflows every finite bound; keep what we have, q = 255 gives +inf)
					if (!fits)
						for (uint32_t k = 0; k < nk; k++) qhi[a][k] = 255;
					expo[a] = (uint32_t)(e + 127);
					break;
				}
			}
		}
		for (int a = 0; a < 3; a++) memcpy(&blk[a], &origin[a], 4);
		blk[3] = expo[0] | (expo[1] << 8) | (expo[2] << 16) | (nk << 24);
		for (int a = 0; a < 3; a++) {
			blk[4 + a] = 0u, blk[7 + a] = 0u;
			for (uint32_t k = 0; k < 4; k++) {
				blk[4 + a] |= (uint32_t)(k < nk ? qlo[a][k] : 255u) << (8 * k); // (an empty slot: lo above hi; the walk counts the slots)
				blk[7 + a] |= (uint32_t)(k < nk ? qhi[a][k] : 0u) << (8 * k);
			}
		}
		blk[10] = tags;
		blk[11] = first;
	}
	// The check wants fmaf as the device rounds it. glibc's fmaf is a call into a software path on some hosts (85 ns: 45 ms of a
	// 10^5-triangle hierarchy's 528,000 checks); a CPU with FMA does it in one instruction, inlined into a copy of the function.
	__attribute__((target("fma"))) static void quantise_hw(const std::vector<BvhNode> &c, const uint32_t *kids, uint32_t nk, uint32_t tags, uint32_t first, uint32_t *blk) {
		quantise_body(c, kids, nk, tags, first, blk, [](float a, float b, float x) __attribute__((target("fma"))) { return __builtin_fmaf(a, b, x); });
	}
	static void quantise(const std::vector<BvhNode> &c, const uint32_t *kids, uint32_t nk, uint32_t tags, uint32_t first, uint32_t *blk) {
		static const bool hw = __builtin_cpu_supports("fma");
		if (hw) quantise_hw(c, kids, nk, tags, first, blk);
		else quantise_body(c, kids, nk, tags, first, blk, [](float a, float b, float x) { return std::fmaf(a, b, x); });
	}
	// fills block `self` (already allocated) from node ci; returns the reference to it
	static uint32_t fold_node(const std::vector<BvhNode> &c, uint32_t ci, uint32_t self, bool balanced, Wide &w, uint32_t &need) {
		const BvhNode &nd = c[ci];
		if (nd.leaf) {
			const uint32_t first = nd.leaf & 0x0fffffffu, cnt = nd.leaf >> 28;
			for (uint32_t k = 0; k < cnt; k++) w.dest[first + k] = (self << 2) | k;
			need = 0;
			return SRT_BVH_LEAF_BIT | (cnt << 28) | self;
		}
		w.inner.push_back(self);
		uint32_t kids[4], nk = 0;
		const uint32_t left = ci + 1u, right = c[left].skip;
		kids[nk++] = left, kids[nk++] = right;
		if (balanced) {
			uint32_t g[4], ng = 0;
			for (uint32_t k = 0; k < 2; k++)
				if (c[kids[k]].leaf) g[ng++] = kids[k];
				else g[ng++] = kids[k] + 1u, g[ng++] = c[kids[k] + 1u].skip;
			nk = ng;
			for (uint32_t k = 0; k < ng; k++) kids[k] = g[k];
		} else {
			while (nk < 4) {
				int open = -1;
				float area = -1.0f;
				for (uint32_t k = 0; k < nk; k++) {
					if (c[kids[k]].leaf) continue;
					const float a = half_area(c[kids[k]].lo, c[kids[k]].hi);
					if (open < 0 || a > area) open = (int)k, area = a; // NaN / inf areas (hostile input) still pick somebody
				}
				if (open < 0) break;
				const uint32_t o = kids[open];
				kids[open] = o + 1u;
				kids[nk++] = c[o + 1u].skip;
			}
		}
		// the children's blocks lie side by side: the walk finds child k at first + k
		const uint32_t first = (uint32_t)(w.blocks.size() / 32);
		w.blocks.resize(w.blocks.size() + 32 * (size_t)nk, 0u);
		uint32_t deepest = 0;
		uint32_t tags = 0;
		for (uint32_t k = 0; k < 4; k++) {
			if (k >= nk) {
				tags |= k << (8 * k);
				continue;
			}
			uint32_t sub = 0;
			const uint32_t ref = fold_node(c, kids[k], first + k, balanced, w, sub); // may grow w.blocks: index, do not keep pointers
			if (sub > deepest) deepest = sub;
			tags |= SRT_BVH_TAG(ref, k) << (8 * k);
		}
		w.jobs.push_back({self, {kids[0], kids[1], nk > 2 ? kids[2] : 0u, nk > 3 ? kids[3] : 0u}, nk, tags, first});
		need = deepest + (nk - 1u);
		return self;
	}
	static void fold_wide(const std::vector<BvhNode> &c, uint32_t records, bool balanced, Wide &w) {
		w.blocks.clear(), w.inner.clear();
		w.dest.assign(records, 0u);
		w.root = SRT_BVH_NONE, w.need = 0;
		if (c.empty()) return;
		w.blocks.reserve(32 * c.size()); // (every node of the binary hierarchy becomes at most one block)
		w.inner.reserve(c.size() / 2 + 1), w.jobs.reserve(c.size() / 2 + 1);
		w.blocks.resize(32, 0u);
		w.jobs.clear();
		w.root = fold_node(c, 0u, 0u, balanced, w, w.need);
		// the blocks' boxes: every inner block by itself (reads the binary nodes, writes its own 48 bytes), large hierarchies on up
		// to eight threads (10^5 triangles: 22k blocks x 24 bounds rounded outwards and checked)
		uint32_t *blocks = w.blocks.data();
		const size_t nj = w.jobs.size();
		auto run = [&](size_t lo, size_t hi) {
			for (size_t i = lo; i < hi; i++) {
				const Wide::Job &j = w.jobs[i];
				quantise(c, j.kids, j.nk, j.tags, j.first, blocks + 32 * (size_t)j.self);
			}
		};
		const size_t parts = nj >= 4096 ? 8 : 1;
		std::vector<std::future<void>> futs;
		try {
			for (size_t t = 1; t < parts; t++) futs.push_back(std::async(std::launch::async, run, nj * t / parts, nj * (t + 1) / parts));
		} catch (const std::system_error &) { // no more threads: the rest is done here
		}
		run(0, nj / parts);
		for (size_t t = futs.size() + 1; t < parts; t++) run(nj * t / parts, nj * (t + 1) / parts);
		for (auto &f : futs) f.get();
		w.jobs.clear();
		w.jobs.shrink_to_fit();
	}

	// Appends the model's nodes and triangle order; returns the root's index.
	uint32_t run(const srt_model &m, const srt_triangle *all, uint32_t first_record) {
		rec_base = first_record;
		load(m, all);
		const uint32_t n0 = (uint32_t)nodes.size();
		build(0, (uint32_t)tris.size(), 1);
		const uint32_t n1 = (uint32_t)nodes.size();
		for (uint32_t i = n0; i < n1; i++)
			if (nodes[i].skip == n1) nodes[i].skip = SRT_BVH_END;
		for (const Tri &t : tris) order.push_back(t.j);
		return n0;
	}
};

// One model instance's hierarchy with indices relative to its own first node / first record, kept
// between srt_update_scene calls together with what it was built from: an edit that leaves a model's
// triangles and transform alone (camera, materials, sun, OTHER shapes) re-uses it instead of paying
// the build again (10^5 triangles: 37 ms -> 1.5 ms for the comparison).
struct BvhCacheEntry {
	uint32_t count = 0;
	uint64_t tri_hash = 0; // of the triangle bytes: looked at before any memcmp
	bool claimed = false;  // taken by a model of the srt_update_scene in progress
	srt_float4 transform[4];
	std::vector<srt_triangle> tris;
	std::vector<BvhNode> nodes;
	std::vector<uint32_t> order;
	BvhBuilder::Wide wide; // what the device walks, block indices relative to the model's first block
	bool balanced = false; // built without the SAH because the SAH tree could overflow a lane's stack
	uint32_t leaves = 0, depth = 0;
	// (re)builds nodes/order from the model and folds them; the fallback keeps every walk inside SRT_BVH_STACK_CAP
	void build(const srt_model &m, const srt_triangle *all) {
		for (int attempt = balanced ? 1 : 0; attempt < 2; attempt++) {
			nodes.clear(), order.clear();
			BvhBuilder bb(nodes, order);
			if (attempt) bb.sah_depth = 0;
			bb.run(m, all, 0u);
			leaves = bb.leaves, depth = bb.max_depth;
			balanced = attempt != 0;
			BvhBuilder::fold_wide(nodes, m.num_triangles, balanced, wide);
			if (wide.need <= SRT_BVH_STACK_CAP) break; // a balanced tree of < 2^28 triangles needs at most 3 * 15
		}
	}
	void refit(const srt_model &m, const srt_triangle *all) {
		BvhBuilder bb(nodes, order);
		bb.refit(m, all);
		BvhBuilder::fold_wide(nodes, m.num_triangles, balanced, wide);
		if (wide.need > SRT_BVH_STACK_CAP) build(m, all); // the new boxes fold differently: start over
	}
	bool same_triangles(const srt_model &m, const srt_triangle *all, uint64_t hash) const {
		return m.num_triangles == count && hash == tri_hash && memcmp(tris.data(), all + m.triangle_index, (size_t)count * sizeof(srt_triangle)) == 0;
	}
	bool same_transform(const srt_model &m) const { return memcmp(transform, m.transform, sizeof transform) == 0; }
};

// 64-bit FNV-1a over 8-byte words (records are 96 B)
uint64_t hash_triangles(const srt_triangle *tris, size_t count) {
	uint64_t h = 0xcbf29ce484222325ull;
	const size_t words = count * sizeof(srt_triangle) / 8;
	for (size_t i = 0; i < words; i++) {
		uint64_t w;
		memcpy(&w, reinterpret_cast<const char *>(tris) + 8 * i, 8);
		h = (h ^ w) * 0x100000001b3ull;
	}
	return h;
}

size_t owned_pixels(const srt_tracer *t) { return (size_t)t->owned_rows * (size_t)t->width; }

int clear_canvas_impl(srt_tracer *t) {
	// enqueue_fill_buffer with 0.0f over the whole canvas (src/tracer.cpp:98-101)
	SRT_HIP(t, hipMemsetAsync(t->canvas, 0, t->canvas_bytes, t->stream));
	return SRT_OK;
}

} // namespace

struct BvhCache {
	std::vector<BvhCacheEntry> entries;
};

int srt_fail(srt_tracer *t, int code, const std::string &msg) {
	if (t) t->err = msg;
	else g_create_error = msg;
	return code;
}


extern "C" {

const char *srt_version(void) {
#ifdef SRT_DEV_KNOBS
	return "srt-hip 0.1 gfx950 parity (fp-contract=off, IEEE div/sqrt, detmath) [dev knobs]";
#else
	return "srt-hip 0.1 gfx950 parity (fp-contract=off, IEEE div/sqrt, detmath)";
#endif
}

const char *srt_last_error(const srt_tracer *t) { return t ? t->err.c_str() : g_create_error.c_str(); }

int srt_partition_owned_rows(int height, int rank, int world, int rpb) {
	if (height < 0 || world < 1 || rank < 0 || rank >= world || rpb < 1) return -1;
	int rows = 0;
	const int nb = num_blocks(height, rpb);
	for (int b = rank; b < nb; b += world) {
		int r = height - b * rpb;
		rows += r < rpb ? r : rpb;
	}
	return rows;
}

int srt_partition_padded_rows(int height, int world, int rpb) {
	if (height < 0 || world < 1 || rpb < 1) return -1;
	const int nb = num_blocks(height, rpb);
	return ((nb + world - 1) / world) * rpb;
}

int srt_partition_global_row(int height, int rank, int world, int rpb, int local_row) {
	if (height < 0 || world < 1 || rank < 0 || rank >= world || rpb < 1 || local_row < 0) return -1;
	const int lb = local_row / rpb;
	const int y = (lb * world + rank) * rpb + (local_row - lb * rpb);
	return y < height ? y : -1;
}

int srt_partition_unpermute(const void *gathered, void *image, int height, int world, int rpb, size_t row_bytes) {
	if (!gathered || !image || height < 0 || world < 1 || rpb < 1) return SRT_ERR_INVALID;
	const int padded = srt_partition_padded_rows(height, world, rpb);
	const char *src = static_cast<const char *>(gathered);
	char *dst = static_cast<char *>(image);
	for (int r = 0; r < world; r++) {
		for (int lr = 0; lr < padded; lr++) {
			const int y = srt_partition_global_row(height, r, world, rpb, lr);
			if (y < 0) continue;
			memcpy(dst + (size_t)y * row_bytes, src + ((size_t)r * padded + lr) * row_bytes, row_bytes);
		}
	}
	return SRT_OK;
}

// Development knobs (scheduling experiments and tests that force rare paths: SRT_WAVES_PER_CU, SRT_SCAN_PAIRS, SRT_JOB_CAP_SUBS,
// SRT_ITEMS_PER_WAVE, SRT_FORCE_BATCH, SRT_POOL_BLOCKS, SRT_NO_SCAN_POOL) are read from the environment by -DSRT_DEV_KNOBS builds only (build.py build_dev():
// lib/variants/dev/). The product library's scheduling does not depend on the caller's environment.
#ifdef SRT_DEV_KNOBS
static const char *dev_env(const char *name) { return getenv(name); }
#else
static const char *dev_env(const char *) { return nullptr; }
#endif

int srt_create(int width, int height, int device_index, srt_tracer **out) {
	if (!out) return fail(nullptr, SRT_ERR_INVALID, "srt_create: out is NULL");
	*out = nullptr;
	if (width <= 0 || height <= 0) return fail(nullptr, SRT_ERR_INVALID, "srt_create: width and height must be positive");
	if ((uint64_t)width * (uint64_t)height >= (1ull << 31))
		return fail(nullptr, SRT_ERR_INVALID, "srt_create: width * height must stay below 2^31 pixels (the kernel's pixel id is 32-bit, as in render.cl:488)");
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0)
		return fail(nullptr, SRT_ERR_HIP, std::string("srt_create: no HIP device (") + hipGetErrorString(e) + ")");
	if (device_index < 0 || device_index >= ndev) return fail(nullptr, SRT_ERR_INVALID, "srt_create: device_index out of range");
	srt_tracer *t = new (std::nothrow) srt_tracer();
	if (!t) return fail(nullptr, SRT_ERR_INVALID, "srt_create: out of host memory");
	t->width = width;
	t->height = height;
	t->device = device_index;
	t->owned_rows = height;
	if (const char *env = getenv("SRT_RADIANCE_BUDGET_MB")) { // the one environment variable a deployment may want: read here, once
		const long long mb = atoll(env);
		if (mb > 0) t->radiance_budget_env = (size_t)mb << 20;
	}
	auto bail = [&](const char *what, hipError_t err) {
		std::string m = std::string("srt_create: ") + what + ": " + hipGetErrorString(err);
		srt_destroy(t);
		return fail(nullptr, SRT_ERR_HIP, m);
	};
	if ((e = hipSetDevice(device_index)) != hipSuccess) return bail("hipSetDevice", e);
	if ((e = hipStreamCreateWithFlags(&t->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
	{
		hipDeviceProp_t prop;
		if ((e = hipGetDeviceProperties(&prop, device_index)) != hipSuccess) return bail("hipGetDeviceProperties", e);
		t->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	}
	t->stream = t->own_stream;
	const size_t px = (size_t)width * height;
	if ((e = t->canvas_own.reserve(px * 4)) != hipSuccess) return bail("canvas alloc", e);
	t->canvas = t->canvas_own.ptr;
	t->canvas_bytes = px * 16;
	if ((e = t->argb.reserve(px * 4)) != hipSuccess) return bail("argb alloc", e);
	if ((e = t->counters.reserve(SRT_CTR_COUNT)) != hipSuccess) return bail("counter alloc", e);
	if ((e = t->wave_counters.reserve((size_t)2 * SRT_WAVE_CTR_SLOTS * SRT_WAVE_CTR_STRIDE)) != hipSuccess) return bail("counter alloc", e);
	if ((e = t->shapes.reserve(1)) != hipSuccess || (e = t->runs.reserve(1)) != hipSuccess ||
	    (e = t->run_data.reserve(32)) != hipSuccess || (e = t->winners.reserve(1)) != hipSuccess ||
	    (e = t->triangles.reserve(1)) != hipSuccess || (e = t->materials.reserve(1)) != hipSuccess ||
	    (e = t->wtris.reserve(SRT_WTRI_FLOATS + 64)) != hipSuccess || (e = t->wtri_offset.reserve(1)) != hipSuccess)
		return bail("scene alloc", e);
	if ((e = hipEventCreate(&t->ev_t0)) != hipSuccess || (e = hipEventCreate(&t->ev_t1)) != hipSuccess ||
	    (e = hipEventCreate(&t->ev_r0)) != hipSuccess || (e = hipEventCreate(&t->ev_r1)) != hipSuccess)
		return bail("hipEventCreate", e);
	if ((e = hipMemsetAsync(t->canvas, 0, t->canvas_bytes, t->stream)) != hipSuccess) return bail("canvas clear", e);
	if ((e = hipMemsetAsync(t->wave_counters.ptr, 0, (size_t)2 * SRT_WAVE_CTR_SLOTS * SRT_WAVE_CTR_STRIDE * sizeof(unsigned long long), t->stream)) != hipSuccess)
		return bail("counter clear", e);
	if ((e = hipMemsetAsync(t->counters.ptr, 0, SRT_CTR_COUNT * sizeof(unsigned long long), t->stream)) != hipSuccess)
		return bail("counter clear", e);
	if ((e = hipStreamSynchronize(t->stream)) != hipSuccess) return bail("sync", e);
	*out = t;
	return SRT_OK;
}

void srt_destroy(srt_tracer *t) {
	if (!t) return;
	(void)hipSetDevice(t->device);
	if (t->own_stream) (void)hipStreamSynchronize(t->own_stream);
	srt_collect_release(t);
	t->canvas_own.release();
	t->argb.release();
	t->shapes.release();
	t->runs.release();
	t->run_data.release();
	t->winners.release();
	t->triangles.release();
	t->materials.release();
	t->wtris.release();
	t->wtri_offset.release();
	t->bvh_blocks.release();
	t->bvh_order.release();
	t->bvh_dest.release();
	t->sky.release();
	t->counters.release();
	t->wave_counters.release();
	t->scan_queue.release();
	t->radiance.release();
	t->running.release();
	if (t->ev_t0) (void)hipEventDestroy(t->ev_t0);
	if (t->ev_t1) (void)hipEventDestroy(t->ev_t1);
	if (t->ev_r0) (void)hipEventDestroy(t->ev_r0);
	if (t->ev_r1) (void)hipEventDestroy(t->ev_r1);
	for (hipEvent_t ev : t->ev_k) (void)hipEventDestroy(ev);
	for (int k = 0; k < 2; k++) {
		if (t->ev_batch_traced[k]) (void)hipEventDestroy(t->ev_batch_traced[k]);
		if (t->ev_batch_reduced[k]) (void)hipEventDestroy(t->ev_batch_reduced[k]);
		if (t->batch_stream[k]) (void)hipStreamDestroy(t->batch_stream[k]);
	}
	if (t->ev_batch_fork) (void)hipEventDestroy(t->ev_batch_fork);
	if (t->own_stream) (void)hipStreamDestroy(t->own_stream);
	delete t->bvh_cache;
	delete t;
}

int srt_set_skybox(srt_tracer *t, const float *rgba, int width, int height) {
	if (!t) return SRT_ERR_INVALID;
	if (!rgba || width <= 0 || height <= 0) return fail(t, SRT_ERR_INVALID, "srt_set_skybox: bad image");
	SRT_HIP(t, hipSetDevice(t->device));
	const size_t n = (size_t)width * height * 4;
	SRT_HIP(t, hipStreamSynchronize(t->stream)); // kernels may still read the old image
	SRT_HIP(t, t->sky.reserve(n));
	SRT_HIP(t, hipMemcpyAsync(t->sky.ptr, rgba, n * sizeof(float), hipMemcpyHostToDevice, t->stream));
	SRT_HIP(t, hipStreamSynchronize(t->stream));
	t->sky_w = width;
	t->sky_h = height;
	return SRT_OK;
}

static int update_scene_impl(srt_tracer *t, const srt_shape *shapes, size_t n_shapes, const srt_triangle *triangles, size_t n_triangles,
                             const srt_material *materials, size_t n_materials, const srt_scene_data *scene);

int srt_update_scene(srt_tracer *t, const srt_shape *shapes, size_t n_shapes, const srt_triangle *triangles,
                     size_t n_triangles, const srt_material *materials, size_t n_materials,
                     const srt_scene_data *scene) {
	// the host pass allocates (std::vector): no C++ exception may cross the C ABI
	try {
		return update_scene_impl(t, shapes, n_shapes, triangles, n_triangles, materials, n_materials, scene);
	} catch (const std::bad_alloc &) {
		if (t) t->err.clear(); // the message itself must not allocate much: a short literal fits the small-string buffer
		return t ? fail(t, SRT_ERR_INVALID, "out of host memory") : SRT_ERR_INVALID;
	} catch (...) {
		return SRT_ERR_INVALID;
	}
}

// Everything a device needs of a scene, made on the host ONCE per srt_update_scene / srt_group_update_scene: shape blocks and group
// headers, winner records, world-triangle offsets, the hierarchy in its device form, the device material table (bernoulli()
// thresholds, Schlick constants). A group of N devices prepares one of these and uploads it N times (round 4; before, the whole
// host pass -- 18 ms of BVH build for 10^5 triangles -- ran once per device, in turn).
struct ScenePrep {
	std::vector<BlockGroup> groups;
	std::vector<float> data;
	std::vector<WinnerRec> winners;
	std::vector<uint32_t> offs, bvh_blocks, bvh_order, bvh_dest;
	std::vector<srt_material> dev_mats;
	uint64_t total_wtris = 0, max_tris = 0;
	int num_models = 0;
	bool use_bvh = false, unit_materials = false, all_materials_ok = true;
	int material_flags = 0;
	uint64_t bvh_info[7] = {0, 0, 0, 0, 0, 0, 0};
};

// host pass; `t` supplies the acceleration mode and owns the hierarchy cache (a group: its first member)
static int prepare_scene(srt_tracer *t, ScenePrep &sp, const srt_shape *shapes, size_t n_shapes, const srt_triangle *triangles, size_t n_triangles,
                         const srt_material *materials, size_t n_materials, const srt_scene_data *scene) {
	if (!t) return SRT_ERR_INVALID;
	if (!scene) return fail(t, SRT_ERR_INVALID, "srt_update_scene: scene is NULL");
	if ((n_shapes && !shapes) || (n_triangles && !triangles) || (n_materials && !materials))
		return fail(t, SRT_ERR_INVALID, "srt_update_scene: NULL array with non-zero count");
	if (n_shapes > 0x7fffffffu || n_triangles > 0xffffffffu) return fail(t, SRT_ERR_INVALID, "srt_update_scene: too many records");

	// Host pass: runs of same-type shapes in array order, their packed scalar-load records,
	// per-shape winner records and world-triangle offsets. The reference would read out of
	// bounds for a bad triangle range or material index; we refuse instead.
	std::vector<ShapeRun> runs;
	std::vector<float> &data = sp.data; // packed, every run starts on a 16-dword boundary
	std::vector<WinnerRec> &winners = sp.winners;
	winners.assign(n_shapes ? n_shapes : 1, WinnerRec());
	std::vector<uint32_t> &offs = sp.offs;
	offs.assign(n_shapes ? n_shapes : 1, 0u);
	uint64_t &total_wtris = sp.total_wtris, &max_tris = sp.max_tris;
	int &num_models = sp.num_models;
	const bool use_bvh = sp.use_bvh = t->accel_mode == SRT_ACCEL_BVH;
	std::vector<uint32_t> &bvh_blocks = sp.bvh_blocks; // 32 dwords each (device_types.h)
	std::vector<uint32_t> &bvh_order = sp.bvh_order, &bvh_dest = sp.bvh_dest;
	uint64_t bvh_leaves = 0, bvh_depth = 0, bvh_reused = 0, bvh_refitted = 0, bvh_canonical_nodes = 0;
	std::deque<BvhCacheEntry> fresh;                 // hierarchies built by this call (deque: growth keeps references valid)
	std::vector<std::pair<bool, size_t>> plan;       // per model with triangles: {from the cache?, index there / in fresh}
	std::vector<std::pair<uint64_t, uint64_t>> range_hashes; // {triangle_index << 32 | count, hash}
	if (use_bvh && !t->bvh_cache) t->bvh_cache = new BvhCache();
	if (t->bvh_cache)
		for (BvhCacheEntry &e : t->bvh_cache->entries) e.claimed = false;
	if (!use_bvh && t->bvh_cache) t->bvh_cache->entries.clear();
	const auto build_t0 = std::chrono::steady_clock::now();
	auto u2f = [](uint32_t u) {
		float f;
		memcpy(&f, &u, 4);
		return f;
	};
	// The kernel walks 64-byte BLOCKS of same-type shapes in array order (device_types.h ShapeRun): four spheres,
	// two planes or two models each, block b at dword 16 * b. A block that is not full is
	// filled up with records that can never be hit (r*r = -inf makes the discriminant -inf or NaN; a zero plane
	// normal makes denom == 0, render.cl:209-211).
	auto pad_run = [&]() {
		if (!runs.empty() && runs.back().type == SRT_SHAPE_SPHERE)
			while (data.size() % 16) data.insert(data.end(), {0.0f, 0.0f, 0.0f, -INFINITY});
		while (data.size() % 16) data.push_back(0.0f);
	};
	for (size_t i = 0; i < n_shapes; i++) {
		const srt_shape &s = shapes[i];
		WinnerRec &wr = winners[i];
		memset(&wr, 0, sizeof wr);
		wr.type = s.type;
		wr.material = s.material;
		if (s.material >= 0 && (size_t)s.material >= n_materials) {
			char buf[128];
			snprintf(buf, sizeof buf, "srt_update_scene: shape %zu uses material %d but only %zu exist", i, s.material, n_materials);
			return fail(t, SRT_ERR_INVALID, buf);
		}
		if (s.type != SRT_SHAPE_SPHERE && s.type != SRT_SHAPE_PLANE && s.type != SRT_SHAPE_MODEL) continue; // ignored, as render.cl:301-366
		const uint32_t block_cap = s.type == SRT_SHAPE_SPHERE ? 4u : 2u;
		// array scan: a big model sits alone in its block (data_off's top bit marks the block until the headers are built)
		const bool big_model = (!use_bvh || srt_bvh_suspends()) && s.type == SRT_SHAPE_MODEL && s.shape.model.num_triangles >= (uint32_t)srt_scan_suspend_min();
		const bool prev_big = !runs.empty() && (runs.back().data_off >> 31);
		if (runs.empty() || runs.back().type != s.type || runs.back().first_shape + runs.back().count != i || runs.back().count == block_cap || big_model || prev_big) {
			pad_run();
			ShapeRun r;
			r.type = s.type;
			r.first_shape = (uint32_t)i;
			r.count = 0;
			r.data_off = (uint32_t)data.size() | (big_model ? 0x80000000u : 0u); // = 16 * block number
			runs.push_back(r);
		}
		runs.back().count++;
		if (s.type == SRT_SHAPE_SPHERE) {
			const srt_sphere &sp = s.shape.sphere;
			data.insert(data.end(), {sp.position.x, sp.position.y, sp.position.z, sp.radius * sp.radius}); // r*r as render.cl:187
			wr.vx = sp.position.x, wr.vy = sp.position.y, wr.vz = sp.position.z, wr.w = sp.radius;
			{
				const float ar = fabsf(sp.radius);
				wr.inv_w = (ar >= 0x1p-40f && ar <= 0x1p40f) ? 1.0f / sp.radius : 0.0f; // (a NaN radius fails both compares)
			}
		} else if (s.type == SRT_SHAPE_PLANE) {
			const srt_plane &pl = s.shape.plane;
			data.insert(data.end(), {pl.position.x, pl.position.y, pl.position.z, 0.0f, pl.normal.x, pl.normal.y, pl.normal.z, 0.0f});
			wr.vx = pl.normal.x, wr.vy = pl.normal.y, wr.vz = pl.normal.z;
		} else {
			const srt_model &m = s.shape.model;
			if ((uint64_t)m.triangle_index + m.num_triangles > n_triangles) {
				char buf[160];
				snprintf(buf, sizeof buf, "srt_update_scene: shape %zu references triangles [%u, %u+%u) but only %zu exist", i,
				         m.triangle_index, m.triangle_index, m.num_triangles, n_triangles);
				return fail(t, SRT_ERR_INVALID, buf);
			}
			if (total_wtris + m.num_triangles > (use_bvh ? 0x0fffffffull : 0xffffffffull))
				return fail(t, SRT_ERR_INVALID, "srt_update_scene: too many world triangles");
			uint32_t link = use_bvh ? SRT_BVH_NONE : (uint32_t)total_wtris; // first world triangle of the model; BVH: root reference (NONE = nothing to walk)
			if (use_bvh && m.num_triangles > 0) {
				// hash of this triangle range, once per distinct range per call (instances share ranges)
				uint64_t th = 0;
				bool have_hash = false;
				for (const auto &rh : range_hashes)
					if (rh.first == (((uint64_t)m.triangle_index << 32) | m.num_triangles)) th = rh.second, have_hash = true;
				if (!have_hash) {
					th = hash_triangles(triangles + m.triangle_index, m.num_triangles);
					range_hashes.emplace_back(((uint64_t)m.triangle_index << 32) | m.num_triangles, th);
				}
				// An entry of the previous call with the same triangles: as it is when the transform did not
				// change either, otherwise refitted. Hash and transform are compared before any memcmp, and
				// entries only LEAVE the cache once the whole shape loop has validated (an early error return
				// keeps every hierarchy).
				BvhCacheEntry *kept = nullptr;
				for (BvhCacheEntry &e : t->bvh_cache->entries) {
					if (e.claimed || e.count != m.num_triangles || e.tri_hash != th) continue;
					const bool exact = e.same_transform(m);
					if (!exact && kept) continue; // already holding a refit candidate: only an exact match improves on it
					if (!e.same_triangles(m, triangles, th)) continue;
					kept = &e;
					if (exact) break;
				}
				BvhCacheEntry *ent;
				if (kept) {
					ent = kept;
					ent->claimed = true;
					if (ent->same_transform(m)) {
						bvh_reused++;
					} else { // the model moved: keep the topology, recompute the boxes
						ent->refit(m, triangles);
						memcpy(ent->transform, m.transform, sizeof ent->transform);
						bvh_refitted++;
					}
					plan.emplace_back(true, (size_t)(kept - t->bvh_cache->entries.data()));
				} else {
					fresh.emplace_back();
					ent = &fresh.back();
					ent->build(m, triangles);
					ent->count = m.num_triangles;
					ent->tri_hash = th;
					memcpy(ent->transform, m.transform, sizeof ent->transform);
					ent->tris.assign(triangles + m.triangle_index, triangles + m.triangle_index + m.num_triangles);
					plan.emplace_back(false, fresh.size() - 1);
				}
				// Indices inside an entry are relative to the model's first block / first record: shift them to where the
				// model lands in the scene's arrays.
				const uint32_t b0 = (uint32_t)(bvh_blocks.size() / 32), r0 = (uint32_t)total_wtris;
				const BvhBuilder::Wide &wd = ent->wide;
				if ((uint64_t)b0 + wd.blocks.size() / 32 > SRT_BVH_INDEX_MAX) return fail(t, SRT_ERR_INVALID, "srt_update_scene: too many BVH blocks");
				if (wd.need > SRT_BVH_STACK_CAP) return fail(t, SRT_ERR_INVALID, "srt_update_scene: BVH deeper than the walk's stack"); // unreachable below 2^28 triangles
				bvh_blocks.insert(bvh_blocks.end(), wd.blocks.begin(), wd.blocks.end());
				for (uint32_t ib : wd.inner) bvh_blocks[32 * (size_t)(b0 + ib) + SRT_BVH_FIRST_DWORD] += b0; // where the block's children lie
				bvh_dest.resize(r0 + (size_t)m.num_triangles);
				for (uint32_t r = 0; r < m.num_triangles; r++) bvh_dest[r0 + r] = wd.dest[r] + (b0 << 2);
				bvh_canonical_nodes += ent->nodes.size();
				bvh_order.insert(bvh_order.end(), ent->order.begin(), ent->order.end());
				link = wd.root == SRT_BVH_NONE ? SRT_BVH_NONE : wd.root + b0; // the root reference (a leaf reference for a model of <= 3 triangles)
				bvh_leaves += ent->leaves;
				if (ent->depth > bvh_depth) bvh_depth = ent->depth;
			}
			data.insert(data.end(), {m.bounding_min.x, m.bounding_min.y, m.bounding_min.z, u2f(link), m.bounding_max.x, m.bounding_max.y,
			                         m.bounding_max.z, u2f(use_bvh ? 0u : m.num_triangles)});
			wr.first_wtri = (uint32_t)total_wtris;
			offs[i] = (uint32_t)total_wtris;
			// brute force: blocks of 4, the tail stays all-zero (never hit); BVH: records are addressed one by one
			total_wtris += use_bvh ? (uint64_t)m.num_triangles : (((uint64_t)m.num_triangles + 3u) & ~3ull);
			if (m.num_triangles > max_tris) max_tris = m.num_triangles;
			num_models++;
		}
	}
	pad_run();
	// group headers: three blocks each (device_types.h BlockGroup); the data of a last, partial group is zero-filled
	std::vector<BlockGroup> &groups = sp.groups;
	groups.assign((runs.size() + 2) / 3, BlockGroup());
	uint32_t n_big = 0; // big model number k waits in scan stack k & 1 (kernels.hip)
	for (size_t b = 0; b < runs.size(); b++) {
		BlockGroup &g = groups[b / 3];
		if (b % 3 == 0) memset(&g, 0, sizeof g);
		const uint32_t big = runs[b].data_off >> 31;
		g.code |= (((uint32_t)runs[b].type + 1u) | (runs[b].count << 2) | (big << 5) | ((big ? (n_big++ & 1u) : 0u) << 6)) << (8 * (b % 3));
		g.first[b % 3] = runs[b].first_shape;
	}
	data.resize(groups.size() * 48 + 16, 0.0f);
	const uint64_t build_us =
	    (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - build_t0).count();

	// Device-side material table = the caller's records with three per-material constants
	// parked in padding floats, so the glass branch does no division at all:
	//   _pad[0] (offset 24) = 1.0f / refraction_index                      (render.cl:442, front face)
	//   _pad[1] (offset 28) = Schlick r0 for mu = 1/refraction_index       (render.cl:174-175, fp64 then float)
	//   color._pad (offset 44) = Schlick r0 for mu = refraction_index      (back face)
	// Same IEEE operations the kernel would run per hit, hence the same bits.
	std::vector<srt_material> &dev_mats = sp.dev_mats;
	dev_mats.assign(materials, materials + n_materials);
	auto schlick_r0 = [](float mu) {
		float r0 = (float)((1.0 - (double)mu) / (1.0 + (double)mu));
		return r0 * r0;
	};
	// bernoulli() thresholds (kernels.hip): T(p) = how many of the generator's 2^32 outputs r give p > (float)r * 2^-32 -- a prefix,
	// the conversion is monotone. When every probability of the scene has T < 2^32 (p <= 1 does) the device table carries the
	// thresholds' bits in place of metallic / specular / transmittance.
	auto threshold = [](float pr) -> uint64_t {
		uint64_t lo = 0, hi = (uint64_t)1 << 32; // first r in [lo, hi] for which !(pr > u(r)); hi = 2^32: none
		while (lo < hi) {
			const uint64_t mid = (lo + hi) >> 1;
			const float u = (float)(uint32_t)mid * 2.3283064365386963e-10f;
			if (pr > u) lo = mid + 1;
			else hi = mid;
		}
		return lo;
	};
	bool unit_materials = true;
	for (const auto &m : dev_mats)
		if (threshold(m.metallic) >> 32 || threshold(m.specular) >> 32 || threshold(m.transmittance) >> 32) unit_materials = false;
	sp.unit_materials = unit_materials;
	{
		// what holds for every material (SRT_MF_*): draws the scene decides are not made by the kernel
		bool no_specular = unit_materials, plain = true;
		for (const auto &m : dev_mats) {
			if (threshold(m.specular) != 0) no_specular = false;
			for (float c : {m.color.x, m.color.y, m.color.z})
				if (!std::isfinite(c) || (c == 0.0f && std::signbit(c))) plain = false;
		}
		sp.material_flags = (no_specular ? SRT_MF_NO_SPECULAR : 0) | (plain ? SRT_MF_PLAIN_COLORS : 0);
	}
	for (auto &m : dev_mats) {
		if (unit_materials) {
			const uint32_t tm = (uint32_t)threshold(m.metallic), ts = (uint32_t)threshold(m.specular), tt = (uint32_t)threshold(m.transmittance);
			memcpy(&m.metallic, &tm, 4), memcpy(&m.specular, &ts, 4), memcpy(&m.transmittance, &tt, 4);
		}
		const float inv_ior = 1.0f / m.refraction_index;
		m._pad[0] = inv_ior;
		m._pad[1] = schlick_r0(inv_ior);
		m.color._pad = schlick_r0(m.refraction_index);
	}
	sp.bvh_info[0] = bvh_canonical_nodes, sp.bvh_info[1] = bvh_leaves, sp.bvh_info[2] = bvh_depth, sp.bvh_info[3] = use_bvh ? build_us : 0;
	sp.bvh_info[4] = use_bvh ? plan.size() - bvh_reused - bvh_refitted : 0, sp.bvh_info[5] = bvh_reused, sp.bvh_info[6] = bvh_refitted;
	if (use_bvh) {
		std::vector<BvhCacheEntry> next_cache;
		next_cache.reserve(plan.size());
		for (const auto &pl : plan) next_cache.push_back(std::move(pl.first ? t->bvh_cache->entries[pl.second] : fresh[pl.second]));
		t->bvh_cache->entries = std::move(next_cache);
	}
	sp.all_materials_ok = true;
	for (size_t i = 0; i < n_shapes; i++)
		if (shapes[i].material < 0) sp.all_materials_ok = false;
	return SRT_OK;
}

// device pass, first half: wait for the handle's previous launches, (re)allocate, enqueue every upload and the pre-pass on its stream
static int upload_scene_begin(srt_tracer *t, const ScenePrep &sp, const srt_shape *shapes, size_t n_shapes, const srt_triangle *triangles,
                              size_t n_triangles, size_t n_materials) {
	const std::vector<BlockGroup> &groups = sp.groups;
	const std::vector<float> &data = sp.data;
	const std::vector<WinnerRec> &winners = sp.winners;
	const std::vector<uint32_t> &offs = sp.offs, &bvh_blocks = sp.bvh_blocks, &bvh_order = sp.bvh_order, &bvh_dest = sp.bvh_dest;
	const std::vector<srt_material> &dev_mats = sp.dev_mats;
	const bool use_bvh = sp.use_bvh;
	const uint64_t total_wtris = sp.total_wtris, max_tris = sp.max_tris;
	const int num_models = sp.num_models;
	SRT_HIP(t, hipSetDevice(t->device));
	SRT_HIP(t, hipStreamSynchronize(t->stream)); // previous launches may still read the old scene
	SRT_HIP(t, t->shapes.reserve(n_shapes));
	SRT_HIP(t, t->runs.reserve(groups.size()));
	SRT_HIP(t, t->run_data.reserve(data.size()));
	SRT_HIP(t, t->winners.reserve(n_shapes));
	SRT_HIP(t, t->wtri_offset.reserve(n_shapes));
	SRT_HIP(t, t->triangles.reserve(n_triangles));
	SRT_HIP(t, t->materials.reserve(n_materials));
	if (use_bvh) {
		SRT_HIP(t, t->bvh_blocks.reserve(bvh_blocks.size()));
		SRT_HIP(t, t->bvh_order.reserve(bvh_order.size()));
		SRT_HIP(t, t->bvh_dest.reserve(bvh_dest.size()));
		if (!bvh_blocks.empty()) // inner blocks complete, leaf blocks zero: srt_prepass_kernel writes their triangles
			SRT_HIP(t, hipMemcpyAsync(t->bvh_blocks.ptr, bvh_blocks.data(), bvh_blocks.size() * sizeof(uint32_t), hipMemcpyHostToDevice, t->stream));
		if (!bvh_order.empty()) {
			SRT_HIP(t, hipMemcpyAsync(t->bvh_order.ptr, bvh_order.data(), bvh_order.size() * sizeof(uint32_t), hipMemcpyHostToDevice, t->stream));
			SRT_HIP(t, hipMemcpyAsync(t->bvh_dest.ptr, bvh_dest.data(), bvh_dest.size() * sizeof(uint32_t), hipMemcpyHostToDevice, t->stream));
		}
	} else {
		SRT_HIP(t, t->wtris.reserve((size_t)total_wtris * SRT_WTRI_FLOATS + 64)); // + slack for the loop's look-ahead pair
	}
	if (n_shapes) {
		SRT_HIP(t, hipMemcpyAsync(t->shapes.ptr, shapes, n_shapes * sizeof(srt_shape), hipMemcpyHostToDevice, t->stream));
		SRT_HIP(t, hipMemcpyAsync(t->winners.ptr, winners.data(), n_shapes * sizeof(WinnerRec), hipMemcpyHostToDevice, t->stream));
		SRT_HIP(t, hipMemcpyAsync(t->wtri_offset.ptr, offs.data(), n_shapes * sizeof(uint32_t), hipMemcpyHostToDevice, t->stream));
	}
	if (!groups.empty())
		SRT_HIP(t, hipMemcpyAsync(t->runs.ptr, groups.data(), groups.size() * sizeof(BlockGroup), hipMemcpyHostToDevice, t->stream));
	SRT_HIP(t, hipMemcpyAsync(t->run_data.ptr, data.data(), data.size() * sizeof(float), hipMemcpyHostToDevice, t->stream));
	if (n_triangles)
		SRT_HIP(t, hipMemcpyAsync(t->triangles.ptr, triangles, n_triangles * sizeof(srt_triangle), hipMemcpyHostToDevice, t->stream));
	if (n_materials)
		SRT_HIP(t, hipMemcpyAsync(t->materials.ptr, dev_mats.data(), n_materials * sizeof(srt_material), hipMemcpyHostToDevice, t->stream));

	if (num_models > 0 && total_wtris > 0) {
		if (!use_bvh) SRT_HIP(t, hipMemsetAsync(t->wtris.ptr, 0, ((size_t)total_wtris * SRT_WTRI_FLOATS + 64) * sizeof(float), t->stream));
		// blockIdx.y = shape index; launch in slabs of 65535 shapes
		for (size_t base = 0; base < n_shapes; base += 65535) {
			PrepassParams pp;
			pp.shapes = t->shapes.ptr + base;
			pp.triangles = t->triangles.ptr;
			pp.wtri_offset = t->wtri_offset.ptr + base;
			pp.wtris = use_bvh ? reinterpret_cast<float *>(t->bvh_blocks.ptr) : t->wtris.ptr;
			pp.order = use_bvh ? t->bvh_order.ptr : nullptr;
			pp.dest = use_bvh ? t->bvh_dest.ptr : nullptr;
			size_t cnt = n_shapes - base;
			pp.num_shapes = (int32_t)(cnt > 65535 ? 65535 : cnt);
			pp.num_triangles = (uint32_t)n_triangles;
			srt_launch_prepass(pp, max_tris, t->stream);
		}
		SRT_HIP(t, hipGetLastError());
	}
	return SRT_OK;
}

// device pass, second half: the uploads have arrived (the host arrays are free again), the handle describes the new scene
static int upload_scene_end(srt_tracer *t, const ScenePrep &sp, size_t n_shapes, size_t n_materials, const srt_scene_data *scene) {
	SRT_HIP(t, hipSetDevice(t->device));
	SRT_HIP(t, hipStreamSynchronize(t->stream));
	t->sd = *scene;
	t->sd.num_shapes = (int32_t)n_shapes; // src/tracer.cpp:94
	t->num_models = sp.num_models;
	t->scan_tris = sp.use_bvh ? 0 : sp.total_wtris;
	t->bvh_active = sp.use_bvh && sp.num_models > 0;
	for (int k = 0; k < 7; k++) t->bvh_info[k] = sp.bvh_info[k];
	t->all_materials_ok = sp.all_materials_ok;
	t->unit_materials = sp.unit_materials;
	t->material_flags = sp.material_flags;
	t->num_runs = (int)sp.groups.size();
	t->num_materials = n_materials;
	t->scene_set = true;
	return SRT_OK;
}


static int update_scene_impl(srt_tracer *t, const srt_shape *shapes, size_t n_shapes, const srt_triangle *triangles, size_t n_triangles,
                             const srt_material *materials, size_t n_materials, const srt_scene_data *scene) {
	if (!t) return SRT_ERR_INVALID;
	ScenePrep sp;
	int rc = prepare_scene(t, sp, shapes, n_shapes, triangles, n_triangles, materials, n_materials, scene);
	if (rc == SRT_OK) rc = upload_scene_begin(t, sp, shapes, n_shapes, triangles, n_triangles, n_materials);
	if (rc == SRT_OK) rc = upload_scene_end(t, sp, n_shapes, n_materials, scene);
	return rc;
}

// One scene for N handles (srt_group_update_scene): prepared once with members[0]'s acceleration mode and hierarchy cache, uploaded to
// all of them -- every device's copies and pre-pass are enqueued before the first is waited for. On an error the members that
// were not reached keep their previous scene.
int srt_update_scene_many(srt_tracer *const *members, size_t n_members, const srt_shape *shapes, size_t n_shapes, const srt_triangle *triangles,
                          size_t n_triangles, const srt_material *materials, size_t n_materials, const srt_scene_data *scene, size_t *failed_member) {
	if (failed_member) *failed_member = 0;
	if (!members || n_members == 0 || !members[0]) return SRT_ERR_INVALID;
	try {
		ScenePrep sp;
		int rc = prepare_scene(members[0], sp, shapes, n_shapes, triangles, n_triangles, materials, n_materials, scene);
		if (rc != SRT_OK) return rc;
		size_t begun = 0;
		for (; begun < n_members && rc == SRT_OK; begun++) {
			if (failed_member) *failed_member = begun;
			members[begun]->accel_mode = members[0]->accel_mode; // (the prepared scene is in this form)
			rc = upload_scene_begin(members[begun], sp, shapes, n_shapes, triangles, n_triangles, n_materials);
		}
		if (rc != SRT_OK) { // what was enqueued on the members before the failing one still reads the host arrays: let it finish
			for (size_t i = 0; i + 1 < begun; i++) (void)upload_scene_end(members[i], sp, n_shapes, n_materials, scene);
			return rc;
		}
		for (size_t i = 0; i < n_members; i++) {
			if (failed_member) *failed_member = i;
			const int r = upload_scene_end(members[i], sp, n_shapes, n_materials, scene);
			if (r != SRT_OK && rc == SRT_OK) rc = r;
		}
		return rc;
	} catch (const std::bad_alloc &) {
		return fail(members[0], SRT_ERR_INVALID, "out of host memory");
	} catch (...) {
		return SRT_ERR_INVALID;
	}
}

int srt_clear_canvas(srt_tracer *t) {
	if (!t) return SRT_ERR_INVALID;
	SRT_HIP(t, hipSetDevice(t->device));
	return clear_canvas_impl(t);
}

int srt_trace(srt_tracer *t, const srt_render_data *options) { return srt_trace_fused(t, options, nullptr, 0u); }

// srt_trace; with fused_argb != NULL the last reduction also resolves every pixel it has just accumulated into fused_argb
// (owned pixels x 4 bytes, device memory) with the divisor ticks_stopped: what srt_resolve would do in a launch of its own
int srt_trace_fused(srt_tracer *t, const srt_render_data *options, uint8_t *fused_argb, uint32_t ticks_stopped) {
	if (!t) return SRT_ERR_INVALID;
	if (!options) return fail(t, SRT_ERR_INVALID, "srt_trace: options is NULL");
	if (options->width != t->width || options->height != t->height)
		return fail(t, SRT_ERR_INVALID, "srt_trace: options width/height differ from the handle's (no resize, tracer.hpp:61-66)");
	if (t->sky_w <= 0) return fail(t, SRT_ERR_STATE, "srt_trace: no skybox set (srt_set_skybox)");
	SRT_HIP(t, hipSetDevice(t->device));
	TraceParams p;
	memset(&p, 0, sizeof p);
	p.rd = *options;
	if (t->scene_set) {
		p.sd = t->sd;
	} else {
		memset(&p.sd, 0, sizeof p.sd); // kernel arg 1 never set: behave as the empty scene
	}
	p.runs = t->runs.ptr;
	p.run_data = t->run_data.ptr;
	p.winners = t->winners.ptr;
	p.num_runs = t->scene_set ? t->num_runs : 0;
	p.num_materials = t->scene_set ? (int32_t)t->num_materials : 0;
	p.shapes = t->shapes.ptr;
	p.triangles = t->triangles.ptr;
	p.materials = t->materials.ptr;
	p.wtris = t->wtris.ptr;
	p.sky = t->sky.ptr;
	p.canvas = t->canvas;
	p.counters = t->counters.ptr;
	p.wave_counters = t->wave_counters.ptr;
	p.sky_w = t->sky_w;
	p.sky_h = t->sky_h;
	p.f_width = (float)options->width;
	p.f_height = (float)options->height;
	p.inv_f_width = 1.0f / p.f_width; // IEEE quotients: the kernel's camera rays divide by multiplying with them (kernels.hip div_by_rcp)
	p.inv_f_height = 1.0f / p.f_height;
	srt_magic_u31(options->width > 0 ? (uint32_t)options->width : 1u, &p.width_magic, &p.width_shift);
	srt_magic_u31(t->rows_per_block > 0 ? (uint32_t)t->rows_per_block : 1u, &p.rpb_magic, &p.rpb_shift);
	p.all_materials_ok = t->scene_set && t->all_materials_ok ? 1 : 0;
	p.unit_materials = t->scene_set && t->unit_materials ? 1 : 0;
	p.material_flags = t->scene_set ? t->material_flags : 0;
	p.f_sky_w = (float)t->sky_w;
	p.f_sky_h = (float)t->sky_h;
	p.sun_focus_int = dm_pow_small_int(p.sd.sun_focus);
	p.num_models = t->num_models;
	p.use_bvh = t->bvh_active ? 1 : 0;
	p.bvh_blocks = reinterpret_cast<const float *>(t->bvh_blocks.ptr);
	p.rank = t->rank;
	p.world = t->world;
	p.rows_per_block = t->rows_per_block;
	p.owned_rows = t->owned_rows;
	// ---- batches of samples: radiance[pixel][sample] must fit the HBM budget ----------
	const size_t pixels = owned_pixels(t);
	const int ns = options->num_samples;
	if (t->radiance_budget == 0) {
		size_t free_b = 0, total_b = 0;
		SRT_HIP(t, hipMemGetInfo(&free_b, &total_b));
		size_t budget = free_b / 2; // leave half of what is free to the caller
		const size_t cap = (size_t)96 << 30;
		if (budget > cap) budget = cap;
		if (t->radiance_budget_env) budget = t->radiance_budget_env; // SRT_RADIANCE_BUDGET_MB, read once by srt_create
		t->radiance_budget = budget;
	}
	uint32_t batch = ns > 0 ? (uint32_t)ns : 0u;
	if (pixels > 0 && batch > 0) {
		size_t fit = t->radiance_budget / (pixels * 12);
		if (fit < 1) fit = 1;
		// the trace kernel numbers the work-items of a launch with 32 bits
		const size_t fit32 = (size_t)0xfffffff0u / pixels;
		if (fit32 < fit) fit = fit32 ? fit32 : 1;
		if (fit < batch) batch = (uint32_t)fit;
		// Array scan of large meshes: a ray that enters a model's box tests every triangle, so one launch over all samples
		// can run for a long time (BASELINE configs[4]: 5.8 s; a denser mesh or more samples: minutes) -- longer than a
		// compute queue should be held. Sample batches bound a launch by SRT_SCAN_PAIRS_PER_LAUNCH ray-triangle pairs counted
		// as if EVERY path entered every box once (~1 s at the measured 9e12 such pairs per second; a launch whose paths all
		// bounce ten times inside a box takes ten times that). A batch ends in a long tail -- the last rays scan a few lanes
		// at a time -- which is why batches overlap (below) and are not made smaller than this: configs[4] at full size,
		// batches of 8 / 16 / 32 samples one after the other 15.0 / - / 7.9 s, overlapped 8.5 / 6.8 / 6.4 s.
		if (t->scan_tris > 4096) {
			const double per_sample = (double)pixels * (double)t->scan_tris;
			double pairs = SRT_SCAN_PAIRS_PER_LAUNCH;
			if (const char *env = dev_env("SRT_SCAN_PAIRS"))
				if (atof(env) > 0.0) pairs = atof(env);
			const double cap = pairs / per_sample;
			const uint32_t cap_u = cap < 1.0 ? 1u : (cap > 1e9 ? 0xffffffffu : (uint32_t)cap);
			if (cap_u < batch) batch = cap_u;
		}
		if (const char *env = dev_env("SRT_FORCE_BATCH")) // (development: sample batches smaller than the memory asks for)
			if (atoi(env) > 0 && (uint32_t)atoi(env) < batch) batch = (uint32_t)atoi(env);
		// several batches alternate between TWO radiance buffers (below): both must fit the budget
		if (SRT_OVERLAP_BATCHES && batch < (uint32_t)ns && (size_t)batch * 2 > fit) batch = (uint32_t)(fit / 2 ? fit / 2 : 1);
		if (batch > 4 && (batch & 3u)) batch &= ~3u; // keep the reduce kernel's 16-byte loads aligned
	}
	// allocate; if the device cannot give that much right now, fall back to smaller batches
	while (batch) {
		const size_t buffers = SRT_OVERLAP_BATCHES && batch < (uint32_t)ns ? 2 : 1;
		hipError_t e = t->radiance.reserve(buffers * (((pixels * (size_t)batch * 3 + 4) + 3) & ~(size_t)3)); // each buffer a whole number of 16-byte units
		if (e == hipSuccess) break;
		(void)hipGetLastError(); // clear the sticky out-of-memory state
		if (e != hipErrorOutOfMemory || batch == 1)
			return fail(t, SRT_ERR_HIP, std::string("srt_trace: radiance buffer: ") + hipGetErrorString(e));
		batch = batch > 8 ? ((batch / 2) & ~3u) : batch / 2;
		t->radiance_budget = pixels * 12 * (size_t)batch;
	}
	const uint32_t n_batches = batch ? ((uint32_t)ns + batch - 1) / batch : 0u;
	if (n_batches > 1) SRT_HIP(t, t->running.reserve(pixels * 4));
	p.radiance = t->radiance.ptr;
	p.queue = t->counters.ptr + SRT_CTR_QUEUE;
	int per_cu = srt_trace_resident_waves_per_cu(p, t->count_tris);
	if (const char *env = dev_env("SRT_WAVES_PER_CU")) {
		const int v = atoi(env);
		if (v > 0 && v < per_cu) per_cu = v;
	}
	t->last_waves_per_cu = per_cu;
	int slots = t->num_cus * per_cu;
	if (slots > SRT_WAVE_CTR_SLOTS) slots = SRT_WAVE_CTR_SLOTS; // one counter line per persistent wave

	// Array scan: one block of scan / park stacks per persistent wave the largest launch of this dispatch starts (46 KB each),
	// the launch-end ray pool's records (84 MB), and a second set of both only when sample batches overlap: 320 MB for a
	// one-launch frame on 256 CUs, 640 MB for an overlapped one. If the device cannot give that, the
	// dispatch runs without the pool (every wave scans its own remainder: slower tails, same canvas) before it fails.
	const bool wants_scan_queue = t->num_models > 0 && (!t->bvh_active || srt_bvh_suspends()) && srt_scan_queue_in_hbm();
	const bool overlap_batches = SRT_OVERLAP_BATCHES && n_batches > 1;
	size_t scan_waves = (size_t)slots;
	{
		const unsigned long long most_items = (unsigned long long)pixels * batch;
		const unsigned long long need = (most_items + 63ull) / 64ull;
		if (need < (unsigned long long)scan_waves) scan_waves = (size_t)(need ? need : 1ull);
	}
	uint32_t pool_blocks = wants_scan_queue && !dev_env("SRT_NO_SCAN_POOL") ? (uint32_t)SRT_POOL_BLOCKS : 0u;
	if (const char *env = dev_env("SRT_POOL_BLOCKS")) { // tests: a pool that overflows
		const int v = atoi(env);
		if (v >= 0 && (uint32_t)v < pool_blocks) pool_blocks = (uint32_t)v;
	}
	if (wants_scan_queue) {
		const size_t sets = overlap_batches ? 2 : 1;
		hipError_t e = t->scan_queue.reserve(sets * SRT_SCAN_SET_FLOATS(scan_waves, pool_blocks != 0u));
		if (e == hipErrorOutOfMemory && pool_blocks != 0u) {
			(void)hipGetLastError();
			pool_blocks = 0u;
			e = t->scan_queue.reserve(sets * SRT_SCAN_SET_FLOATS(scan_waves, false));
		}
		if (e != hipSuccess) {
			(void)hipGetLastError();
			return fail(t, SRT_ERR_HIP, std::string("srt_trace: scan stacks: ") + hipGetErrorString(e));
		}
	}

	ReduceParams rp;
	rp.radiance = t->radiance.ptr;
	rp.running = t->running.ptr;
	rp.canvas = t->canvas;
	rp.counters = t->counters.ptr;
	// The kernel timers (srt_last_kernel_ms, srt_last_trace_kernel_ms) are four event records per dispatch: 10-17 us of a 150 us
	// interactive frame. srt_trace always takes them; the render calls only when asked to (srt_set_kernel_timers).
	const bool timed = !fused_argb || t->timers_in_render;
	rp.num_pixels = (uint32_t)pixels;
	rp.num_samples = ns;
	rp.queue_reset = nullptr;
	rp.argb = nullptr;
	rp.num_steps = ticks_stopped;

	while (t->ev_k.size() < 2 * (size_t)n_batches) { // std::vector growth is the only throwing step: srt_trace's callers catch nothing
		hipEvent_t ev = nullptr;
		SRT_HIP(t, hipEventCreate(&ev));
		try {
			t->ev_k.push_back(ev);
		} catch (...) {
			(void)hipEventDestroy(ev);
			return fail(t, SRT_ERR_INVALID, "out of host memory");
		}
	}
	t->ev_k_used = 0;
	if (timed) SRT_HIP(t, hipEventRecord(t->ev_t0, t->stream));
	if (n_batches == 0) {
		// num_samples <= 0: no paths; the reduction still applies colour = 0 / num_samples (render.cl:520-522)
		rp.batch_samples = 0;
		rp.first_batch = rp.last_batch = 1;
		rp.argb = fused_argb;
		srt_launch_reduce(rp, t->stream);
	}
	// Several sample batches: even and odd batches trace on two streams of their own, each into its own radiance buffer,
	// work cursor and set of per-wave counter lines, so that the tail of a batch (its last long paths, a few lanes per wave
	// and most waves gone: half of a launch of the 10^5-triangle array scan) runs under the next batch instead of leaving
	// the GPU idle. The ordered reductions stay on the caller's stream, in batch order; a batch's trace waits for the
	// reduction that last read its buffer.
	const bool overlap = SRT_OVERLAP_BATCHES && n_batches > 1;
	t->batches_overlapped = overlap;
	const size_t radiance_stride = ((pixels * (size_t)batch * 3 + 4) + 3) & ~(size_t)3; // floats per buffer: a multiple of 4, so that the second buffer's float4 stores stay 16-byte aligned for any pixel count and batch size
	if (overlap) {
		for (int k = 0; k < 2; k++) {
			if (!t->batch_stream[k]) {
				// The two streams must not share a hardware queue, or the batches they carry run one after the other: the runtime
				// deals its few queues out to streams as they are created, and in a process that holds other streams (bench.py:
				// torch's, the headline handle's) both of these landed on one -- configs[4] 5.5 s instead of 4.3 s. Streams of
				// different priority never share a queue, so the odd batches' stream is created one level above the even ones'.
				int pr_low = 0, pr_high = 0;
				(void)hipDeviceGetStreamPriorityRange(&pr_low, &pr_high); // (numerically lower = higher priority)
				const int pr = (k == 1 && pr_high < pr_low) ? pr_low - 1 : pr_low;
				SRT_HIP(t, hipStreamCreateWithPriority(&t->batch_stream[k], hipStreamNonBlocking, pr));
			}
			if (!t->ev_batch_traced[k]) SRT_HIP(t, hipEventCreateWithFlags(&t->ev_batch_traced[k], hipEventDisableTiming));
			if (!t->ev_batch_reduced[k]) SRT_HIP(t, hipEventCreateWithFlags(&t->ev_batch_reduced[k], hipEventDisableTiming));
		}
		if (!t->ev_batch_fork) SRT_HIP(t, hipEventCreateWithFlags(&t->ev_batch_fork, hipEventDisableTiming));
		SRT_HIP(t, hipEventRecord(t->ev_batch_fork, t->stream)); // everything the caller's stream holds so far (scene upload, clear, ...)
		for (int k = 0; k < 2; k++) SRT_HIP(t, hipStreamWaitEvent(t->batch_stream[k], t->ev_batch_fork, 0));
	}
	for (uint32_t b = 0; b < n_batches; b++) {
		const int par = overlap ? (int)(b & 1u) : 0;
		hipStream_t ts = overlap ? t->batch_stream[par] : t->stream;
		p.radiance = t->radiance.ptr + (size_t)par * radiance_stride;
		rp.radiance = p.radiance;
		p.queue = t->counters.ptr + (par ? SRT_CTR_QUEUE2 : SRT_CTR_QUEUE);
		p.wave_counters = t->wave_counters.ptr + (size_t)par * SRT_WAVE_CTR_SLOTS * SRT_WAVE_CTR_STRIDE;
		p.scan_queue = wants_scan_queue ? t->scan_queue.ptr + (size_t)par * SRT_SCAN_SET_FLOATS(scan_waves, pool_blocks != 0u) : nullptr;
		p.pool_blocks = pool_blocks;
		if (overlap && b >= 2) SRT_HIP(t, hipStreamWaitEvent(ts, t->ev_batch_reduced[par], 0)); // batch b - 2 has been summed up
		const uint32_t s0 = b * batch;
		const uint32_t nbs = (uint32_t)ns - s0 < batch ? (uint32_t)ns - s0 : batch;
		p.batch_samples = nbs;
		p.nbs_magic16 = nbs ? (65536u + nbs - 1u) / nbs : 0u;
		p.first_sample = s0;
		p.total_items = (unsigned long long)pixels * nbs;
		// chunks per atomic: ~8 per resident wave for balance, whole sub-jobs (so that every sub-job starts
		// 16-byte aligned in the radiance buffer), at most 5 of them. A dispatch too small for that
		// (an interactive 960x540 frame at 2 spp is 200 items per wave) gets ONE chunk per wave instead:
		// measured 0.39 -> 0.35 ms against two rounds of single sub-jobs.
		const unsigned long long sub = (unsigned long long)srt_sub_job_items(t->num_models > 0, t->bvh_active);
		// A small dispatch ends in the tail of its longest paths, during which every resident wave still
		// issues whole iterations for a few live lanes: round 2 measured fewer, faster waves as the winner there
		// (960x540x2spp: 2 / 3 / 4 / 5 waves per SIMD = 0.28 / 0.27 / 0.30 / 0.31 ms, hence "at least ~320 items per wave");
		// with round 4's loop (full SHADE phases, cheap bookkeeping) every resident wave pays again down to ~190 items
		// each. Never fewer than 2 waves per SIMD.
		unsigned long long items_per_wave = 192ull; // (round 4's kernel, 960x540x2 spp: 128 / 192 / 256 / 320 / 448 items per wave = 0.104 / 0.104 / 0.104 / 0.117 / 0.143 ms)
		if (const char *env = dev_env("SRT_ITEMS_PER_WAVE"))
			if (atoi(env) > 0) items_per_wave = (unsigned long long)atoi(env);
		unsigned long long slots_b = p.total_items / items_per_wave;
		if (slots_b < (unsigned long long)t->num_cus * 8ull) slots_b = (unsigned long long)t->num_cus * 8ull;
		if (slots_b > (unsigned long long)slots) slots_b = (unsigned long long)slots;
		unsigned long long job = (p.total_items / (slots_b * 8ull) / sub) * sub;
		if (job < sub) job = ((p.total_items + slots_b - 1ull) / slots_b + sub - 1ull) / sub * sub;
		if (job < sub) job = sub;
		// Scenes with models: what a chunk costs varies wildly with where it lies (pixels on a glass mesh: ten walks or scans
		// per path; sky pixels: none), and the launch ends when the wave with the last expensive chunk does. Small chunks
		// shorten that tail, but a wave that hops between distant pixels loses the coherence of neighbouring rays (BVH blocks,
		// scans shared by a wave-full): about 2.5 pixels' worth of samples per chunk, between 2 and 8 sub-jobs (BVH: 16). Measured, chunks
		// of 1 / 2 / 5 sub-jobs: BVH walk of the 10^5-triangle mesh at 16 spp 5.1 / 5.0 / 7.2 ms, at 256 spp 63.9 / 46.8 / 39.0;
		// array scan of the two 968-triangle meshes at 32 spp 20.2 / 16.0 / 18.1. At full size (round 3, profiles/README.md),
		// chunks of 4 / 5 / 6 / 8 / 12 / 16 sub-jobs: configs[2] array scan (512 spp, sub-jobs of 64) 134.7 / 128.0 / 124.1 / 121.7 /
		// 125.1 / 130.6 ms, configs[2] BVH (sub-jobs of 128) 43.1 / 40.6 / 38.8 / 37.1 / 37.5 / 38.9 ms; configs[4] BVH (256 spp:
		// 2.5 pixels = 5 sub-jobs) 37.4 / 36.3 / 35.9 / 37.8 / 41.2 / 46.7 ms; configs[4] array scan (36 samples per launch: 2) 1 / 2 / 3
		// sub-jobs 4,524 / 4,508 / 4,593 ms.
		unsigned long long cap_subs = 5ull;
		if (t->num_models > 0) {
			cap_subs = (5ull * nbs / 2ull + sub - 1ull) / sub;
			// (BVH, sub-jobs of 64: chunks of 10 / 16 sub-jobs configs[2] 39.3 / 35.7 ms, configs[4] 35.7-36.5 / 36.5 ms; sub-jobs of 128
			// and 8: 37.0 / 36.3. Round 4's walks, chunks of 6 / 8 / 10 / 12 / 16 / 24 / 32 sub-jobs -- float boxes: configs[2] 47.1 / 40.7 /
			// 37.5 / 35.8 / 33.7 / 33.5 / 36.0 ms, configs[4] 36.0 / 33.6 / 32.4 / 32.2 / 33.4 / 36.9 / 41.0; boxes as bytes: 46.7 / 40.3 / 37.0 /
			// 35.3 / 33.2 / 33.7 / 35.4 and 33.9 / 31.6 / 30.6 / 31.2 / 31.7 / 36.4 / 40.2: two and a half pixels' worth, at most 16)
			const unsigned long long most = t->bvh_active ? 16ull : 8ull;
			cap_subs = cap_subs < 2ull ? 2ull : (cap_subs > most ? most : cap_subs);
		}
		if (const char *env = dev_env("SRT_JOB_CAP_SUBS"))
			if (atoi(env) > 0) cap_subs = (unsigned long long)atoi(env);
		const unsigned long long job_cap = cap_subs * sub;
		if (job > job_cap) job = job_cap;
		p.job_items = (uint32_t)job;
		const unsigned long long waves_needed = (p.total_items + 63ull) / 64ull;
		const int num_waves = (int)(waves_needed < slots_b ? waves_needed : slots_b);
		// the work cursor is zero: the reduction behind the launch that used it last has reset it (srt_reduce_kernel). Only a launch
		// whose reduction was never enqueued (an error in between) leaves it dirty.
		if (t->queue_dirty[par]) SRT_HIP(t, hipMemsetAsync(p.queue, 0, sizeof(unsigned long long), ts));
		t->queue_dirty[par] = true;
		if (p.pool_blocks) SRT_HIP(t, hipMemsetAsync(p.scan_queue, 0, (size_t)SRT_POOL_CTL_WORDS * sizeof(uint32_t), ts));
		if (timed) SRT_HIP(t, hipEventRecord(t->ev_k[2 * b], ts));
		t->last_grid = num_waves;
		srt_launch_trace(p, t->count_tris, num_waves, ts);
		SRT_HIP(t, hipGetLastError());
		if (timed) SRT_HIP(t, hipEventRecord(t->ev_k[2 * b + 1], ts));
		if (overlap) {
			SRT_HIP(t, hipEventRecord(t->ev_batch_traced[par], ts));
			SRT_HIP(t, hipStreamWaitEvent(t->stream, t->ev_batch_traced[par], 0));
		}
		t->ev_k_used = 2 * (size_t)(b + 1);
		rp.batch_samples = nbs;
		rp.first_batch = (b == 0);
		rp.last_batch = (b == n_batches - 1);
		rp.queue_reset = p.queue;
		rp.argb = rp.last_batch ? fused_argb : nullptr;
		srt_launch_reduce(rp, t->stream);
		SRT_HIP(t, hipGetLastError());
		t->queue_dirty[par] = false; // (no pixels: neither launch ran, the cursor is untouched)
		if (overlap) SRT_HIP(t, hipEventRecord(t->ev_batch_reduced[par], t->stream));
	}
	if (timed) SRT_HIP(t, hipEventRecord(t->ev_t1, t->stream));
	t->have_trace_ev = timed;
	t->have_kernel_ev = timed && n_batches > 0;
	return SRT_OK;
}

int srt_resolve(srt_tracer *t, uint32_t ticks_stopped) {
	if (!t) return SRT_ERR_INVALID;
	SRT_HIP(t, hipSetDevice(t->device));
	ResolveParams rp;
	rp.canvas = t->canvas;
	rp.argb = t->argb.ptr;
	rp.num_steps = ticks_stopped;
	rp.num_pixels = (uint32_t)owned_pixels(t);
	SRT_HIP(t, hipEventRecord(t->ev_r0, t->stream));
	srt_launch_resolve(rp, t->stream);
	SRT_HIP(t, hipGetLastError());
	SRT_HIP(t, hipEventRecord(t->ev_r1, t->stream));
	t->have_resolve_ev = true;
	return SRT_OK;
}

int srt_resolve_external(srt_tracer *t, const void *device_canvas, uint32_t num_pixels, uint32_t ticks_stopped,
                         void *device_argb) {
	if (!t) return SRT_ERR_INVALID;
	if (!device_canvas || !device_argb) return fail(t, SRT_ERR_INVALID, "srt_resolve_external: NULL buffer");
	SRT_HIP(t, hipSetDevice(t->device));
	ResolveParams rp;
	rp.canvas = static_cast<const float *>(device_canvas);
	rp.argb = static_cast<uint8_t *>(device_argb);
	rp.num_steps = ticks_stopped;
	rp.num_pixels = num_pixels;
	SRT_HIP(t, hipEventRecord(t->ev_r0, t->stream));
	srt_launch_resolve(rp, t->stream);
	SRT_HIP(t, hipGetLastError());
	SRT_HIP(t, hipEventRecord(t->ev_r1, t->stream));
	t->have_resolve_ev = true;
	return SRT_OK;
}

int srt_synchronize(srt_tracer *t) {
	if (!t) return SRT_ERR_INVALID;
	SRT_HIP(t, hipSetDevice(t->device));
	SRT_HIP(t, hipStreamSynchronize(t->stream));
	return SRT_OK;
}

int srt_render(srt_tracer *t, const srt_render_data *options, uint32_t ticks_stopped, uint8_t *argb_out) {
	if (!t) return SRT_ERR_INVALID;
	if (!argb_out) return fail(t, SRT_ERR_INVALID, "srt_render: argb_out is NULL");
	int rc = srt_trace_fused(t, options, t->argb.ptr, ticks_stopped); // trace, ordered reduction and resolve: the last two in one launch
	if (rc) return rc;
	t->have_resolve_ev = false;
	// blocking read-back, as queue.enqueue_read_buffer (src/tracer.cpp:115)
	SRT_HIP(t, hipMemcpyAsync(argb_out, t->argb.ptr, owned_pixels(t) * 4, hipMemcpyDeviceToHost, t->stream));
	SRT_HIP(t, hipStreamSynchronize(t->stream));
	return SRT_OK;
}

int srt_render_async(srt_tracer *t, const srt_render_data *options, uint32_t ticks_stopped, uint8_t *argb_out) {
	if (!t) return SRT_ERR_INVALID;
	if (!argb_out) return fail(t, SRT_ERR_INVALID, "srt_render_async: argb_out is NULL");
	int rc = srt_trace_fused(t, options, t->argb.ptr, ticks_stopped);
	if (rc) return rc;
	t->have_resolve_ev = false;
	SRT_HIP(t, hipMemcpyAsync(argb_out, t->argb.ptr, owned_pixels(t) * 4, hipMemcpyDeviceToHost, t->stream));
	return SRT_OK; // argb_out is valid after srt_synchronize()
}

int srt_read_canvas(srt_tracer *t, float *rgba_out) {
	if (!t) return SRT_ERR_INVALID;
	if (!rgba_out) return fail(t, SRT_ERR_INVALID, "srt_read_canvas: NULL");
	SRT_HIP(t, hipSetDevice(t->device));
	SRT_HIP(t, hipMemcpyAsync(rgba_out, t->canvas, owned_pixels(t) * 16, hipMemcpyDeviceToHost, t->stream));
	SRT_HIP(t, hipStreamSynchronize(t->stream));
	return SRT_OK;
}

int srt_read_argb(srt_tracer *t, uint8_t *argb_out) {
	if (!t) return SRT_ERR_INVALID;
	if (!argb_out) return fail(t, SRT_ERR_INVALID, "srt_read_argb: NULL");
	SRT_HIP(t, hipSetDevice(t->device));
	SRT_HIP(t, hipMemcpyAsync(argb_out, t->argb.ptr, owned_pixels(t) * 4, hipMemcpyDeviceToHost, t->stream));
	SRT_HIP(t, hipStreamSynchronize(t->stream));
	return SRT_OK;
}

int srt_get_counters(srt_tracer *t, srt_counters *out) {
	if (!t) return SRT_ERR_INVALID;
	if (!out) return fail(t, SRT_ERR_INVALID, "srt_get_counters: NULL");
	SRT_HIP(t, hipSetDevice(t->device));
	unsigned long long h[SRT_CTR_COUNT];
	std::vector<unsigned long long> w;
	try {
		w.resize((size_t)2 * SRT_WAVE_CTR_SLOTS * SRT_WAVE_CTR_STRIDE);
	} catch (...) {
		return fail(t, SRT_ERR_INVALID, "out of host memory");
	}
	SRT_HIP(t, hipMemcpyAsync(h, t->counters.ptr, sizeof h, hipMemcpyDeviceToHost, t->stream));
	SRT_HIP(t, hipMemcpyAsync(w.data(), t->wave_counters.ptr, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, t->stream));
	SRT_HIP(t, hipStreamSynchronize(t->stream));
	unsigned long long sum[5] = {0, 0, 0, 0, 0};
	for (size_t i = 0; i < (size_t)2 * SRT_WAVE_CTR_SLOTS; i++)
		for (int k = 0; k < 5; k++) sum[k] += w[i * SRT_WAVE_CTR_STRIDE + k];
	out->rays = sum[0];
	out->sky = sum[1];
	out->paths = sum[2];
	out->tri_tests = sum[3];
	out->tri_pass_u = sum[4];
	out->nan_pixels = h[SRT_CTR_NAN];
	out->watchdog = h[SRT_CTR_WATCHDOG];
	return SRT_OK;
}

int srt_reset_counters(srt_tracer *t) {
	if (!t) return SRT_ERR_INVALID;
	SRT_HIP(t, hipSetDevice(t->device));
	SRT_HIP(t, hipMemsetAsync(t->counters.ptr, 0, SRT_CTR_COUNT * sizeof(unsigned long long), t->stream));
	SRT_HIP(t, hipMemsetAsync(t->wave_counters.ptr, 0, (size_t)2 * SRT_WAVE_CTR_SLOTS * SRT_WAVE_CTR_STRIDE * sizeof(unsigned long long), t->stream));
	return SRT_OK;
}

int srt_last_kernel_ms(srt_tracer *t, float *trace_ms, float *resolve_ms) {
	if (!t) return SRT_ERR_INVALID;
	SRT_HIP(t, hipSetDevice(t->device));
	SRT_HIP(t, hipStreamSynchronize(t->stream));
	if (trace_ms) {
		*trace_ms = 0.f;
		if (t->have_trace_ev) SRT_HIP(t, hipEventElapsedTime(trace_ms, t->ev_t0, t->ev_t1));
	}
	if (resolve_ms) {
		*resolve_ms = 0.f;
		if (t->have_resolve_ev) SRT_HIP(t, hipEventElapsedTime(resolve_ms, t->ev_r0, t->ev_r1));
	}
	return SRT_OK;
}

int srt_set_kernel_timers(srt_tracer *t, int enable) {
	if (!t) return SRT_ERR_INVALID;
	t->timers_in_render = enable != 0;
	return SRT_OK;
}

int srt_set_radiance_budget(srt_tracer *t, size_t bytes) {
	if (!t) return SRT_ERR_INVALID;
	t->radiance_budget = bytes; // 0 = choose from free HBM at the next srt_trace
	return SRT_OK;
}

int srt_last_trace_kernel_ms(srt_tracer *t, float *kernel_ms) {
	if (!t) return SRT_ERR_INVALID;
	if (!kernel_ms) return fail(t, SRT_ERR_INVALID, "srt_last_trace_kernel_ms: NULL");
	SRT_HIP(t, hipSetDevice(t->device));
	SRT_HIP(t, hipStreamSynchronize(t->stream));
	*kernel_ms = 0.f;
	if (t->have_kernel_ev && t->batches_overlapped && t->ev_k_used >= 4) {
		// overlapping launches: from the first batch's start to the later of the last two ends (the reductions of the earlier
		// batches run inside that span)
		for (size_t i = t->ev_k_used - 4; i + 1 < t->ev_k_used; i += 2) {
			float ms = 0.f;
			SRT_HIP(t, hipEventElapsedTime(&ms, t->ev_k[0], t->ev_k[i + 1]));
			if (ms > *kernel_ms) *kernel_ms = ms;
		}
	} else if (t->have_kernel_ev) {
		for (size_t i = 0; i + 1 < t->ev_k_used; i += 2) { // one pair per sample batch: the reductions between them are not counted
			float ms = 0.f;
			SRT_HIP(t, hipEventElapsedTime(&ms, t->ev_k[i], t->ev_k[i + 1]));
			*kernel_ms += ms;
		}
	}
	return SRT_OK;
}

int srt_last_trace_launches(const srt_tracer *t, int *launches, int *overlapped) {
	if (!t) return SRT_ERR_INVALID;
	if (launches) *launches = (int)(t->ev_k_used / 2);
	if (overlapped) *overlapped = t->batches_overlapped ? 1 : 0;
	return SRT_OK;
}

int srt_device_buffers(srt_tracer *t, void **canvas, size_t *canvas_bytes, void **argb, size_t *argb_bytes) {
	if (!t) return SRT_ERR_INVALID;
	if (canvas) *canvas = t->canvas;
	if (canvas_bytes) *canvas_bytes = t->canvas_bytes;
	if (argb) *argb = t->argb.ptr;
	if (argb_bytes) *argb_bytes = t->argb.cap;
	return SRT_OK;
}

int srt_bind_canvas(srt_tracer *t, void *device_canvas, size_t bytes) {
	if (!t) return SRT_ERR_INVALID;
	SRT_HIP(t, hipSetDevice(t->device));
	SRT_HIP(t, hipStreamSynchronize(t->stream));
	if (!device_canvas) {
		t->canvas = t->canvas_own.ptr;
		t->canvas_bytes = t->canvas_own.cap * sizeof(float);
		return SRT_OK;
	}
	if (bytes < owned_pixels(t) * 16) return fail(t, SRT_ERR_INVALID, "srt_bind_canvas: buffer smaller than owned_rows*width*16");
	t->canvas = static_cast<float *>(device_canvas);
	t->canvas_bytes = bytes;
	return SRT_OK;
}

int srt_bind_stream(srt_tracer *t, void *hip_stream) {
	if (!t) return SRT_ERR_INVALID;
	SRT_HIP(t, hipSetDevice(t->device));
	SRT_HIP(t, hipStreamSynchronize(t->stream));
	t->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : t->own_stream;
	return SRT_OK;
}

int srt_set_partition(srt_tracer *t, int rank, int world, int rows_per_block) {
	if (!t) return SRT_ERR_INVALID;
	if (world < 1 || rank < 0 || rank >= world || rows_per_block < 1)
		return fail(t, SRT_ERR_INVALID, "srt_set_partition: need 0 <= rank < world and rows_per_block >= 1");
	SRT_HIP(t, hipSetDevice(t->device));
	SRT_HIP(t, hipStreamSynchronize(t->stream));
	const int padded = srt_partition_padded_rows(t->height, world, rows_per_block);
	const size_t need = (size_t)padded * t->width * 4; // floats; padded >= owned
	if (t->canvas == t->canvas_own.ptr) {
		SRT_HIP(t, t->canvas_own.reserve(need));
		t->canvas = t->canvas_own.ptr;
		t->canvas_bytes = t->canvas_own.cap * sizeof(float);
	} else if (t->canvas_bytes < need * sizeof(float)) {
		return fail(t, SRT_ERR_INVALID, "srt_set_partition: bound canvas too small for padded_rows*width*16");
	}
	SRT_HIP(t, t->argb.reserve((size_t)padded * t->width * 4));
	t->rank = rank;
	t->world = world;
	t->rows_per_block = rows_per_block;
	t->owned_rows = srt_partition_owned_rows(t->height, rank, world, rows_per_block);
	SRT_HIP(t, hipMemsetAsync(t->argb.ptr, 0, t->argb.cap, t->stream));
	return clear_canvas_impl(t);
}

int srt_selftest_math(srt_tracer *t, uint32_t stride, uint64_t out[16]) {
	if (!t) return SRT_ERR_INVALID;
	if (!out || stride == 0) return fail(t, SRT_ERR_INVALID, "srt_selftest_math: bad arguments");
	SRT_HIP(t, hipSetDevice(t->device));
	unsigned long long *d = nullptr;
	SRT_HIP(t, hipMalloc(reinterpret_cast<void **>(&d), 16 * sizeof(unsigned long long)));
	hipError_t e = hipMemsetAsync(d, 0, 16 * sizeof(unsigned long long), t->stream);
	if (e == hipSuccess) {
		srt_launch_selftest(d, stride, t->stream);
		e = hipGetLastError();
	}
	unsigned long long h[16] = {0};
	if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, t->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(t, SRT_ERR_HIP, std::string("srt_selftest_math: ") + hipGetErrorString(e));
	for (int i = 0; i < 16; i++) out[i] = h[i];
	return SRT_OK;
}

int srt_set_acceleration(srt_tracer *t, int mode) {
	if (!t) return SRT_ERR_INVALID;
	if (mode != SRT_ACCEL_NONE && mode != SRT_ACCEL_BVH) return fail(t, SRT_ERR_INVALID, "srt_set_acceleration: unknown mode");
	t->accel_mode = mode;
	return SRT_OK;
}

int srt_bvh_build_host(const srt_shape *model, const srt_triangle *triangles, size_t n_triangles, srt_bvh_node *nodes_out, size_t nodes_cap,
                       uint32_t *order_out, size_t order_cap, size_t *n_nodes) {
	if (!model || !n_nodes || model->type != SRT_SHAPE_MODEL || (n_triangles && !triangles)) return SRT_ERR_INVALID;
	const srt_model &m = model->shape.model;
	if ((uint64_t)m.triangle_index + m.num_triangles > n_triangles || m.num_triangles > 0x0fffffffu) return SRT_ERR_INVALID;
	try {
		std::vector<BvhNode> nodes;
		std::vector<uint32_t> order;
		if (m.num_triangles > 0) {
			BvhBuilder bb(nodes, order);
			bb.run(m, triangles, 0u);
		}
		*n_nodes = nodes.size();
		if (nodes_out) memcpy(nodes_out, nodes.data(), std::min(nodes.size(), nodes_cap) * sizeof(BvhNode));
		if (order_out) memcpy(order_out, order.data(), std::min(order.size(), order_cap) * sizeof(uint32_t));
	} catch (...) { // std::bad_alloc: no C++ exception may cross the C ABI
		return SRT_ERR_INVALID;
	}
	return SRT_OK;
}

int srt_bvh_wide_host(const srt_shape *model, const srt_triangle *triangles, size_t n_triangles, int force_balanced, uint32_t *blocks_out,
                      size_t blocks_cap, uint32_t *dest_out, size_t dest_cap, size_t *n_blocks, uint32_t *root, uint32_t *stack_need, int *balanced) {
	if (!model || !n_blocks || model->type != SRT_SHAPE_MODEL || (n_triangles && !triangles)) return SRT_ERR_INVALID;
	const srt_model &m = model->shape.model;
	if ((uint64_t)m.triangle_index + m.num_triangles > n_triangles || m.num_triangles > 0x0fffffffu) return SRT_ERR_INVALID;
	try {
		BvhCacheEntry ent;
		ent.balanced = force_balanced != 0;
		if (m.num_triangles > 0) ent.build(m, triangles);
		*n_blocks = ent.wide.blocks.size() / 32;
		if (root) *root = ent.wide.root;
		if (stack_need) *stack_need = ent.wide.need;
		if (balanced) *balanced = ent.balanced ? 1 : 0;
		if (blocks_out) memcpy(blocks_out, ent.wide.blocks.data(), std::min(ent.wide.blocks.size(), blocks_cap * 32) * sizeof(uint32_t));
		if (dest_out) memcpy(dest_out, ent.wide.dest.data(), std::min(ent.wide.dest.size(), dest_cap) * sizeof(uint32_t));
	} catch (...) { // std::bad_alloc: no C++ exception may cross the C ABI
		return SRT_ERR_INVALID;
	}
	return SRT_OK;
}

int srt_acceleration_info(const srt_tracer *t, uint64_t out[7]) {
	if (!t || !out) return SRT_ERR_INVALID;
	for (int i = 0; i < 7; i++) out[i] = t->bvh_active ? t->bvh_info[i] : 0;
	return SRT_OK;
}

/* diagnostics: sums of the eight per-wave counter slots since the last reset (slot 5 = unused since round 4, 6 = wave
 * iterations, 7 = SHADE phases) */
int srt_debug_counters(srt_tracer *t, uint64_t out[18]) {
	if (!t || !out) return SRT_ERR_INVALID;
	SRT_HIP(t, hipSetDevice(t->device));
	std::vector<unsigned long long> w;
	try {
		w.resize((size_t)2 * SRT_WAVE_CTR_SLOTS * SRT_WAVE_CTR_STRIDE);
	} catch (...) {
		return fail(t, SRT_ERR_INVALID, "out of host memory");
	}
	SRT_HIP(t, hipMemcpyAsync(w.data(), t->wave_counters.ptr, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, t->stream));
	SRT_HIP(t, hipStreamSynchronize(t->stream));
	for (int k = 0; k < 18; k++) out[k] = 0;
	for (size_t i = 0; i < (size_t)2 * SRT_WAVE_CTR_SLOTS; i++) {
		for (int k = 0; k < 8; k++) out[k] += w[i * SRT_WAVE_CTR_STRIDE + k];
		for (int k = 8; k < 16; k++) out[k + 2] += w[i * SRT_WAVE_CTR_STRIDE + k]; // phase clocks of -DSRT_PHASE_CLOCK builds
	}
	out[8] = (uint64_t)t->last_waves_per_cu;
	out[9] = (uint64_t)t->last_grid;
	return SRT_OK;
}

/* -DSRT_REGION_COUNT builds: per region of the trace kernel (kernels.hip SRT_REGION_LIST, in that order) how often a wave
 * ran it and with how many lanes, summed over the waves since the last reset. *written = 0 in the product build. */
int srt_debug_region_counters(srt_tracer *t, uint64_t *out, int capacity, int *written) {
	if (!t || !out || !written || capacity < 0) return SRT_ERR_INVALID;
	*written = 0;
#ifdef SRT_REGION_COUNT
	SRT_HIP(t, hipSetDevice(t->device));
	std::vector<unsigned long long> w;
	try {
		w.resize((size_t)2 * SRT_WAVE_CTR_SLOTS * SRT_WAVE_CTR_STRIDE);
	} catch (...) {
		return fail(t, SRT_ERR_INVALID, "out of host memory");
	}
	SRT_HIP(t, hipMemcpyAsync(w.data(), t->wave_counters.ptr, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, t->stream));
	SRT_HIP(t, hipStreamSynchronize(t->stream));
	const int n = capacity < 2 * SRT_REGION_MAX ? capacity : 2 * SRT_REGION_MAX;
	for (int k = 0; k < n; k++) out[k] = 0;
	for (size_t i = 0; i < (size_t)2 * SRT_WAVE_CTR_SLOTS; i++)
		for (int k = 0; k < n; k++) out[k] += w[i * SRT_WAVE_CTR_STRIDE + 16 + k];
	*written = n;
#endif
	return SRT_OK;
}

/* test hook: build with triangle counters (instrumented kernel variant) */
int srt_set_count_triangles(srt_tracer *t, int enable) {
	if (!t) return SRT_ERR_INVALID;
	t->count_tris = enable != 0;
	return SRT_OK;
}

} // extern "C"
