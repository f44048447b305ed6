"""CPU: the host-side BVH builder behind srt_set_acceleration, through the host-only entry
points srt_bvh_build_host (the binary hierarchy that is built, cached and refitted) and
srt_bvh_wide_host (its four-wide folding, which the kernel walks); no GPU involved. Structure
invariants, box containment, the bound on a walk's stack, and a float32 re-enactment of the walk
against a brute-force double-precision Moller-Trumbore: every triangle a ray really hits must be
among the triangles the walk tests."""
import numpy as np
import pytest

from simple_raytracer_amd import records as R, scenes as S, tracer as T

END = T.BVH_END
FLT_MAX = np.finfo(np.float32).max
LEAF_MAX = 3  # SRT_BVH_LEAF_MAX (csrc/device_types.h)


def world_vertices(shape, tris):
    first, n = int(shape["triangle_index"]), int(shape["num_triangles"])
    pos = tris["v"]["pos"][first:first + n].reshape(-1, 3)
    return R.transform_points(np.asarray(shape["transform"], np.float32), pos, 1.0).reshape(n, 3, 3).astype(np.float64)


def check_structure(nodes, order, n_tris):
    n = len(nodes)
    assert sorted(order.tolist()) == list(range(n_tris)), "order must be a permutation of the model's triangles"
    skip, leaf = nodes["skip"].astype(np.int64), nodes["leaf"]
    skip_eff = np.where(skip == END, n, skip)
    assert (skip_eff > np.arange(n)).all(), "skip links must point forward (the walk's termination argument)"
    covered = np.zeros(n_tris, np.int32)
    depth_max = 0
    stack = [(0, n, 1)]  # (node, end of its subtree, depth)
    while stack:
        i, end, depth = stack.pop()
        depth_max = max(depth_max, depth)
        assert skip_eff[i] == end, (i, skip_eff[i], end)
        if leaf[i]:
            cnt, first = int(leaf[i] >> 28), int(leaf[i] & 0x0FFFFFFF)
            assert 1 <= cnt <= LEAF_MAX and first + cnt <= n_tris
            covered[first:first + cnt] += 1
            assert end == i + 1, "a leaf has no subtree"
        else:
            left = i + 1
            right = int(skip_eff[left])
            assert left < right < end, "an inner node has exactly two children, back to back"
            for c in (left, right):  # children boxes inside the parent's
                assert (nodes["lo"][c] >= nodes["lo"][i]).all() and (nodes["hi"][c] <= nodes["hi"][i]).all()
            stack.append((left, right, depth + 1))
            stack.append((right, end, depth + 1))
    assert (covered == 1).all(), "every triangle record belongs to exactly one leaf"
    return depth_max


def check_boxes(nodes, order, wv):
    leaf = nodes["leaf"]
    for i in np.nonzero(leaf)[0]:
        cnt, first = int(leaf[i] >> 28), int(leaf[i] & 0x0FFFFFFF)
        v = wv[order[first:first + cnt]].reshape(-1, 3)
        assert (v >= nodes["lo"][i] - 0).all() and (v <= nodes["hi"][i] + 0).all(), f"leaf {i} does not contain its triangles"
        assert (v.min(axis=0) > nodes["lo"][i]).all() and (v.max(axis=0) < nodes["hi"][i]).all(), "boxes are padded"


def inner_children(blocks, idx):
    """(first, nk, tags[4], lo[4][3], hi[4][3]) of inner block idx (csrc/device_types.h): boxes as bytes on the block's power-of-two
    grid -- origin in dwords 0-2, exponents and count in 3, lo bytes 4-6, hi bytes 7-9, tags 10, first 11; bound = fmaf(byte, 2^e,
    origin) (the product is exact, the float64 sum rounds to float32 once: the kernel's fmaf up to double rounding)."""
    b = blocks[idx]
    assert not b[12:].any()
    origin = b.view(np.float32)[0:3].astype(np.float64)
    ex, nk = int(b[3]), int(b[3]) >> 24
    scale = np.array([((ex >> (8 * a)) & 255) << 23 for a in range(3)], np.uint32).view(np.float32).astype(np.float64)
    byte = lambda w, k: float((int(w) >> (8 * k)) & 255)
    with np.errstate(all="ignore"):
        lo = [np.array([byte(b[4 + a], k) * scale[a] + origin[a] for a in range(3)]).astype(np.float32) for k in range(4)]
        hi = [np.array([byte(b[7 + a], k) * scale[a] + origin[a] for a in range(3)]).astype(np.float32) for k in range(4)]
    return int(b[11]), nk, [(int(b[10]) >> (8 * k)) & 255 for k in range(4)], lo, hi


def check_wide(wide, nodes, order, n_tris):
    """Every record sits in exactly one slot of one leaf block, references are in range and each block is referenced
    once, the four boxes of an inner block are boxes of the binary hierarchy, and no walk can have more than
    stack_need (<= the kernel's 64) children waiting. Returns (inner blocks, leaf blocks)."""
    blocks, dest, root = wide["blocks"], wide["dest"], wide["root"]
    if n_tris == 0:
        assert root == T.BVH_NONE and len(blocks) == 0
        return 0, 0
    fl = blocks.view(np.float32)
    seen = np.zeros(len(blocks), np.int32)
    slots = {}
    box_lo, box_hi = nodes["lo"].astype(np.float64), nodes["hi"].astype(np.float64)
    inner = leaves = 0

    def ref_of(tag, idx):  # a child's tag + block index in the form of the root reference
        return ((tag >> 4) << 31) | (((tag >> 2) & 3) << 28) | idx

    def need_of(ref):
        nonlocal inner, leaves
        idx = ref & T.BVH_INDEX_MASK
        assert idx < len(blocks)
        seen[idx] += 1
        if ref & T.BVH_LEAF_BIT:
            cnt = (ref >> 28) & 3
            assert 1 <= cnt <= LEAF_MAX
            assert not blocks[idx].any(), "leaf blocks arrive empty: the device writes the triangles"
            for k in range(cnt):
                slots[(idx << 2) | k] = slots.get((idx << 2) | k, 0) + 1
            leaves += 1
            return 0
        inner += 1
        first, nk, tags, los, his = inner_children(blocks, idx)
        assert 2 <= nk <= 4 and first + nk <= len(blocks)
        assert [t & 3 for t in tags] == [0, 1, 2, 3], "a tag carries its slot"
        # every decoded box contains a box of the binary hierarchy and overshoots it by less than two cells of the block's grid
        if len(box_lo) <= 4000 or inner % 97 == 0:  # (all blocks of small hierarchies, a sample of large ones)
            cell = np.array([((int(blocks[idx, 3]) >> (8 * a)) & 255) << 23 for a in range(3)], np.uint32).view(np.float32).astype(np.float64)
            for k in range(nk):
                with np.errstate(all="ignore"):
                    ok = ((box_lo >= los[k]) & (box_hi <= his[k]) & (box_lo - los[k] < 2 * cell + 1e-30) & (his[k] - box_hi < 2 * cell + 1e-30)).all(axis=1)
                assert ok.any(), (idx, k)
        for k in range(nk, 4):  # empty slots: lo above hi, children are packed to the front (the walk counts the slots)
            assert all((int(blocks[idx, 4 + a]) >> (8 * k)) & 255 == 255 and (int(blocks[idx, 7 + a]) >> (8 * k)) & 255 == 0 for a in range(3)) and tags[k] == k
        return nk - 1 + max(need_of(ref_of(tags[k], first + k)) for k in range(nk))

    import sys
    sys.setrecursionlimit(10000)
    need = need_of(root)
    assert (seen == 1).all(), "every block hangs off exactly one reference"
    assert sorted(slots) == sorted(dest.tolist()) and all(v == 1 for v in slots.values()) and len(dest) == n_tris
    assert need == wide["stack_need"] <= T.BVH_STACK_CAP
    return inner, leaves


def walk(wide, rec_of_slot, org, d):
    """The device's walk (csrc/kernels.hip walk_bvh) with tmin = inf, in float32; returns the tested records, the
    blocks fetched and the deepest the stack got (waiting children, as the host's bound counts them)."""
    f = np.float32
    org, d = org.astype(f), d.astype(f)
    with np.errstate(all="ignore"):
        inv = np.where(np.abs(d) >= f(2.0 ** -100), f(1) / d, np.copysign(f(2.0 ** 100), d)).astype(f)
    neg = inv < 0
    blocks = wide["blocks"]
    fl = blocks.view(np.float32)
    tested, steps, deepest = [], 0, 0
    root = wide["root"]
    stack = []  # (key, first)
    cur = None if root == T.BVH_NONE else (root & T.BVH_INDEX_MASK, ((root >> 31) << 4) | (((root >> 28) & 3) << 2))
    while cur is not None:
        steps += 1
        idx, tag = cur
        cur = None
        if tag & 16:
            tested.extend(rec_of_slot[(idx << 2) | k] for k in range((tag >> 2) & 3))
        else:
            first, nk, tags, los, his = inner_children(blocks, idx)
            keys = []
            for k in range(4):
                lo, hi = los[k], his[k]
                near, far = np.where(neg, hi, lo), np.where(neg, lo, hi)  # what the lane's selects pick
                with np.errstate(all="ignore"):
                    tn = max(((near - org) * inv).max(), f(0))
                    tf = ((far - org) * inv).min()
                    hit = tn <= tf * f(1.000001) and k < nk
                bits = int(np.array([tn], f).view(np.uint32)[0]) if hit else 0x7F800000
                keys.append((bits & ~31) | tags[k])
            keys.sort()
            enter = [k for k in keys if k < 0x7F800000]
            assert all(k & 3 < nk for k in enter), "an empty slot was entered"
            if enter:
                cur = (first + (enter[0] & 3), enter[0])
                stack.extend((k, first) for k in reversed(enter[1:]))
                deepest = max(deepest, len(stack))
        if cur is None and stack:
            key, first = stack.pop()
            cur = (first + (key & 3), key)
    return tested, steps, deepest


def true_hits(wv, org, d):
    v0, e1, e2 = wv[:, 0], wv[:, 1] - wv[:, 0], wv[:, 2] - wv[:, 0]
    h = np.cross(d, e2)
    a = (e1 * h).sum(-1)
    with np.errstate(all="ignore"):
        f = 1.0 / a
        s = org - v0
        u = f * (s * h).sum(-1)
        q = np.cross(s, e1)
        v = f * (q * d).sum(-1)
        t = f * (e2 * q).sum(-1)
        return np.nonzero((a != 0) & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 0))[0]


MESHES = {
    "blob968": lambda: (S.blob_mesh(22, 23, seed=1), R.mat_mul(R.translate((0.4, -0.1, 0.2)), R.mat_mul(R.euler_yxz(0.6, -0.3, 0.2), R.scale_matrix((1.0, 0.7, 1.3))))),
    "box12": lambda: (R.box_triangles(), R.mat_mul(R.translate((1, 2, 3)), R.euler_yxz(0.3, 0.9, 0.0))),
    "flat48": lambda: (S.blob_mesh(6, 5, seed=9, smooth=False), R.identity4()),
    "single": lambda: (R.box_triangles()[:1], R.identity4()),
}


@pytest.mark.parametrize("name", sorted(MESHES))
def test_structure_boxes_and_walk(name):
    tris, xf = MESHES[name]()
    tris = R.as_records(tris, R.TRIANGLE)
    shape = R.model(0, tris, 0, len(tris), xf)
    nodes, order = T.bvh_build_host(shape, tris)
    assert len(nodes) >= 1 and len(order) == len(tris)
    check_structure(nodes, order, len(tris))
    wv = world_vertices(shape, tris)
    check_boxes(nodes, order, wv)
    wide = T.bvh_wide_host(shape, tris)
    inner, leaves = check_wide(wide, nodes, order, len(tris))
    assert leaves == int((nodes["leaf"] != 0).sum()) and not wide["balanced"]
    rec_of_slot = {int(s): r for r, s in enumerate(wide["dest"])}
    rng = np.random.RandomState(11)
    centre, radius = wv.reshape(-1, 3).mean(axis=0), np.abs(wv.reshape(-1, 3) - wv.reshape(-1, 3).mean(axis=0)).max() * 2.5
    n_hits = 0
    for k in range(300):
        org = centre + rng.normal(size=3) * radius * (0.3 if k % 3 == 0 else 1.0)  # some origins inside the mesh
        target = wv[rng.randint(len(wv))].mean(axis=0) + rng.normal(size=3) * 0.05
        d = target - org
        d /= np.linalg.norm(d)
        if k % 10 == 0:
            d[rng.randint(3)] = 0.0  # axis-parallel components exercise the safe inverse
        tested, steps, deepest = walk(wide, rec_of_slot, org, d)
        assert steps <= len(wide["blocks"]) and deepest <= wide["stack_need"]
        hit = true_hits(wv, org, d)
        rec_of = np.empty(len(order), np.int64)
        rec_of[order] = np.arange(len(order))
        missing = set(rec_of[hit].tolist()) - set(tested)
        assert not missing, f"ray {k}: triangles {sorted(missing)} are hit but were never tested"
        n_hits += len(hit)
    assert n_hits > 100


def test_100k_triangle_mesh_builds_fast_and_well_formed():
    import time
    shapes, tris, _ = S.mesh_scene(1, 224, 224, smooth=False)
    model = shapes[shapes["type"] == 2][0]
    t0 = time.time()
    nodes, order = T.bvh_build_host(model, tris)
    dt = time.time() - t0
    n = int(model["num_triangles"])
    depth = check_structure(nodes, order, n)
    assert n // LEAF_MAX <= int((nodes["leaf"] != 0).sum()) <= n
    assert depth <= 64 and dt < 5.0
    t0 = time.time()
    wide = T.bvh_wide_host(model, tris)
    dt = time.time() - t0
    inner, leaves = check_wide(wide, nodes, order, n)
    assert inner * 2 <= leaves and dt < 5.0  # four-wide: about a third as many inner blocks as leaves
    # subtrees of a mesh this size are built on several threads: the result must not depend on their timing
    nodes2, order2 = T.bvh_build_host(model, tris)
    assert nodes.tobytes() == nodes2.tobytes() and np.array_equal(order, order2)
    assert len(wide["blocks"]) * 128 < 9 << 20  # the whole hierarchy, triangles included: 8.5 MB


def test_a_hierarchy_too_deep_for_the_stack_is_rebuilt_balanced():
    """Triangles whose sizes and spacing grow geometrically: the SAH peels them off one at a time, a chain whose walk
    could have ~3 children waiting per level. The library must notice (stack_need) and fall back to median splits."""
    n = 400
    tris = np.zeros(n, R.TRIANGLE)
    for i in range(n):
        s, x = 1.02 ** i, 60.0 * (1.02 ** i)
        tris[i] = R.flat_triangle((0, 0, 1), (x, 0, 0), (x + s, 0, 0), (x, s, 0))
    shape = R.model(0, tris, 0, n)
    nodes, order = T.bvh_build_host(shape, tris)
    depth = check_structure(nodes, order, n)
    wide = T.bvh_wide_host(shape, tris)
    if depth > 70:  # the SAH tree really is a chain: only the balanced form fits
        assert wide["balanced"]
    assert wide["stack_need"] <= T.BVH_STACK_CAP
    forced = T.bvh_wide_host(shape, tris, force_balanced=True)
    assert forced["balanced"] and forced["stack_need"] <= 3 * 6  # ceil(log4(400 / 3)) + 1 levels of at most 3 waiting
    rec_of_slot = {int(s): r for r, s in enumerate(forced["dest"])}
    tested, steps, deepest = walk(forced, rec_of_slot, np.array([0.0, 0.1, 5.0]), np.array([1.0, 0.0, -0.001]))
    assert deepest <= forced["stack_need"]


@pytest.mark.parametrize("offset,scale", [((0, 0, 0), 1.0), ((3e6, -2e6, 1e6), 1.0), ((0.5, 0.25, -0.125), 2.0 ** -60), ((1e30, 0, 0), 1e25)])
def test_byte_boxes_of_flat_far_tiny_and_huge_meshes(offset, scale):
    """What the quantiser of the inner blocks has to get right at the edges: a planar mesh (zero extent on one axis: the
    coarsest useful grid there is the finest), coordinates millions of model sizes from the origin (the grid's origin
    swallows the bytes' low bits), a mesh of 2^-60 and one of 10^25 units: every decoded box contains its child's box,
    within two cells, and the float32 walk still reaches every triangle a double-precision ray hits."""
    n = 24
    tris = np.zeros(2 * n * n, R.TRIANGLE)
    k = 0
    for i in range(n):
        for j in range(n):
            p = [(i, j, 0), (i + 1, j, 0), (i, j + 1, 0), (i + 1, j + 1, 0)]
            tris[k] = R.flat_triangle((0, 0, 1), p[0], p[1], p[2])
            tris[k + 1] = R.flat_triangle((0, 0, 1), p[1], p[3], p[2])
            k += 2
    xf = R.mat_mul(R.translate(offset), R.scale_matrix((scale, scale, scale)))
    shape = R.model(0, tris, 0, len(tris), xf)
    nodes, order = T.bvh_build_host(shape, tris)
    check_structure(nodes, order, len(tris))
    wide = T.bvh_wide_host(shape, tris)
    check_wide(wide, nodes, order, len(tris))
    wv = world_vertices(shape, tris)
    rec_of_slot = {int(s): r for r, s in enumerate(wide["dest"])}
    rec_of = np.empty(len(order), np.int64)
    rec_of[order] = np.arange(len(order))
    rng = np.random.RandomState(5)
    centre = wv.reshape(-1, 3).mean(axis=0)
    n_hits = 0
    for _ in range(120):
        target = wv[rng.randint(len(wv))].mean(axis=0)
        org = target + np.array([rng.normal() * 5, rng.normal() * 5, 10 + rng.uniform(0, 30)]) * scale
        d = target - org
        d /= np.linalg.norm(d)
        tested, steps, deepest = walk(wide, rec_of_slot, org, d)
        assert steps <= len(wide["blocks"]) and deepest <= wide["stack_need"]
        hit = true_hits(wv, org.astype(np.float32).astype(np.float64), d.astype(np.float32).astype(np.float64))
        missing = set(rec_of[hit].tolist()) - set(tested)
        assert not missing, (offset, scale, sorted(missing))
        n_hits += len(hit)
    assert n_hits > 20 or scale != 1.0  # (far from the origin float32 rays are coarse: hits are fewer, never untested)


def test_degenerate_inputs():
    tris = np.zeros(8, R.TRIANGLE)
    for i in range(8):  # eight copies of ONE triangle: coincident centroids force the index split
        tris[i] = R.flat_triangle((0, 0, 1), (0, 0, 0), (1, 0, 0), (0, 1, 0))
    shape = R.model(0, tris, 0, 8)
    nodes, order = T.bvh_build_host(shape, tris)
    check_structure(nodes, order, 8)
    bad = tris.copy()
    bad["v"]["pos"][3, 1] = (np.nan, 0, np.inf)  # a non-finite vertex: the triangle lands in an all-embracing box
    nodes, order = T.bvh_build_host(R.model(0, tris, 0, 8), bad)
    check_structure(nodes, order, 8)
    assert np.isfinite(nodes["lo"]).all() and np.isfinite(nodes["hi"]).all()
    empty = R.model(0, tris, 0, 0)
    nodes, order = T.bvh_build_host(empty, tris)
    assert len(nodes) == 0 and len(order) == 0
    check_wide(T.bvh_wide_host(empty, tris), nodes, order, 0)
    nodes, order = T.bvh_build_host(shape, tris)
    check_wide(T.bvh_wide_host(shape, tris), nodes, order, 8)
    with pytest.raises(T.SrtError):
        T.bvh_build_host(R.sphere(0, (0, 0, 0), 1.0), tris)  # not a model
