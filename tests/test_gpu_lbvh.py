"""Device-side BVH build: rttnw_scene_set_bvh_builder(RTTNW_BVH_DEVICE_LBVH | RTTNW_BVH_DEVICE_SAH).

The tree only decides WHICH records a ray tests, never what a test returns, and exact ties are resolved by list
order, so a render through a device-built tree — the linear BVH or the binned-SAH one — must equal the render through
the host SAH tree bit for bit, in f64 and in f32.  Plus structural checks of the node records the kernels wrote.
"""
import ctypes as C

import numpy as np
import pytest

import util
from rttnw_amd import abi, render
from rttnw_amd import scene as S

pytestmark = pytest.mark.gpu

CHILD_EMPTY = util.CHILD_EMPTY

# (scene, param, width, height, spp): every primitive kind, instances, media, an un-BVH'd list, a deep tree
SCENES = [("cornell_box", 0, 64, 64, 8), ("smoke_cornell_box", 0, 48, 48, 4), ("final_scene", 0, 64, 64, 4),
          ("random_scene", 0, 64, 40, 4), ("simple_light", 0, 48, 32, 4), ("two_perlin_spheres", 0, 32, 32, 2),
          ("spheres_1m", 20000, 64, 64, 4)]


def nodes_of(gpu, sc):
    return util.nodes_of(gpu, sc)


DEVICE_BUILDERS = pytest.mark.parametrize("builder", [abi.BVH_DEVICE_LBVH, abi.BVH_DEVICE_SAH], ids=["lbvh", "dsah"])


@pytest.mark.parametrize("case", SCENES, ids=[c[0] for c in SCENES])
@pytest.mark.parametrize("precision", [abi.F64, abi.F32], ids=["f64", "f32"])
@DEVICE_BUILDERS
def test_lbvh_render_equals_sah_render(gpu, scenes_lib, earth, case, precision, builder):
    name, param, w, h, spp = case
    s_sah, setup = util.build(gpu, scenes_lib, name, earth, param)
    s_lbvh, _ = util.build(gpu, scenes_lib, name, earth, param, bvh=builder)
    bi = s_lbvh.build_info()
    assert bi.builder == builder and bi.n_prims == s_sah.build_info().n_prims
    cam, p = util.params_for(setup, w, h, spp, precision=precision, seed=5, collect_counters=1)
    lin_a, rgba_a, st_a = render.render_host(s_sah, cam, p)
    lin_b, rgba_b, st_b = render.render_host(s_lbvh, cam, p)
    assert np.array_equal(lin_a, lin_b), (name, np.abs(lin_a - lin_b).max())
    assert np.array_equal(rgba_a, rgba_b)
    assert st_a.rays == st_b.rays  # the same paths; node / primitive counts differ with the tree


@DEVICE_BUILDERS
@pytest.mark.parametrize("n_spheres", [5000, 70000])
def test_lbvh_structure(gpu, scenes_lib, builder, n_spheres):
    sc, _ = util.build(gpu, scenes_lib, "spheres_1m", None, n_spheres, bvh=builder)
    bi = sc.build_info()
    assert bi.device_ms > 0 and bi.n_prims == n_spheres + 1  # + the area light
    nodes, root = nodes_of(gpu, sc)
    assert len(nodes) == bi.n_prims - 1 and root == 0  # a binary tree over n leaves; the hierarchy's node 0 is its root
    seen_leaf, seen_node = set(), set()
    depth_max = 0
    stack = [(root, 1, np.full(3, -np.inf, np.float32), np.full(3, np.inf, np.float32))]
    while stack:
        i, depth, plo, phi = stack.pop()
        assert i not in seen_node
        seen_node.add(i)
        depth_max = max(depth_max, depth)
        nd = nodes[i]
        for lo, hi, ch in ((nd["lo0"], nd["hi0"], nd["child"][0]), (nd["lo1"], nd["hi1"], nd["child"][1])):
            assert (lo <= hi).all() and (lo >= plo).all() and (hi <= phi).all()  # child boxes nest in the parent's
            if ch >= 0:
                stack.append((int(ch), depth + 1, lo, hi))
            else:
                assert ch != CHILD_EMPTY
                bits = ~int(ch) & 0xFFFFFFFF
                assert (bits >> 26) & 3 == 0  # one record per leaf
                key = (bits >> 28, bits & 0x3FFFFFF)
                assert key not in seen_leaf
                seen_leaf.add(key)
    assert len(seen_node) == len(nodes) and len(seen_leaf) == bi.n_prims
    assert sum(1 for k, _ in seen_leaf if k == 0) == n_spheres  # PRIM_SPHERE = 0
    # the kernels walk the 4-wide collapse of this tree: the same leaves, and a stack bound that covers its deepest walk
    n4, root4 = util.nodes_of(gpu, sc, wide=True)
    leaves4, need, seen4 = util.check_wide_tree(n4, root4)
    assert len(seen4) == len(n4) == bi.n_nodes and len(leaves4) == bi.n_prims
    assert need + 1 == bi.stack_depth and bi.stack_depth <= 3 * ((depth_max + 1) // 2) + 1  # three pending children per wide level at most
    # the build is deterministic: atomics only ever feed order-independent sums, minima and maxima, and lists whose order nothing depends on
    sc2, _ = util.build(gpu, scenes_lib, "spheres_1m", None, n_spheres, bvh=builder)
    nodes2, _ = nodes_of(gpu, sc2)
    n4b, _ = util.nodes_of(gpu, sc2, wide=True)
    assert nodes.tobytes() == nodes2.tobytes() and n4.tobytes() == n4b.tobytes()


@DEVICE_BUILDERS
def test_lbvh_small_and_empty_worlds(gpu, builder):
    # 0 and 1 objects never reach the device builder; 2 objects is its smallest tree; 64 / 65 straddle the SAH builder's
    # wave-per-segment limit
    for n in (0, 1, 2, 3, 64, 65, 130):
        sc = S.Scene(gpu, 1)
        sc.set_bvh_builder(builder)
        red = sc.lambertian(sc.solid(0.8, 0.2, 0.2))
        world = sc.list()
        for k in range(n):
            sc.push(world, sc.sphere((2.5 * (k % 7), 2.5 * (k // 49), -5.0 - 2.5 * ((k // 7) % 7)), 1.0, red))
        sc.set_world(world)
        sc.commit()
        ref = S.Scene(gpu, 1)
        red = ref.lambertian(ref.solid(0.8, 0.2, 0.2))
        world = ref.list()
        for k in range(n):
            ref.push(world, ref.sphere((2.5 * (k % 7), 2.5 * (k // 49), -5.0 - 2.5 * ((k // 7) % 7)), 1.0, red))
        ref.set_world(world)
        ref.commit()
        cam = abi.CameraDesc()
        cam.lookfrom[:] = (2.0, 0.5, 3.0); cam.lookat[:] = (2.0, 0.0, -5.0); cam.view_up[:] = (0.0, 1.0, 0.0)
        cam.vertical_fov = 50.0; cam.aspect_ratio = 1.0; cam.aperture = 0.0; cam.focus_distance = 1.0
        cam.open_time = 0.0; cam.close_time = 1.0
        p = S.make_params(32, 32, 4, background=(0.7, 0.8, 1.0), precision=abi.F64)
        a, _, _ = render.render_host(sc, cam, p)
        b, _, _ = render.render_host(ref, cam, p)
        assert np.array_equal(a, b), n


@DEVICE_BUILDERS
def test_coincident_and_nested_leaves(gpu, builder):
    """Leaves the builders cannot separate by centroid: 150 spheres with ONE centre (radii differ: nested shells), beside 40
    copies of one sphere — more than a wave's worth, so the SAH builder's large-segment path meets an extent of zero on every
    axis and has to halve by position.  The render must still equal the host tree's (the later list item wins exact ties)."""
    def world(sc):
        mats = [sc.lambertian(sc.solid(0.2 + 0.1 * (k % 7), 0.5, 0.9 - 0.1 * (k % 5))) for k in range(6)]
        glass = sc.dielectric(1.5)
        w = sc.list()
        for k in range(150):
            sc.push(w, sc.sphere((0.0, 0.0, -6.0), 0.5 + 0.01 * k, glass if k % 3 == 0 else mats[k % 6]))
        for k in range(40):
            sc.push(w, sc.sphere((4.0, 0.0, -6.0), 1.0, mats[k % 6]))
        sc.set_world(w)
        sc.commit()
    a, b = S.Scene(gpu, 1), S.Scene(gpu, 1)
    b.set_bvh_builder(builder)
    world(a); world(b)
    cam = abi.CameraDesc()
    cam.lookfrom[:] = (2.0, 0.5, 3.0); cam.lookat[:] = (2.0, 0.0, -6.0); cam.view_up[:] = (0.0, 1.0, 0.0)
    cam.vertical_fov = 50.0; cam.aspect_ratio = 1.0; cam.aperture = 0.0; cam.focus_distance = 1.0
    cam.open_time = 0.0; cam.close_time = 1.0
    for precision in (abi.F64, abi.F32):
        p = S.make_params(48, 48, 4, background=(0.7, 0.8, 1.0), precision=precision)
        x, _, _ = render.render_host(a, cam, p)
        y, _, _ = render.render_host(b, cam, p)
        assert np.array_equal(x, y) and np.isfinite(x).all() and x.max() > 0


def test_builder_choice_is_frozen_by_commit(gpu):
    sc = S.Scene(gpu, 1)
    sc.set_world(sc.list())
    sc.commit()
    assert gpu.scene_set_bvh_builder(sc.handle, abi.BVH_DEVICE_LBVH) == -2  # RTTNW_ERR_STATE
    sc2 = S.Scene(gpu, 1)
    assert gpu.scene_set_bvh_builder(sc2.handle, 7) == -1  # RTTNW_ERR_INVALID
    assert gpu.scene_set_bvh_builder(sc2.handle, abi.BVH_DEVICE_SAH) == 0


def test_default_builder_picks_by_tree_size(gpu, scenes_lib, earth):
    """RTTNW_BVH_AUTO (ABI 3, the default): trees below RTTNW_BVH_AUTO_DEVICE_LEAVES = 100 000 leaves are built on the host (no device build
    time; the small-scene leaf tuning applies: final_scene keeps the host build's records), larger ones by the device SAH builder — and the
    image is the host-built tree's, bit for bit."""
    sc, setup = util.build(gpu, scenes_lib, "final_scene", earth)                      # default builder
    info = sc.build_info()
    assert info.builder == abi.BVH_AUTO and info.device_ms == 0.0
    sh, _ = util.build(gpu, scenes_lib, "final_scene", earth, bvh=abi.BVH_HOST_SAH)
    assert sh.build_info().n_nodes == info.n_nodes
    sc, setup = util.build(gpu, scenes_lib, "spheres_1m", param=120000)
    info = sc.build_info()
    assert info.builder == abi.BVH_AUTO and info.device_ms > 0.0
    small, _ = util.build(gpu, scenes_lib, "spheres_1m", param=60000)
    assert small.build_info().device_ms == 0.0
    sd, _ = util.build(gpu, scenes_lib, "spheres_1m", param=120000, bvh=abi.BVH_DEVICE_SAH)
    sh, _ = util.build(gpu, scenes_lib, "spheres_1m", param=120000, bvh=abi.BVH_HOST_SAH)
    assert sd.build_info().n_nodes == info.n_nodes
    for prec in (abi.F64, abi.F32):
        cam, p = util.params_for(setup, 48, 40, 4, precision=prec, seed=5)
        a, ra, _ = render.render_host(sc, cam, p)
        b, rb, _ = render.render_host(sh, cam, p)
        assert np.array_equal(a, b) and np.array_equal(ra, rb)


@DEVICE_BUILDERS
def test_full_size_config5_invariants(gpu, scenes_lib, builder):
    """BASELINE config 5 at its full size (10^6 spheres, 1024x1024) is beyond the oracle's reference-shaped builder, so
    check size-independent properties: the device-built tree and the host tree render the same image bit for bit,
    a run repeats exactly, and the partition over 8 ranks reassembles to the single-rank image."""
    s_sah, setup = util.build(gpu, scenes_lib, "spheres_1m", None, 0)
    s_lbvh, _ = util.build(gpu, scenes_lib, "spheres_1m", None, 0, bvh=builder)
    assert s_sah.build_info().n_prims == 1000001 and gpu.debug_scene_nodes(s_lbvh.handle, None, 0, None) == 1000000
    assert 330000 <= s_lbvh.build_info().n_nodes <= 520000                       # 4-wide records: a third to a half of the binary nodes
    cam, p = util.params_for(setup, 1024, 1024, 2, precision=abi.F32, seed=7)
    a, rgba_a, st = render.render_host(s_sah, cam, p)
    b, rgba_b, _ = render.render_host(s_lbvh, cam, p)
    a2, _, _ = render.render_host(s_sah, cam, p)
    assert (st.reserved & 1) == 1 and st.samples == 1024 * 1024 * 2
    assert np.array_equal(a, a2) and np.array_equal(a, b) and np.array_equal(rgba_a, rgba_b)
    assert np.isfinite(a).all() and a.min() >= 0 and (rgba_a[..., 3] == 255).all()
    # one rank's tiles of an 8-way partition hold exactly the single-rank values
    from rttnw_amd import tiles
    import torch
    cam8, p8 = util.params_for(setup, 1024, 1024, 2, precision=abi.F32, seed=7, tile_rank=5, tile_world=8)
    r = render.DeviceRenderer(s_sah, cam8, p8)
    r.trace()
    torch.cuda.synchronize()
    want = tiles.pack_rank(a, 5, 8)
    got = r.packed.cpu().numpy().astype(np.float64)
    assert np.array_equal(got[:, :3], want[:, :3])


@DEVICE_BUILDERS
def test_record_numbering_is_level_order_and_result_free(gpu, scenes_lib, builder, monkeypatch):
    """The device collapse numbers the 4-wide records in LEVEL order (round 6: a record's inner children side by side — the two 64-byte quantised
    records of a 128-byte line are siblings); RTTNW_NODE_ORDER=pre keeps the binary tree's pre-order of rounds 1-5.  The numbering is layout only:
    the same records (as a multiset of boxes and leaves), the same image bit for bit, the same node visits and record tests.  Level order itself:
    every record's inner children are consecutive numbers, and a child's number is larger than its parent's."""
    n = 120000
    sc, setup = util.build(gpu, scenes_lib, "spheres_1m", None, n, bvh=builder)
    monkeypatch.setenv("RTTNW_NODE_ORDER", "pre")
    sc_pre, _ = util.build(gpu, scenes_lib, "spheres_1m", None, n, bvh=builder)
    monkeypatch.delenv("RTTNW_NODE_ORDER")
    n4, root = util.nodes_of(gpu, sc, wide=True)
    n4p, rootp = util.nodes_of(gpu, sc_pre, wide=True)
    assert root == rootp == 0 and len(n4) == len(n4p)
    siblings = 0
    for i in range(len(n4)):
        inner = [int(c) for c in n4[i]["child"] if c >= 0]
        assert all(c > i for c in inner)
        assert inner == list(range(inner[0], inner[0] + len(inner))) if inner else True
        siblings += len(inner) >= 2
    assert siblings > len(n4) // 8
    # pre-order: a record's FIRST inner child is the next record; level order puts it a level away
    first_next = sum(1 for i in range(len(n4p)) if any(int(c) == i + 1 for c in n4p[i]["child"][:1] if c >= 0))
    assert first_next > 0

    def boxes(nodes):   # the records as sets of (boxes, leaf children), whatever their numbers
        out = []
        for nd in nodes:
            leaves = tuple(sorted(int(c) for c in nd["child"] if c < 0))
            out.append((nd["lo"].tobytes(), nd["hi"].tobytes(), leaves, sum(1 for c in nd["child"] if c >= 0)))
        return sorted(out)
    assert boxes(n4) == boxes(n4p)
    cam, p = util.params_for(setup, 96, 96, 3, precision=abi.F64_STRICT, seed=9, collect_counters=1)
    a, ra, sa = render.render_host(sc, cam, p)
    b, rb, sb = render.render_host(sc_pre, cam, p)
    assert np.array_equal(a, b) and np.array_equal(ra, rb)
    assert (sa.rays, sa.nodes_visited, sa.prims_tested) == (sb.rays, sb.nodes_visited, sb.prims_tested)
