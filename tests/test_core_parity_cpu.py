"""The product's tracing core + lowering (host build, tests/hostsim) against the oracle.

This is the logic gate that can run without a GPU: same rt_core.hpp / scene_lower.cpp the kernels are
compiled from, same job/chunk accumulation order as the trace + resolve kernels.  f64 must agree with
the oracle to rounding on every scene of the catalogue; f32 must agree statistically.
"""
import ctypes as C

import numpy as np
import pytest

import util
from golden_cases import CASES, load
from oracle import rto
from rttnw_amd import abi


@pytest.mark.parametrize("world_spheres", [0, 1, 2], ids=["spheres_in_groups", "spheres_in_world_space", "world_boxes_group_frame_test"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_core_f64_equals_golden(hostsim, scenes_lib, earth, case, world_spheres, monkeypatch):
    """The host build of the tracing core against the oracle's golden images.  With the spheres of transformed groups left
    in their groups' trees (RTTNW_WORLD_SPHERES=0) the core does the oracle's arithmetic: equal to rounding.  By default
    the lowering tests those spheres in world space (scene_lower.cpp): t is the root of the same quadratic written in
    another frame, a last-place difference that a few bounces off small spheres amplify — the T1 bar of the device tests
    (1e-9 on >= 99.9 % of the pixels) applies.  RTTNW_WORLD_SPHERES=2 is what RTTNW_F64_STRICT renders since round 5: the copies' world-space
    BOXES in the top tree (culling never shapes a result), the sphere test in the group's frame (leaf kind PRIM_SPHERE_WC, rt_core.hpp
    sphere_wc_t): the oracle's arithmetic again, held to the same 1e-12."""
    monkeypatch.setenv("RTTNW_WORLD_SPHERES", str(world_spheres))
    key, name, w, h, spp, chunk, param = case
    sc, setup = util.build(hostsim, scenes_lib, name, earth, param)
    cam, p = util.params_for(setup, w, h, spp, spp_chunk=chunk, precision=abi.F64)
    hostsim.lib.hostsim_max_stack()
    lin, _ = util.hostsim_render(hostsim, sc, cam, p)
    # the LDS traversal stacks of the device are sized by the lowering's bound: it must cover what the walk really used
    dims = (C.c_uint32 * 8)()
    hostsim.lib.hostsim_scene_dims(sc.handle, dims)
    assert hostsim.lib.hostsim_max_stack() <= dims[7], (key, dims[7])
    g = load()[key + "_linear"]
    d = np.abs(lin - g)
    if world_spheres == 1:
        assert (d.max(axis=2) <= 1e-9).mean() >= 0.999, (key, d.max())
    else:  # recursion (oracle) vs throughput loop (core) differ by rounding only
        assert d.max() <= 1e-12 * max(1.0, g.max()), (key, d.max())


def test_big_cloud_takes_the_sphere_index_as_its_material_slot(hostsim, oracle, scenes_lib):
    """A LEAN cloud of >= 65 536 spheres (spheres_1m) is lowered so that sphere i's material slot holds i — kept where the records stay in creation
    order (the device builder: one record per leaf), re-established by moving the materials behind the spheres where a builder reordered them (the
    host SAH build, which this test runs) —; the scene view then carries NO slot array (FlatScene::sphere_mat_is_index, SceneView::sphere_mat ==
    nullptr) and make_record takes the index itself: one L2-miss line per hit less on the device.  The host build of the core with that view
    against the oracle (median-split tree: the reference's builder cannot make this one) on a small frame: every pixel within 1e-12 — the
    materials included, or the colours would differ."""
    n = 70000
    sh, setup = util.build(hostsim, scenes_lib, "spheres_1m", None, n)
    so, _ = util.build(oracle, scenes_lib, "spheres_1m", None, n, bvh=rto.BVH_MEDIAN_SPLIT)
    flags = hostsim.lib.hostsim_scene_flags(sh.handle)
    assert flags & 1, "spheres_1m is a LEAN scene"
    cam, p = util.params_for(setup, 40, 32, 3, spp_chunk=3, precision=abi.F64)
    lin, _ = util.hostsim_render(hostsim, sh, cam, p)
    lo, _, _ = rto.render(so, cam, p, want_rgba8=False)[:3]
    assert np.abs(lin - lo).max() <= 1e-12 * max(1.0, lo.max()), np.abs(lin - lo).max()
    assert flags & 2, "the big LEAN cloud must come out with slot i == i"
    small, _ = util.build(hostsim, scenes_lib, "spheres_1m", None, 3000)
    assert hostsim.lib.hostsim_scene_flags(small.handle) == 1   # (a small cloud keeps its slots: they are staged in LDS)


def test_core_counts_match_oracle_rays(hostsim, oracle, scenes_lib, earth):
    """Same number of world.hit() calls as the oracle; far fewer node visits than the reference's tree."""
    for name in ("cornell_box", "final_scene"):
        so, setup = util.build(oracle, scenes_lib, name, earth)
        sh, _ = util.build(hostsim, scenes_lib, name, earth)
        cam, p = util.params_for(setup, 32, 32, 8, spp_chunk=4, collect_counters=1)
        _, _, st_o = rto.render(so, cam, p)
        _, st_h = util.hostsim_render(hostsim, sh, cam, p)
        assert st_h.rays == st_o.rays and st_h.samples == st_o.samples == 32 * 32 * 8
        if name == "final_scene":
            assert st_h.nodes_visited < st_o.nodes_visited / 3


def test_core_f32_is_statistically_equal(hostsim, oracle, scenes_lib, earth):
    """f32 shares every uniform's top 24 bits with f64, so images differ only through rare branch flips."""
    for name, tol in (("cornell_box", 0.001), ("final_scene", 0.002)):
        so, setup = util.build(oracle, scenes_lib, name, earth)
        sh, _ = util.build(hostsim, scenes_lib, name, earth)
        cam, p = util.params_for(setup, 48, 48, 32)
        lo, _, _ = rto.render(so, cam, p)
        p.precision = abi.F32
        lh, _ = util.hostsim_render(hostsim, sh, cam, p)
        assert abs(lh.mean() - lo.mean()) / lo.mean() < tol
        # most pixels agree closely; flips touch a minority
        assert (np.abs(lh - lo).max(axis=2) < 1e-3).mean() > 0.9


def test_chunking_changes_only_rounding(hostsim, scenes_lib):
    sc, setup = util.build(hostsim, scenes_lib, "cornell_box")
    imgs = []
    for chunk in (1, 3, 16, 0):
        cam, p = util.params_for(setup, 24, 24, 16, spp_chunk=chunk)
        imgs.append(util.hostsim_render(hostsim, sc, cam, p)[0])
    for im in imgs[1:]:
        assert np.abs(im - imgs[0]).max() < 1e-13


def test_lowering_shapes(hostsim, scenes_lib, earth):
    dims = (C.c_uint32 * 8)()
    sc, _ = util.build(hostsim, scenes_lib, "cornell_box")
    hostsim.lib.hostsim_scene_dims(sc.handle, dims)
    nodes, sph, mov, rect, box, inst, media, stack = list(dims)
    assert (sph, mov, rect, box, inst, media) == (0, 0, 6, 2, 2, 0)             # Cube = one box record
    sc, _ = util.build(hostsim, scenes_lib, "final_scene", earth)
    hostsim.lib.hostsim_scene_dims(sc.handle, dims)
    nodes, sph, mov, rect, box, inst, media, stack = list(dims)
    # 1000 cluster spheres + their 1000 world-space copies (what the walk tests) + 5 free + 2 boundaries; the cluster's
    # chain stays as an instance record without a tree
    assert (sph, mov, rect, box, inst, media) == (2007, 1, 1, 400, 1, 2)
    assert nodes < 1500 and stack <= 40
    sc, _ = util.build(hostsim, scenes_lib, "smoke_cornell_box")
    hostsim.lib.hostsim_scene_dims(sc.handle, dims)
    assert list(dims)[3:7] == [6, 2, 2, 2]                                       # box boundaries under rotate+translate
    import graph_shapes
    sc, _, _ = graph_shapes.build(hostsim, "many_moved_spheres")
    hostsim.lib.hostsim_scene_dims(sc.handle, dims)
    nodes, sph, mov, rect, box, inst, media, stack = list(dims)
    assert (sph, inst) == (600 + 512, 600)       # 512 chains can be named by a world-space copy; the other 88 spheres stay instances


def test_unsupported_graphs_are_rejected(hostsim):
    from rttnw_amd import scene as S
    sc = S.Scene(hostsim)
    m = sc.lambertian((0.5, 0.5, 0.5))
    x = sc.sphere((0, 0, 0), 1.0, m)
    for k in range(9):                                                           # nine wrappers around one object (limit: 8)
        x = sc.translate(x, (0.1, 0, 0))
    sc.set_world(sc.list([x]))
    with pytest.raises(abi.RttnwError, match="UNSUPPORTED"):
        sc.commit()
    sc2 = S.Scene(hostsim)
    m2 = sc2.lambertian((0.5, 0.5, 0.5))
    rect = sc2.rectangle(abi.XY, (0, 1), (0, 1), 0.0, m2)
    sc2.set_world(sc2.list([sc2.constant_medium(rect, 0.1, (1, 1, 1))]))         # non-closed boundary
    with pytest.raises(abi.RttnwError, match="UNSUPPORTED"):
        sc2.commit()
    sc3 = S.Scene(hostsim)
    with pytest.raises(abi.RttnwError):
        sc3.commit()                                                             # world not set
    sc4 = S.Scene(hostsim)
    w4 = sc4.list()
    for k in range(17):                                                          # 17 media: one RNG slot too many
        sc4.push(w4, sc4.constant_medium(sc4.sphere((3.0 * k, 0, 0), 1.0, sc4.dielectric(1.5)), 0.1, (1, 1, 1)))
    sc4.set_world(w4)
    with pytest.raises(abi.RttnwError, match="UNSUPPORTED"):
        sc4.commit()


@pytest.mark.parametrize("shape", sorted(__import__("graph_shapes").SHAPES))
def test_graph_shapes_the_trait_objects_allow(hostsim, oracle, shape):
    """Any Hittable can be wrapped, nested and used as a medium boundary in the reference (hittable.rs:51-65,731): the
    lowering takes those graphs too (tests/graph_shapes.py) and the host build of the core agrees with the oracle."""
    import graph_shapes
    sh, cam, p = graph_shapes.build(hostsim, shape)
    so, _, _ = graph_shapes.build(oracle, shape)
    lin, _ = util.hostsim_render(hostsim, sh, cam, p)
    lo, _, _ = rto.render(so, cam, p)
    assert np.abs(lin - lo).max() <= 1e-11 * max(1.0, lo.max()), shape
    assert lo.std() > 0.01                                                        # the scene is really in view


def test_sample_ranges_compose(hostsim, oracle, scenes_lib):
    """rttnw_params.sample_begin: passes over disjoint sample ranges are the single render of their union (the same
    keyed draws), and the oracle follows the same convention."""
    sc, setup = util.build(hostsim, scenes_lib, "cornell_box")
    so, _ = util.build(oracle, scenes_lib, "cornell_box")
    cam, p_all = util.params_for(setup, 24, 24, 12, spp_chunk=4)
    whole = util.hostsim_render(hostsim, sc, cam, p_all)[0]
    parts = []
    for begin, n in ((0, 4), (4, 8)):
        _, p = util.params_for(setup, 24, 24, n, spp_chunk=4, sample_begin=begin)
        parts.append((n, util.hostsim_render(hostsim, sc, cam, p)[0]))
        assert np.abs(parts[-1][1] - rto.render(so, cam, p)[0]).max() <= 1e-12
    mean = sum(n * im for n, im in parts) / 12
    assert np.abs(mean - whole).max() <= 1e-12 * max(1.0, whole.max())
    assert np.abs(parts[0][1] - parts[1][1]).max() > 1e-3  # different samples, different estimates


def test_chunk_schedule_and_launch_split(hostsim):
    """plan_chunks / launch_chunks / plan_jobs (rt_types.hpp) for small, ordinary and very large renders: the chunks partition
    [0, spp) exactly and end on single-sample chunks, the schedule depends on spp ALONE (not on the image or on how many tiles
    a rank owns — so the per-pixel chain, hence the image, is the same for any tile_world), the chunk sums of one launch stay
    within the budget (the 4 GiB fallback, a device's, a shrunk one), job indices of a launch below 2^32, and a launch takes whole job groups (16 chunks) wherever
    the budget allows one."""
    out = (C.c_uint32 * 6)()
    hostsim.lib.hostsim_plan.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32)]
    cases = [(1, 1), (5, 10), (31, 10), (32, 10), (33, 10), (1000, 10000), (5000, 10000), (8000, 1250), (10000, 5000),
             (10000, 40000), (100000, 40000), (7, 40000), (1000000, 160000), (1000, 4000000)]
    for spp, tiles in cases:
        for bytes_per_sum in (12, 24):
            schedules = set()
            for world, budget in ((1, 4 * 2**30), (2, 24 * 2**30), (3, 4 * 2**30), (4, 2**30), (8, 24 * 2**30)):  # (the fallback, an MI355X's, a shrunk one)
                my_tiles = -(-tiles // world)
                assert hostsim.lib.hostsim_plan(spp, 0, my_tiles, bytes_per_sum, 0 if budget == 4 * 2**30 else budget, out) == 0, (spp, tiles, world)
                chunk, n_main, n_chunks, per_launch, launches, n_jobs = list(out)
                schedules.add((chunk, n_main, n_chunks))
                assert launches == -(-n_chunks // per_launch) and 1 <= per_launch <= n_chunks
                assert per_launch * my_tiles * 64 * bytes_per_sum <= budget or per_launch == 1
                assert per_launch % 16 == 0 or per_launch == n_chunks or per_launch < 16
                assert n_jobs >= min(per_launch, n_chunks) * my_tiles * 64 and n_jobs < 2**32
            assert len(schedules) == 1, (spp, tiles, schedules)
            assert chunk == 4 and n_main * 4 + (n_chunks - n_main) == spp, (spp, list(out))
            assert n_chunks - n_main >= min(spp, max(1, spp // 32))                  # always tapered
    assert hostsim.lib.hostsim_plan(40, 7, 100, 12, 0, out) == 0 and list(out)[:3] == [7, 6, 6]   # explicit chunking is uniform
    # BASELINE configs in f64.  With the fallback budget of 4 GiB: the headline frame (800x800 spp 1000, 4.2 GB of sums) is ONE launch,
    # spp 5000 six launches of 240 chunks, one rank of 8 of 1600x1600 spp 10000 six, the whole frame on one GPU 43 of 64 chunks.
    assert hostsim.lib.hostsim_plan(1000, 0, 10000, 24, 0, out) == 0 and list(out)[2:5] == [274, 274, 1]
    assert hostsim.lib.hostsim_plan(5000, 0, 10000, 24, 0, out) == 0 and list(out)[3:5] == [240, 6]
    assert hostsim.lib.hostsim_plan(10000, 0, 5000, 24, 0, out) == 0 and list(out)[3:5] == [464, 6]
    assert hostsim.lib.hostsim_plan(10000, 0, 40000, 24, 0, out) == 0 and list(out)[3:5] == [64, 43]
    # With an MI355X's 24 GiB (a twelfth of its HBM): every one of them is ONE launch but the whole 1600x1600 frame on one GPU (seven)
    mi355x = 24 * 2**30
    assert hostsim.lib.hostsim_plan(5000, 0, 10000, 24, mi355x, out) == 0 and list(out)[3:5] == [1367, 1]
    assert hostsim.lib.hostsim_plan(10000, 0, 5000, 24, mi355x, out) == 0 and list(out)[3:5] == [2734, 1]
    assert hostsim.lib.hostsim_plan(10000, 0, 40000, 24, mi355x, out) == 0 and list(out)[4] == 7


def test_core_chain_equals_oracle(hostsim, oracle, scenes_lib):
    """The host build of the core folds a pixel as ONE chain of chunk sums in chunk order (what the device's resolve step
    continues from launch to launch): it differs from the oracle's flat fold by rounding only, for the tapered and for an
    explicit chunking."""
    sh, setup = util.build(hostsim, scenes_lib, "cornell_box")
    so, _ = util.build(oracle, scenes_lib, "cornell_box")
    for chunk in (0, 3):
        cam, p = util.params_for(setup, 48, 48, 50, spp_chunk=chunk, seed=4)
        lin, _ = util.hostsim_render(hostsim, sh, cam, p)
        lo, _, _ = rto.render(so, cam, p)
        assert np.abs(lin - lo).max() <= 1e-12 * max(1.0, lo.max())


@pytest.mark.parametrize("name", ["cornell_box", "final_scene", "smoke_cornell_box", "random_scene"])
def test_core_per_bounce_records_equal_oracle(hostsim, oracle, scenes_lib, earth, name):
    """SURVEY section 4's second tier on the host build of the core: every world.hit() of 120 paths per scene — t, p, normal,
    front_face, material, (u, v) — against the oracle's recursion (the device probe kernel runs the same comparison in
    tests/test_gpu_parity.py)."""
    so, setup = util.build(oracle, scenes_lib, name, earth)
    sh, _ = util.build(hostsim, scenes_lib, name, earth)
    cam, p = util.params_for(setup, 40, 40, 4, seed=9)
    hostsim.lib.hostsim_probe_path.restype = C.c_int
    hostsim.lib.hostsim_probe_path.argtypes = [C.c_void_p, C.POINTER(abi.CameraDesc), C.POINTER(abi.Params), C.c_uint32, C.c_uint32,
                                               C.c_uint32, C.c_void_p, C.c_uint32]
    rng = np.random.default_rng(5)
    pairs = [(int(rng.integers(40)), int(rng.integers(40)), int(rng.integers(4))) for _ in range(120)]
    n, bounces, _ = util.compare_paths(lambda x, y, s: util.product_probe(hostsim.lib.hostsim_probe_path, hostsim, sh, cam, p, x, y, s),
                                       lambda x, y, s: rto.probe_path(so, cam, p, x, y, s), pairs,
                                       growth=8.0 if name in ("final_scene", "random_scene") else 1.0)  # (world-space copies of the cluster's spheres; small spheres)
    assert n == 120 and bounces > 150



def test_ball_draw_shortcut_is_the_full_rejection_loop(hostsim):
    """random_in_unit_space (vec3.rs:149-160) in the product core decides most candidates from the ONE word that holds the
    leading 21 bits of their three uniforms and fetches the other three words only for the accepted candidate (or when the
    leading bits leave the test `|v|^2 < 1` open).  Against the plain loop over full candidates, restated here in numpy from
    the draw spec, for 300 000 keys: the f64 vector is identical, the f32 vector is the f32 view of the
    same 53-bit uniforms (its accept decision may differ from f64's only within f32 rounding of the sphere)."""
    import ctypes as C
    GAMMA = 0x9E3779B97F4A7C15
    M = (1 << 64) - 1

    def mix64(z):
        z = z.copy()
        z ^= z >> np.uint64(30); z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27); z *= np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))

    rng = np.random.default_rng(11)
    n, bounce = 300000, 3
    keys = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    out64 = np.zeros((n, 3)); out32 = np.zeros((n, 3), dtype=np.float32)
    hostsim.lib.hostsim_ball.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    assert hostsim.lib.hostsim_ball(n, keys.ctypes.data, bounce, out64.ctypes.data, out32.ctypes.data) == 0
    with np.errstate(over="ignore"):
        def mixd(z):   # the mixer of a draw: folds by 32 around two products with 32-bit constants
            z = z.copy()
            z ^= z >> np.uint64(32); z *= np.uint64(0x9E3779B1)
            z ^= z >> np.uint64(32); z *= np.uint64(0x85EBCA6B)
            return z ^ (z >> np.uint64(32))
        word = lambda ctr: mixd(keys + np.uint64(((ctr + 1) * GAMMA) & M))   # noqa: E731
        want = np.zeros((n, 3)); want32 = np.zeros((n, 3)); todo = np.ones(n, dtype=bool)
        for it in range(64):
            base = (bounce + 1) * 1024 + 32 + 4 * it
            h = word(base)
            fields = [h >> np.uint64(43), (h >> np.uint64(22)) & np.uint64(0x1FFFFF), (h >> np.uint64(1)) & np.uint64(0x1FFFFF)]
            second, third = word(base + 1), word(base + 2)
            lows = [second >> np.uint64(32), second & np.uint64(0xFFFFFFFF), third >> np.uint64(32)]
            m = [(f << np.uint64(32)) | lo for f, lo in zip(fields, lows)]
            v = np.stack([2.0 * (x.astype(np.float64) * 2.0**-53) - 1.0 for x in m], axis=1)
            v32 = np.stack([2.0 * ((x >> np.uint64(29)).astype(np.float64) * 2.0**-24) - 1.0 for x in m], axis=1)
            acc = todo & ((v * v).sum(axis=1) < 1.0)
            want[acc] = v[acc]; want32[acc] = v32[acc]
            todo &= ~acc
            if not todo.any():
                break
    assert not todo.any() and np.array_equal(out64, want)
    same = np.abs(out32 - want32).max(axis=1) <= 2e-7          # f32 accepted the same candidate (all but those on the sphere's f32 skin)
    assert same.mean() > 0.9999 and ((out32.astype(np.float64) ** 2).sum(axis=1) < 1.0 + 1e-6).all()


@pytest.mark.parametrize("name,param,w,h,spp", [("cornell_box", 0, 48, 48, 6), ("final_scene", 0, 48, 48, 6), ("spheres_1m", 30000, 64, 64, 6),
                                                ("random_scene", 0, 64, 36, 6), ("smoke_cornell_box", 0, 40, 40, 4)])
def test_quantised_records_never_change_an_image(hostsim, scenes_lib, earth, name, param, w, h, spp, monkeypatch):
    """The decoupled kernels walk the trees through 8-bit quantised records made by bvh_quant.hpp from the f32 ones (rt_types.hpp Bvh4QNode; rt_core.hpp
    trav_node_step4q; HOSTSIM_QUANT=1 in the host build of the same code).  Conservative boxes only ever OPEN more nodes, so the image is bit-identical in
    both precisions and the same world.hit() calls are made; the visits grow by a few per cent."""
    sc, setup = util.build(hostsim, scenes_lib, name, earth, param)
    for prec in (abi.F64, abi.F32):
        cam, p = util.params_for(setup, w, h, spp, precision=prec, collect_counters=1, seed=9)
        monkeypatch.delenv("HOSTSIM_QUANT", raising=False)
        a, sta = util.hostsim_render(hostsim, sc, cam, p)
        for mode, growth in (("1", 1.25),):
            monkeypatch.setenv("HOSTSIM_QUANT", mode)
            b, stb = util.hostsim_render(hostsim, sc, cam, p)
            assert np.array_equal(a, b) and sta.rays == stb.rays, mode
            assert sta.nodes_visited <= stb.nodes_visited <= sta.nodes_visited * growth, (mode, sta.nodes_visited, stb.nodes_visited)
            assert sta.prims_tested <= stb.prims_tested



@pytest.mark.parametrize("name,w,h,spp", [("cornell_box", 40, 40, 6), ("final_scene", 40, 40, 4), ("random_scene", 48, 27, 4), ("smoke_cornell_box", 32, 32, 4)])
def test_f64_box_tests_only_cull(hostsim, scenes_lib, earth, name, w, h, spp, monkeypatch):
    """The f64 kernels test the f32 boxes in f32 with the error bounds folded into per-walk constants (rt_core.hpp slab_ray / slab_hit4):
    a box the exact test would pass must always pass.  Held against a walk that never culls (HOSTSIM_QUANT=3: every used slot's box opened to the
    whole space, unused slots left inverted): same image bit for bit, same world.hit() calls, fewer visits."""
    sc, setup = util.build(hostsim, scenes_lib, name, earth)
    cam, p = util.params_for(setup, w, h, spp, precision=abi.F64, collect_counters=1, seed=11)
    monkeypatch.delenv("HOSTSIM_QUANT", raising=False)
    a, sta = util.hostsim_render(hostsim, sc, cam, p)
    monkeypatch.setenv("HOSTSIM_QUANT", "3")
    b, stb = util.hostsim_render(hostsim, sc, cam, p)
    assert np.array_equal(a, b) and sta.rays == stb.rays and sta.nodes_visited < stb.nodes_visited
    assert sta.prims_tested < stb.prims_tested


@pytest.mark.parametrize("form", [0, 1], ids=["comparisons", "min3max3"])
@pytest.mark.parametrize("f32", [0, 1], ids=["f64", "f32"])
def test_fast_cube_test_is_the_six_rectangle_test(hostsim, f32, form):
    """rt_core.hpp box_t_fast — ONE exact quotient for the face the walk's f32 plane distances single out, where the ray is clear of the box's edges,
    of t_min and of the incumbent by a margin — against Cube::hit's six rectangle tests (box_t, hittable.rs:560-569,503-513) in the host build's
    arithmetic (IEEE, nothing contracted): the same (hit, t bit for bit, face) on 6 million generated cases per precision — origins far, near, ON a
    face plane (the ray that has just scattered off the cube), inside; rays aimed at corners, edges and face points displaced by 0 .. 1e-3 of the
    box; axis-parallel rays; ranges that end or start at, one ulp beside, or 1e-6 beside the exact t of a face; one box in eight with swapped or coincident
    corners on an axis (tests/hostsim box_fast_check_t).
    With the margin set to 0 the same cases give ~6 % mismatches: the test sees what it is for.  Both ways box_classify writes its verdicts
    (the strict build compiles one, the contracted builds the other) are held."""
    import ctypes as C
    lib = hostsim.lib
    lib.hostsim_box_fast_check.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_void_p]
    decided = hits = total = odd = 0
    for seed in (1, 2, 3):
        out = np.zeros(8, dtype=np.uint64)
        assert lib.hostsim_box_fast_check(2_000_000, seed, f32, form, out.ctypes.data) == 0
        assert out[0] == 0, "first mismatch at case %d of seed %d" % (int(out[5]), seed)
        decided += int(out[1] + out[2]); hits += int(out[4]); total += int(out[1] + out[2] + out[3]); odd += int(out[6])
    assert total == 6_000_000 and decided > 0.5 * total and hits > 0.3 * total   # (the generator is hostile: the fast path still decides most cases)
    # one case in eight is a cube from corners that are not min / max on an axis, or flat on one (the round-5 advisor's counter-example: mn.x = 1,
    # mx.x = 0 — the six rectangles still hit the faces across x, the unguarded classification said "miss"): such records are box_t's
    assert 0.10 * total < odd < 0.15 * total


def test_f64_box_test_passes_whatever_the_reference_test_passes(hostsim):
    """Property test of the f64 kernels' box test (rt_core.hpp slab_ray + slab_hit4: f32 arithmetic, error bounds folded into per-walk constants)
    against Bound::hit (bound.rs:13-32) evaluated in f64: on 1.2 million (box, ray, range) cases — boxes and origins from 1e-3 to 1e6, directions with
    tiny and zero components, open and closed ranges, and rays aimed AT corners, edges and faces so that entry and exit coincide — every box
    the reference test passes, the product's test passes too (it may pass more)."""
    import ctypes as C
    rng = np.random.default_rng(2026)
    n = 300_000
    lib = hostsim.lib
    lib.hostsim_slab4_f64.argtypes = [C.c_uint32] + [C.c_void_p] * 7
    passed_ref = passed_prod = 0
    for scale in (1.0, 1e3, 1e-3, 1e6):
        c = (rng.random((n, 3, 4)) * 2 - 1) * scale * 10
        ext = scale * 10.0 ** rng.uniform(-4, 0.5, (n, 3, 4))
        ext[rng.random((n, 3, 4)) < 0.1] = 0.0                                      # flat boxes
        lo = (c - ext).astype(np.float32); hi = (c + ext).astype(np.float32)
        lo, hi = np.minimum(lo, hi), np.maximum(lo, hi)
        o = (rng.random((n, 3)) * 2 - 1) * scale * 10 * 10.0 ** rng.integers(0, 3, (n, 1))
        d = rng.normal(size=(n, 3))
        tiny = rng.random((n, 3)) < 0.15
        d[tiny] *= 10.0 ** rng.uniform(-12, -4, tiny.sum())
        d[rng.random((n, 3)) < 0.05] = 0.0
        d[(d == 0).all(axis=1)] = (0.0, 0.0, 1.0)
        # half of the rays aimed at a point ON box 0 (a corner, an edge, a face point, by choosing lo / hi / between per axis): grazing hits
        aim = rng.random(n) < 0.5
        pick = rng.integers(0, 3, (n, 3))
        between = lo[:, :, 0].astype(np.float64) + rng.random((n, 3)) * (hi[:, :, 0].astype(np.float64) - lo[:, :, 0])
        target = np.where(pick == 0, lo[:, :, 0], np.where(pick == 1, hi[:, :, 0], between)).astype(np.float64)
        d[aim] = (target - o)[aim] * 10.0 ** rng.uniform(-2, 1, (aim.sum(), 1))
        tmin = np.where(rng.random(n) < 0.8, 1e-3 * scale, 0.0)
        tmax = np.where(rng.random(n) < 0.5, np.finfo(np.float64).max, 10.0 ** rng.uniform(-2, 3, n))
        out = np.zeros((n, 4), dtype=np.uint8)
        lo = np.ascontiguousarray(lo); hi = np.ascontiguousarray(hi); o = np.ascontiguousarray(o); d = np.ascontiguousarray(d)
        lib.hostsim_slab4_f64(n, lo.ctypes.data, hi.ctypes.data, o.ctypes.data, d.ctypes.data, tmin.ctypes.data, tmax.ctypes.data, out.ctypes.data)
        # Bound::hit in f64 (f64::max / min ignore a NaN operand, as np.fmax / np.fmin do)
        with np.errstate(all="ignore"):
            inv = 1.0 / d
            mn = np.repeat(tmin[:, None], 4, axis=1); mx = np.repeat(tmax[:, None], 4, axis=1)
            ok = np.ones((n, 4), dtype=bool)
            for a in range(3):
                t0 = (lo[:, a, :].astype(np.float64) - o[:, a, None]) * inv[:, a, None]
                t1 = (hi[:, a, :].astype(np.float64) - o[:, a, None]) * inv[:, a, None]
                neg = (inv[:, a] < 0)[:, None]
                t0, t1 = np.where(neg, t1, t0), np.where(neg, t0, t1)
                mn = np.fmax(t0, mn); mx = np.fmin(t1, mx)
                ok &= ~(mx < mn)
        missed = ok & (out == 0)
        assert not missed.any(), (scale, int(missed.sum()), np.argwhere(missed)[:3])
        passed_ref += int(ok.sum()); passed_prod += int(out.sum())
    assert passed_ref > 100_000 and passed_ref <= passed_prod <= passed_ref * 1.5, (passed_ref, passed_prod)   # (more: tiny and zero direction components cost slack; but not everything)


def test_quantised_records_on_hostile_geometry(hostsim, monkeypatch):
    """The quantised node records (bvh_quant.hpp) on boxes that stress the quantisation: flat rectangles (an axis of zero extent), a huge
    sphere beside tiny ones (one child spans the node, the others a cell of it), coordinates around 1e6 and around 1e-3, and a camera
    whose rays are exactly parallel to an axis (NaN plane distances).  Conservative boxes may only OPEN more nodes: same image bit for
    bit, same world.hit() calls, in both precisions."""
    from rttnw_amd import scene as S
    rng = np.random.default_rng(7)

    def world(sc, scale, shift):
        items = []
        grey = sc.lambertian((0.6, 0.6, 0.6))
        for k in range(300):
            c = shift + scale * (rng.random(3) * 20 - 10)
            items.append(sc.sphere(tuple(c), scale * float(10 ** rng.uniform(-3, 0)), grey if k % 3 else sc.metal((0.8, 0.7, 0.6), 0.1)))
        for k in range(60):
            a = shift + scale * (rng.random(3) * 20 - 10)
            e = scale * rng.random(2) * 3
            items.append(sc.rectangle(int(k % 3), (a[0], a[0] + e[0]), (a[1], a[1] + e[1]), a[2], grey))
            b = shift + scale * (rng.random(3) * 20 - 10)
            items.append(sc.cube(tuple(b), tuple(b + scale * (0.01 + rng.random(3))), grey))
        items.append(sc.sphere(tuple(shift + np.array([0, -1000 * scale, 0])), 990 * scale, grey))      # the ground: as large as the node
        items.append(sc.rectangle(abi.XZ, (shift[0] - 5 * scale, shift[0] + 5 * scale), (shift[2] - 5 * scale, shift[2] + 5 * scale), shift[1] + 15 * scale,
                                  sc.diffuse_light((5, 5, 5))))
        return sc.list([sc.bvh_tree(sc.list(items))])

    for scale, shift in ((1.0, np.zeros(3)), (1e-3, np.array([1e-3, 2e-3, -1e-3])), (1.0, np.array([1e6, -2e6, 5e5]))):
        sc = S.Scene(hostsim, 3)
        sc.set_world(world(sc, scale, shift))
        sc.commit()
        for lookfrom in (shift + scale * np.array([0.0, 2.0, -40.0]), shift + scale * np.array([0.0, 0.0, -40.0])):   # the second: the centre ray is exactly (0, 0, 1)
            cam = S.camera_desc(tuple(lookfrom), tuple(shift + scale * np.array([0.0, (lookfrom - shift)[1] / scale, 0.0])), 35.0, 1.0)
            for prec in (abi.F64, abi.F32):
                p = S.make_params(33, 33, 3, background=(0.2, 0.3, 0.5), precision=prec, seed=4, collect_counters=1, t_min=1e-3 * scale)
                monkeypatch.delenv("HOSTSIM_QUANT", raising=False)
                a, sta = util.hostsim_render(hostsim, sc, cam, p)
                for mode in ("1",):   # 8-bit quantised records
                    monkeypatch.setenv("HOSTSIM_QUANT", mode)
                    b, stb = util.hostsim_render(hostsim, sc, cam, p)
                    assert np.array_equal(a, b) and sta.rays == stb.rays and np.isfinite(a).all(), (scale, prec, mode)
                if prec == abi.F64:   # ... and a walk that never culls (every used slot's box opened to the whole space) finds the same hits: the f64 kernels' box tests are conservative
                    monkeypatch.setenv("HOSTSIM_QUANT", "3")
                    b, stb = util.hostsim_render(hostsim, sc, cam, p)
                    assert np.array_equal(a, b) and sta.rays == stb.rays and stb.nodes_visited > sta.nodes_visited, (scale, "never culls")
                assert a.max() > 0   # (far from the origin the quantised step visits FEWER nodes in f64: it subtracts the origin in double, the f32-record test pays a slack of 2.4e-7 |o / d|)


@pytest.mark.parametrize("precision", [abi.F64, abi.F32], ids=["f64", "f32"])
def test_rays_nothing_can_cull_stay_inside_the_stack_bound(hostsim, scenes_lib, earth, precision):
    """A camera with lookfrom == lookat (w = unit(0): every ray NaN — the reference renders NaN pixels, main.rs:219-225 writes 0) and one whose
    view_up is parallel to the view: no plane distance of such a ray is a number, so even the inverted box of an unused node slot 'passes' (the
    node steps leave the slot's own test to its box, rt_core.hpp RT_NODE_EMPTY_CHECK).  The walk must not start (rt_core.hpp
    slab_ray_can_be_culled): its stack stays inside the bound the lowering sized the device's LDS stacks by, and the pixels are NaN (or black)."""
    for name in ("cornell_box", "final_scene"):
        sc, setup = util.build(hostsim, scenes_lib, name, earth)
        dims = (C.c_uint32 * 8)()
        hostsim.lib.hostsim_scene_dims(sc.handle, dims)
        for degenerate in ("lookfrom_is_lookat", "view_up_along_the_view"):
            cam, p = util.params_for(setup, 16, 16, 4, spp_chunk=4, precision=precision)
            if degenerate == "lookfrom_is_lookat":
                for k in range(3):
                    cam.lookat[k] = cam.lookfrom[k]
            else:
                for k in range(3):
                    cam.view_up[k] = cam.lookat[k] - cam.lookfrom[k]
            hostsim.lib.hostsim_max_stack()  # (reads and resets the high-water mark)
            lin, _ = util.hostsim_render(hostsim, sc, cam, p)
            assert hostsim.lib.hostsim_max_stack() <= dims[7], (name, degenerate)
            # (final_scene: the fog's boundary test accepts NaN roots as Sphere::hit does, hittable.rs:100-107 — such a path bounces inside the
            # medium to the depth limit and returns black, in the reference as here)
            assert (np.isnan(lin) | (lin == 0)).all() and (name != "cornell_box" or np.isnan(lin).all()), (name, degenerate)


def test_l2_model_of_the_decoupled_kernel_runs_and_adds_up(hostsim, scenes_lib):
    """tests/hostsim/cache_model.hpp — the host model config 5's layouts were priced on (one XCD's waves of the decoupled kernel, the product's own walk
    steps, against an LRU cache) — on a small cloud with a small cache: every stream's misses are at most its accesses, node visits and sphere tests
    per ray are those of the walk, a layout that puts every node record on a line of its own costs lines, and a bigger cache misses less."""
    import ctypes as C
    sc, setup = util.build(hostsim, scenes_lib, "spheres_1m", None, 20000)
    cam, p = util.params_for(setup, 256, 256, 16, precision=abi.F64, seed=1)
    dims = (C.c_uint32 * 8)()
    hostsim.lib.hostsim_scene_dims(sc.handle, dims)
    n4 = dims[0]

    def run(cache_bytes, node_perm=None, node_bytes=64):
        prm = np.zeros(20, dtype=np.uint32)
        prm[:] = [8, 16, 5, 8, 2, 2, cache_bytes, 0, 32, 40, 0, 8, 0, 5, 0, node_bytes, 0, 0, 0, 5]
        out = np.zeros(192, dtype=np.uint64)
        hostsim.lib.hostsim_cache_model(sc.handle, C.byref(cam), C.byref(p), prm.ctypes.data_as(C.c_void_p),
                                        node_perm.ctypes.data_as(C.c_void_p) if node_perm is not None else None, None, out.ctypes.data_as(C.c_void_p))
        return out

    small = run(64 << 10)
    rays, nodes, tests = int(small[1]), int(small[2]), int(small[3])
    assert rays > 1000 and 5 * rays < nodes < 60 * rays and 0 < tests < 10 * rays
    for s in range(7):
        acc, miss = int(small[8 + 4 * s]), int(small[9 + 4 * s])
        assert miss <= acc
    assert int(small[8]) == nodes and sum(int(small[64 + d]) for d in range(32)) == nodes      # node accesses = node visits, by depth too
    big = run(8 << 20)
    assert int(big[9]) < int(small[9])                                                         # a cache that holds the tree misses less
    spread = run(64 << 10, np.arange(n4, dtype=np.uint32) * 2)                                 # every record on a line of its own
    assert int(spread[9]) > int(small[9])
    assert int(small[128]) > 0 and int(small[129]) <= 64 * int(small[128]) and int(small[132]) > 0   # step executions and the lanes they served
