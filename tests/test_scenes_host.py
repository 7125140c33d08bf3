"""The host-side scenes.rs mirror (rttnw_amd/host/scenes.cpp) and the camera/size table of main.rs:66-183."""
import ctypes as C

import numpy as np
import pytest

import util
from golden_cases import CASES, load
from oracle import rto
from rttnw_amd import abi
from rttnw_amd import scene as S

TABLE = {  # main.rs:66-183 + defaults main.rs:255,184
    "random_scene": (1, (0.7, 0.8, 1.0), (13, 2, 3), (0, 0, 0), 20, 0.1, 400, 225, 100),
    "two_spheres": (2, (0.7, 0.8, 1.0), (13, 2, 3), (0, 0, 0), 20, 0.0, 400, 225, 100),
    "two_perlin_spheres": (3, (0.7, 0.8, 1.0), (13, 2, 3), (0, 0, 0), 20, 0.0, 400, 225, 100),
    "earth": (4, (0.7, 0.8, 1.0), (13, 2, 3), (0, 0, 0), 20, 0.0, 400, 225, 100),
    "simple_light": (5, (0, 0, 0), (26, 3, 6), (0, 2, 0), 20, 0.0, 400, 225, 400),
    "empty_cornell_box": (6, (0, 0, 0), (278, 278, -800), (278, 278, 0), 40, 0.0, 600, 600, 200),
    "cornell_box": (7, (0, 0, 0), (278, 278, -800), (278, 278, 0), 40, 0.0, 600, 600, 200),
    "smoke_cornell_box": (8, (0, 0, 0), (278, 278, -800), (278, 278, 0), 40, 0.0, 600, 600, 200),
    "final_scene": (9, (0, 0, 0), (478, 278, -600), (278, 278, 0), 40, 0.0, 800, 800, 10000),
}


@pytest.mark.parametrize("name", sorted(TABLE))
def test_scene_table(oracle, scenes_lib, earth, name):
    num, bg, frm, at, fov, ap, w, h, spp = TABLE[name]
    sc, st = util.build(oracle, scenes_lib, name, earth)
    assert st.scene_number == num and scenes_lib.scenes_name(num).decode() == name
    assert tuple(st.background) == bg and tuple(st.camera.lookfrom) == frm and tuple(st.camera.lookat) == at
    assert st.camera.vertical_fov == fov and st.camera.aperture == ap
    assert (st.width, st.height, st.spp) == (w, h, spp)
    assert tuple(st.camera.view_up) == (0, 1, 0) and st.camera.focus_distance == 10
    assert (st.camera.open_time, st.camera.close_time) == (0, 1)
    assert st.camera.aspect_ratio == w / h


def test_unknown_scene_is_an_error(oracle, scenes_lib):
    sc = S.Scene(oracle, scenes_binding=scenes_lib)
    with pytest.raises(abi.RttnwError):
        sc.build_named("no_such_scene")                                          # main.rs:179-182
    assert scenes_lib.scenes_name(0) is None and scenes_lib.scenes_name(10) is None


def test_scene_rng_is_the_documented_stream(oracle, scenes_lib):
    a = np.zeros(16)
    b = np.zeros(16)
    scenes_lib.scenes_rng_f64(0x5EED0001, 1, 16, a.ctypes.data_as(C.POINTER(C.c_double)))
    oracle.probe_scene_rng(0x5EED0001, 1, 16, b.ctypes.data_as(C.POINTER(C.c_double)))
    assert (a == b).all() and a.min() >= 0 and a.max() < 1
    scenes_lib.scenes_rng_f64(0x5EED0001, 2, 16, b.ctypes.data_as(C.POINTER(C.c_double)))
    assert not (a == b).any()


def test_final_scene_structure(oracle, scenes_lib, earth):
    """scenes.rs:238-334: 11 world items; floor heights in [1,101); cluster under rotate 15 + translate."""
    sc, _ = util.build(oracle, scenes_lib, "final_scene", earth)
    box, bp = rto.darr(*([0.0] * 6))
    assert oracle.probe_bbox(sc.handle, -1, 0.0, 1.0, bp) == 1
    assert box[0] <= -5000 and box[3] >= 5000                                    # the fog boundary dominates
    # the ground: a ray straight down far from everything hits a box top with 1 <= y < 101
    _, ray = rto.darr(-950.0, 500.0, -950.0, 0.0, -1.0, 0.0, 0.0)
    rec, rp = rto.darr(*([0.0] * 11))
    # bounce 49 keyed draws for the media are irrelevant to a 500-unit ray at density 1e-4 most of the time;
    # probe the BVH of boxes through the world and accept either the ground or a (rare) fog event
    r = oracle.probe_hit(sc.handle, -1, ray, 0.001, 1e30, 1, 0, 0, 0, 1, rp)
    assert r == 1 and (1.0 <= rec[2] < 101.0 or rec[10] >= 0)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_reproduces_goldens(oracle, scenes_lib, earth, case):
    """The committed golden vectors are reproducible outputs of the oracle (bit-exact, f64)."""
    key, name, w, h, spp, chunk, param = case
    sc, setup = util.build(oracle, scenes_lib, name, earth, param)
    cam, p = util.params_for(setup, w, h, spp, spp_chunk=chunk)
    lin, rgba, _ = rto.render(sc, cam, p)
    g = load()
    assert np.array_equal(lin, g[key + "_linear"]) and np.array_equal(rgba, g[key + "_rgba8"])


@pytest.mark.parametrize("name,param", [("cornell_box", 0), ("final_scene", 0), ("random_scene", 0), ("smoke_cornell_box", 0),
                                        ("spheres_1m", 5000), ("two_spheres", 0)])
def test_wide_nodes_are_the_collapsed_binary_tree(hostsim, scenes_lib, earth, name, param):
    """The kernels walk 4-wide records made by collapsing the builder's binary tree (scene_lower.cpp collapse4): the top-level
    wide tree holds exactly the leaves of the top-level binary tree, boxes nest, every record of every tree is reached
    from the top root or an instance's root, and the traversal-stack bound the lowering reports covers the deepest walk."""
    sc, _ = util.build(hostsim, scenes_lib, name, earth, param)
    n2, root2 = util.nodes_of(hostsim, sc)
    n4, root4 = util.nodes_of(hostsim, sc, wide=True)
    leaves4, need, seen = util.check_wide_tree(n4, root4)
    assert leaves4 == util.leaves_of_binary(n2, root2)
    assert len(n4) <= max(1, (len(n2) + 1) // 2 + 2) or len(n2) < 8          # about half as many records (a third when the tree is full)
    dims = (C.c_uint32 * 8)()
    hostsim.lib.hostsim_scene_dims(sc.handle, dims)
    assert need + 1 <= dims[7]   # (+ the instances' trees, held against real walks by test_core_f64_equals_golden)


def test_parallel_host_builder_is_deterministic_and_complete(hostsim, scenes_lib):
    """A big flat list goes through the host builder's parallel paths (scene_lower.cpp: parallel collection, SAH topology by
    splits whose halves run on two threads from 32 768 items up, record emission in chunks): every sphere must sit in exactly
    one leaf, child boxes must nest, and two builds must be the same tree bit for bit — nothing may depend on thread timing."""
    n_spheres = 70000
    trees = []
    for _ in range(2):
        sc, _ = util.build(hostsim, scenes_lib, "spheres_1m", None, n_spheres)
        n2, root2 = util.nodes_of(hostsim, sc)
        n4, root4 = util.nodes_of(hostsim, sc, wide=True)
        trees.append((n2.tobytes(), n4.tobytes(), root2, root4))
    assert trees[0] == trees[1]
    leaves = util.leaves_of_binary(n2, root2)                    # leaf codes: ~(kind << 28 | (records - 1) << 26 | first record)
    records = []
    for code in leaves:
        bits = ~code & 0xFFFFFFFF
        records += [(bits >> 28, (bits & 0x3FFFFFF) + k) for k in range(((bits >> 26) & 3) + 1)]
    assert len(records) == len(set(records)) == n_spheres + 1    # every record in exactly one leaf (+ the area light)
    assert sorted(r for k, r in records if k == 0) == list(range(n_spheres))   # PRIM_SPHERE = 0: records 0 .. n-1, emitted in leaf order
    leaves4, need, seen = util.check_wide_tree(n4, root4)        # boxes nest, every 4-wide record reached once
    assert leaves4 == leaves and len(seen) == len(n4)
