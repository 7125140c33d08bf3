"""`Hittable::bounding_box` on the boundary: rttnw_hittable_bounds (include/rttnw_hip.h) against the CPU oracle's restatement of
hittable.rs:125-130,165-176,233-244,370-372,532-546,585-591,619-628,798-800 (rto_probe_bbox) on the SAME graph built through both bindings.
Runs on the host build of the C-ABI's host half (tests/hostsim links rttnw_amd/csrc/capi_builder.cpp: the function only reads the scene graph),
so no GPU is needed; tests/test_abi_exports.py holds the product library to exporting it.
Every kind but YRotate: bit-identical to the oracle.  YRotate: the CORRECT rotation of the item's eight corners — the reference's box (quirk Q2,
hittable.rs:661-662: the z line reads the overwritten x) can fail to contain the object, which this test also shows."""
import numpy as np
import pytest

from oracle import rto
from rttnw_amd import abi
from rttnw_amd import scene as S


def oracle_box(oracle, sc, hid, t0=0.0, t1=1.0):
    out, p = rto.darr(*([0.0] * 6))
    rc = oracle.probe_bbox(sc.handle, hid, t0, t1, p)
    return (np.array(out[:3]), np.array(out[3:])) if rc == 1 else None


def build(sc):
    """The same little graph through any binding (ids need not agree: constant_medium makes its Isotropic as an object of its own in the product)."""
    m = sc.lambertian((0.5, 0.5, 0.5))
    ids = {}
    ids["sphere"] = sc.sphere((1.5, -2.25, 3.0), 0.75, m)
    ids["neg_radius"] = sc.sphere((0.0, 1.0, 0.0), -0.45, m)           # the hollow-glass trick of the book: min > max, literally
    ids["moving"] = sc.moving_sphere((400, 400, 200), (430, 410, 200), 0.25, 1.5, 50.0, m)
    ids["rect_xy"] = sc.rectangle(abi.XY, (3, 5), (1, 3), -2.0, m)
    ids["rect_xz"] = sc.rectangle(abi.XZ, (213, 343), (227, 332), 554.0, m)
    ids["rect_yz"] = sc.rectangle(abi.YZ, (0, 555), (0, 555), 555.0, m)
    ids["cube"] = sc.cube((0, 0, 0), (165, 330, 165), m)
    ids["cube_swapped"] = sc.cube((10, 0, 5), (0, 3, 7), m)            # corners as given (Cube::new takes any two points)
    ids["moved"] = sc.translate(ids["cube"], (265, 0, 295))
    ids["list"] = sc.list([ids["sphere"], ids["moving"], ids["rect_xz"]])
    ids["empty"] = sc.list()
    ids["list_with_empty"] = sc.list([ids["sphere"], ids["empty"]])
    cloud = sc.list([sc.sphere((10.0 * k, 3.0 * (k % 4), -7.0 * k), 1.0 + 0.1 * k, m) for k in range(9)] + [ids["moving"]])
    ids["bvh"] = sc.bvh_tree(cloud)
    ids["medium"] = sc.constant_medium(ids["moved"], 0.01, (1.0, 1.0, 1.0))
    ids["moved_bvh"] = sc.translate(ids["bvh"], (-100, 200, -100))
    return ids


@pytest.fixture()
def both(hostsim, oracle):
    a, b = S.Scene(hostsim), S.Scene(oracle)
    return a, b, build(a), build(b)


def test_bounds_equal_the_reference_restatement(hostsim, oracle, both):
    a, b, ids, ids_o = both
    for name, hid in ids.items():
        for (t0, t1) in ((0.0, 1.0), (0.25, 0.75), (1.0, 1.0)):
            got, want = a.bounding_box(hid, t0, t1), oracle_box(oracle, b, ids_o[name], t0, t1)
            assert (got is None) == (want is None), (name, got, want)
            if want is not None:
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (name, t0, t1, got, want)
    assert a.bounding_box(ids["empty"]) is None and a.bounding_box(ids["list_with_empty"]) is None    # hittable.rs:165-176
    mn, mx = a.bounding_box(ids["neg_radius"])
    assert (mn > mx).all()                                                                               # centre -+ radius, literally
    mn, mx = a.bounding_box(ids["moved"])
    assert np.array_equal(mn, [265, 0, 295]) and np.array_equal(mx, [430, 330, 460])                   # the KAT of tests/test_oracle_kat.py
    mn, mx = a.bounding_box(ids["rect_xz"])
    assert np.array_equal(mn, [213, 554 - 0.0001, 227]) and np.array_equal(mx, [343, 554 + 0.0001, 332])
    # a BvhTree answers with the bound it stored at construction (times 0..1), whatever is asked — a List asks its members again
    assert np.array_equal(a.bounding_box(ids["bvh"], 0.0, 1.0)[1], a.bounding_box(ids["bvh"], 5.0, 9.0)[1])
    assert not np.array_equal(a.bounding_box(ids["list"], 0.0, 1.0)[1], a.bounding_box(ids["list"], 5.0, 9.0)[1])


def test_rotated_bounds_contain_the_object_and_the_references_do_not(hostsim, oracle, both):
    a, b, ids, ids_o = both
    for deg in (15.0, -18.0, 45.0, 90.0, 200.0):
        ra, rb = a.rotate_y(ids["cube"], deg), b.rotate_y(ids_o["cube"], deg)
        got = a.bounding_box(ra)
        th = np.radians(deg)
        s, c = np.sin(th), np.cos(th)
        corners = np.array([[x, y, z] for x in (0, 165) for y in (0, 330) for z in (0, 165)], dtype=np.float64)
        world = np.stack([c * corners[:, 0] + s * corners[:, 2], corners[:, 1], -s * corners[:, 0] + c * corners[:, 2]], axis=1)
        assert np.allclose(got[0], world.min(axis=0), rtol=0, atol=1e-12 * 330) and np.allclose(got[1], world.max(axis=0), rtol=0, atol=1e-12 * 330)
        # the object's corners, taken through the wrapper's own transform, lie inside the library's box ...
        assert (world >= got[0] - 1e-9).all() and (world <= got[1] + 1e-9).all()
        # ... and stored once: the times asked do not matter (hittable.rs:719-721)
        assert np.array_equal(a.bounding_box(ra, 3.0, 4.0)[0], got[0])
        # the reference's own box (quirk Q2) is a different one, and for these angles it does not contain the rotated cube
        ref = oracle_box(oracle, b, rb)
        if deg in (15.0, 45.0, 200.0):
            assert not ((world >= ref[0] - 1e-9).all() and (world <= ref[1] + 1e-9).all()), deg
    # wrappers compose: translate(rotate_y(cube)) = the rotated box moved (hittable.rs:619-628 over :719-721)
    r = a.rotate_y(ids["cube"], -18.0)
    t = a.translate(r, (130, 0, 65))
    rb_, tb_ = a.bounding_box(r), a.bounding_box(t)
    assert np.array_equal(tb_[0], rb_[0] + [130, 0, 65]) and np.array_equal(tb_[1], rb_[1] + [130, 0, 65])


def test_bad_ids_are_errors(hostsim):
    sc = S.Scene(hostsim)
    tex = sc.solid(0.5)
    mat = sc.lambertian(tex)
    for bad in (tex, mat, 99, -1):
        with pytest.raises(Exception):
            sc.bounding_box(bad)
