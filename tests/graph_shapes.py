"""Scene graphs the reference's trait objects allow (any Hittable can be wrapped by `.translate()` / `.rotate_y()`, put in a
List / BvhTree, or handed to ConstantMedium as its boundary — hittable.rs:51-65,731) beyond the shapes its nine scenes
use.  Built through the Python mirror of the C ABI for any binding; tests compare product and oracle on them."""
from rttnw_amd import abi
from rttnw_amd import scene as S


def _room(sc):
    """A lit open box so that everything inside is visible: floor, back wall, area light."""
    white = sc.lambertian((0.73, 0.73, 0.73))
    world = sc.list()
    sc.push(world, sc.rectangle(abi.XZ, (-300, 300), (-300, 300), 0.0, white))
    sc.push(world, sc.rectangle(abi.XY, (-300, 300), (0, 400), 300.0, sc.lambertian((0.2, 0.5, 0.7))))
    sc.push(world, sc.rectangle(abi.XZ, (-120, 120), (-120, 120), 399.0, sc.diffuse_light((9.0, 9.0, 9.0))))
    return world, white


def nested_transforms(sc):
    """An instance inside an instance, five wrappers deep on one object, a BvhTree inside both."""
    world, white = _room(sc)
    red = sc.lambertian((0.7, 0.2, 0.15))
    inner = sc.translate(sc.rotate_y(sc.cube((0, 0, 0), (60, 90, 60), red), 25.0), (40, 0, 30))   # an instance ...
    ball = sc.sphere((-50, 40, 10), 40.0, sc.metal((0.8, 0.8, 0.6), 0.2))
    cluster = sc.bvh_tree(sc.list([sc.sphere((10 + 22 * k, 130, -20 + 9 * k), 12.0, white) for k in range(6)]))
    cluster = sc.translate(cluster, (-60, 0, 0))                                                    # ... and another
    group = sc.list([inner, ball, cluster])
    group = sc.translate(sc.rotate_y(group, -20.0), (-30, 0, 60))                                   # ... inside an instance
    deep = sc.cube((-15, 0, -15), (15, 50, 15), sc.lambertian((0.2, 0.7, 0.3)))
    for k in range(5):                                                                              # five wrappers
        deep = sc.rotate_y(deep, 8.0 + k) if k % 2 else sc.translate(deep, (25.0, 2.0 * k, -12.0))
    # ... and a sphere under five wrappers: the lowering tests it in world space and makes its record through the whole chain
    deep_ball = sc.sphere((5, 25, -8), 22.0, sc.metal((0.7, 0.6, 0.5), 0.0))
    for k in range(5):
        deep_ball = sc.rotate_y(deep_ball, -11.0 - 3 * k) if k % 2 == 0 else sc.translate(deep_ball, (-18.0, 3.0 * k, 14.0))
    sc.push(world, group)
    sc.push(world, deep)
    sc.push(world, deep_ball)
    return world


def medium_in_group(sc):
    """A constant medium INSIDE a rotated + translated group (its hit record goes through the group's wrappers), its
    boundary wrapped once more; and a world-level medium whose boundary is a wrapped cube."""
    world, white = _room(sc)
    smoke = sc.constant_medium(sc.translate(sc.sphere((0, 60, 0), 55.0, white), (10, 0, -5)), 0.03, (0.9, 0.9, 0.9))
    post = sc.cube((60, 0, -20), (90, 140, 10), sc.lambertian((0.6, 0.3, 0.1)))
    group = sc.translate(sc.rotate_y(sc.list([post, smoke]), 30.0), (-40, 0, 40))
    sc.push(world, group)
    box = sc.translate(sc.rotate_y(sc.cube((0, 0, 0), (70, 70, 70), white), -15.0), (70, 0, -60))
    sc.push(world, sc.constant_medium(box, 0.02, (0.1, 0.1, 0.1)))
    return world


def list_boundaries(sc):
    """ConstantMedium over a List of a sphere and a cube (treated as convex, hittable.rs:739) and over a BvhTree."""
    world, white = _room(sc)
    both = sc.list([sc.sphere((-60, 70, 0), 50.0, white), sc.cube((-70, 20, -30), (20, 90, 30), white)])
    sc.push(world, sc.constant_medium(both, 0.04, (0.8, 0.3, 0.2)))
    tree = sc.bvh_tree(sc.list([sc.sphere((80 + 30 * k, 60, 20 * k), 35.0, white) for k in range(3)]))
    sc.push(world, sc.constant_medium(sc.translate(tree, (0, 10, -40)), 0.05, (0.2, 0.3, 0.9)))
    return world


def wide_shutter(sc):
    """Moving spheres seen through a shutter that is open from -0.5 to 1.7 (BvhTree::from_time, hittable.rs:261)."""
    world, white = _room(sc)
    for k in range(5):
        sc.push(world, sc.moving_sphere((-150 + 60 * k, 60, 0), (-150 + 60 * k + 40, 60 + 30, -20), 0.0, 1.0, 25.0,
                                        sc.lambertian((0.7, 0.3 + 0.1 * k, 0.1))))
    return world


def many_moved_spheres(sc):
    """600 spheres, each under its own rotate_y + translate: more transform chains than the 512 a world-space sphere copy can
    name (rt_types.hpp MAT_HOME_INST_MAX) — the first 512 are tested in world space, the others stay instances."""
    world, white = _room(sc)
    mats = [sc.lambertian((0.8, 0.3, 0.2)), sc.metal((0.7, 0.7, 0.8), 0.1), white]
    for k in range(600):
        i, j = k % 30, k // 30
        ball = sc.sphere((0.0, 0.0, 0.0), 4.5, mats[k % 3])
        sc.push(world, sc.translate(sc.rotate_y(ball, 7.0 * k), (-145.0 + 10.0 * i, 20.0 + 9.0 * j, -60.0 + 4.0 * ((i + j) % 7))))
    return world


SHAPES = {"nested_transforms": nested_transforms, "medium_in_group": medium_in_group, "list_boundaries": list_boundaries,
          "wide_shutter": wide_shutter, "many_moved_spheres": many_moved_spheres}


def build(binding, shape, w=56, h=40, spp=6, precision=abi.F64, seed=13):
    sc = S.Scene(binding, 7)
    sc.set_world(SHAPES[shape](sc))
    sc.commit()
    shutter = (-0.5, 1.7) if shape == "wide_shutter" else (0.0, 1.0)
    cam = S.camera_desc((0.0, 160.0, -520.0), (0.0, 120.0, 0.0), 40.0, w / h, open_time=shutter[0], close_time=shutter[1])
    p = S.make_params(w, h, spp, precision=precision, seed=seed, spp_chunk=3)
    return sc, cam, p
