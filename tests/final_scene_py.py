"""final_scene (reference src/scenes.rs:238-334) written against the Python mirror of the C ABI, with KNOBS.

Test infrastructure: the catalogue's own final_scene lives in rttnw_amd/host/scenes.cpp (fixed constants, like the
reference).  This restatement draws the same scene stream in the same order — with the defaults it is the same scene,
object for object (tests/test_oracle_png_pins.py::test_python_final_scene_is_the_catalogue_scene) — and lets a test
change one constant at a time to show that a pin against image.png would notice.
"""
import ctypes as C

import numpy as np

from rttnw_amd import abi
from rttnw_amd import scene as S

DEFAULTS = dict(fog_density=0.0001, blue_density=0.2, metal_fuzz=1.0, moving_dx=30.0, light=7.0, n_cluster=1000)


def scene_stream(scenes_lib, seed, n):
    out = np.zeros(n, dtype=np.float64)
    scenes_lib.scenes_rng_f64(seed, 1, n, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def build(binding, scenes_lib, earth, seed=0x5EED0001, **knobs):
    k = dict(DEFAULTS, **knobs)
    ns = k["n_cluster"]
    draws = iter(scene_stream(scenes_lib, seed, 400 + 3 * ns))
    rng = lambda a, b: a + (b - a) * next(draws)                                   # noqa: E731  gen_range(a..b)
    sc = S.Scene(binding, seed, scenes_binding=scenes_lib)
    boxes = sc.list()
    ground = sc.lambertian((0.48, 0.83, 0.53))
    for i in range(20):                                                            # scenes.rs:244-253
        for j in range(20):
            x0, z0 = -1000.0 + i * 100.0, -1000.0 + j * 100.0
            sc.push(boxes, sc.cube((x0, 0.0, z0), (x0 + 100.0, rng(1.0, 101.0), z0 + 100.0), ground))
    world = sc.list()
    sc.push(world, sc.bvh_tree(boxes))
    sc.push(world, sc.rectangle(abi.XZ, (123.0, 423.0), (147.0, 412.0), 554.0, sc.diffuse_light((k["light"],) * 3)))
    sc.push(world, sc.moving_sphere((400.0, 400.0, 400.0), (400.0 + k["moving_dx"], 400.0, 400.0), 0.0, 1.0, 50.0,
                                    sc.lambertian((0.7, 0.3, 0.1))))
    sc.push(world, sc.sphere((260.0, 150.0, 45.0), 50.0, sc.dielectric(1.5)))
    sc.push(world, sc.sphere((0.0, 150.0, 45.0), 50.0, sc.metal((0.8, 0.8, 0.9), k["metal_fuzz"])))
    boundary = sc.sphere((360.0, 150.0, 145.0), 70.0, sc.dielectric(1.5))
    sc.push(world, boundary)
    sc.push(world, sc.constant_medium(boundary, k["blue_density"], (0.2, 0.4, 0.9)))
    sc.push(world, sc.constant_medium(sc.sphere((0.0, 0.0, 0.0), 5000.0, sc.dielectric(1.5)), k["fog_density"], (1.0, 1.0, 1.0)))
    sc.push(world, sc.sphere((400.0, 200.0, 400.0), 100.0, sc.lambertian(sc.image(earth))))
    sc.push(world, sc.sphere((220.0, 280.0, 300.0), 80.0, sc.lambertian(sc.noise(0.1))))
    cluster = sc.list()
    white = sc.lambertian((0.73, 0.73, 0.73))
    for _ in range(ns):                                                            # scenes.rs:319-325
        sc.push(cluster, sc.sphere((rng(0.0, 165.0), rng(0.0, 165.0), rng(0.0, 165.0)), 10.0, white))
    sc.push(world, sc.translate(sc.rotate_y(sc.bvh_tree(cluster), 15.0), (-100.0, 270.0, 395.0)))
    sc.set_world(world)
    sc.commit()
    return sc
