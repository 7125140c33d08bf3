"""The catalogue scenes against numbers typed from the reference's src/scenes.rs — not from rttnw_amd/host/scenes.cpp.

Product and oracle are both driven by scenes.cpp, so a wrong constant there passes every product-vs-oracle test.  Here
the catalogue drives a RECORDING implementation of the builder table (tests/recording_builder.py) and the recorded calls
are compared, object for object, with the structures below: each is the reference's scene function transcribed as data
(file:line cited per scene).  cornell_box and final_scene are additionally pinned through rendered pixels
(tests/test_oracle_png_pins.py); final_scene's object list has its own Python restatement (tests/final_scene_py.py)."""
import math

import numpy as np
import pytest

import recording_builder as rb
from rttnw_amd import abi

XY, XZ, YZ = abi.XY, abi.XZ, abi.YZ


def solid(r, g=None, b=None):
    return ("solid", (r, r, r) if g is None else (r, g, b))


def lamb(tex):
    return ("lambertian", tex)


def rect(plane, a, b, k, mat):
    return ("rectangle", plane, a, b, k, mat)


CHECKER = ("checker", solid(0.2, 0.3, 0.1), solid(0.9, 0.9, 0.9))       # scenes.rs:14-17, :92-95 (odd, even)
RED, WHITE, GREEN = lamb(solid(0.65, 0.05, 0.05)), lamb(solid(0.73)), lamb(solid(0.12, 0.45, 0.15))   # scenes.rs:160-162


def cornell_walls(light, lx, lz):
    """scenes.rs:165-170 (and :207-212): green YZ at x = 555, red YZ at x = 0, the light, ceiling, floor, back wall."""
    return [rect(YZ, (0., 555.), (0., 555.), 555., GREEN), rect(YZ, (0., 555.), (0., 555.), 0., RED),
            rect(XZ, lx, lz, 554., ("diffuse_light", solid(light))),
            rect(XZ, (0., 555.), (0., 555.), 555., WHITE), rect(XZ, (0., 555.), (0., 555.), 0., WHITE),
            rect(XY, (0., 555.), (0., 555.), 555., WHITE)]


def block(size, angle, offset):   # Cube::new(0, size, white).rotate_y(angle).translate(offset) — scenes.rs:181-193
    return ("translate", ("rotate_y", ("cube", (0., 0., 0.), size, WHITE), angle), offset)


TALL, SHORT = block((165., 330., 165.), 15., (265., 0., 295.)), block((165., 165., 165.), -18., (130., 0., 65.))

EXPECTED = {
    # scenes.rs:90-108
    "two_spheres": ("list", [("sphere", (0., -10., 0.), 10., lamb(CHECKER)), ("sphere", (0., 10., 0.), 10., lamb(CHECKER))]),
    # scenes.rs:110-125
    "two_perlin_spheres": ("list", [("sphere", (0., -1000., 0.), 1000., lamb(("noise", 4.))),
                                    ("sphere", (0., 2., 0.), 2., lamb(("noise", 4.)))]),
    # scenes.rs:127-136
    "earth": ("list", [("sphere", (0., 0., 0.), 2., lamb(("image", True, 1200, 600)))]),
    # scenes.rs:138-155
    "simple_light": ("list", [("sphere", (0., -1000., 0.), 1000., lamb(("noise", 4.))), ("sphere", (0., 2., 0.), 2., lamb(("noise", 4.))),
                              rect(XY, (3., 5.), (1., 3.), -2., ("diffuse_light", solid(4.)))]),
    # scenes.rs:157-173
    "empty_cornell_box": ("list", cornell_walls(15., (213., 343.), (227., 332.))),
    # scenes.rs:175-196
    "cornell_box": ("list", cornell_walls(15., (213., 343.), (227., 332.)) + [TALL, SHORT]),
    # scenes.rs:198-236: light 7 on 113..443 x 127..432; the blocks as media, density 0.01, black and white
    "smoke_cornell_box": ("list", cornell_walls(7., (113., 443.), (127., 432.)) +
                          [("constant_medium", TALL, 0.01, solid(0.)), ("constant_medium", SHORT, 0.01, solid(1.))]),
}


@pytest.mark.parametrize("name", sorted(EXPECTED))
def test_fixed_scenes_are_the_reference_scenes(scenes_lib, earth, name):
    world, setup, rec = rb.record(scenes_lib, name, earth)
    assert world == EXPECTED[name]
    if name in ("two_spheres", "two_perlin_spheres", "simple_light"):
        # ONE texture object shared by both spheres (`Arc::clone`, scenes.rs:99-105,112-122,140-150): for NoiseTexture that is
        # one Perlin table, not two
        tex_kind = "checker" if name == "two_spheres" else "noise"
        assert sum(1 for o in rec.objs if o[0] == tex_kind) == 1


def test_random_scene_follows_the_reference_generator(scenes_lib):
    """scenes.rs:11-88: the ground, then for a, b in -11..11 a small sphere at (a + 0.9 + U, 0.2, b + 0.9 + U) unless within
    0.9 of (4, 0.2, 0) — 80 % moving Lambertian (albedo U*U per channel, end centre + (0, U[0, 0.5), 0), time 0..1), 15 % Metal
    (albedo 0.5 (1 - U), fuzz 0.5 U), 5 % Dielectric 1.5 — then the three big spheres.  The draws come from the scene
    stream, so the values are checked as ranges and the grid cell of every sphere exactly."""
    world, setup, _ = rb.record(scenes_lib, "random_scene")
    kind, items = world
    assert kind == "list"
    assert items[0] == ("sphere", (0., -1000., 0.), 1000., lamb(CHECKER))                                  # :18-22
    assert items[-3:] == [("sphere", (0., 1., 0.), 1., ("dielectric", 1.5)), ("sphere", (-4., 1., 0.), 1., lamb(solid(0.4, 0.2, 0.1))),
                          ("sphere", (4., 1., 0.), 1., ("metal", (0.7, 0.6, 0.5), 0.0))]                  # :71-85
    small = items[1:-3]
    assert 22 * 22 - 8 <= len(small) <= 22 * 22                                                             # the (4, 0.2, 0) exclusion takes a few
    cells, counts = set(), {"moving": 0, "metal": 0, "glass": 0}
    for it in small:
        if it[0] == "moving_sphere":                                                                        # :33-45
            _, c0, c1, t0, t1, r, mat = it
            counts["moving"] += 1
            assert (t0, t1) == (0., 1.) and c1[0] == c0[0] and c1[2] == c0[2] and 0.0 <= c1[1] - c0[1] < 0.5
            assert mat[0] == "lambertian" and mat[1][0] == "solid" and all(0.0 <= x < 1.0 for x in mat[1][1])
        else:
            _, c0, r, mat = it
            if mat[0] == "metal":                                                                           # :46-60
                counts["metal"] += 1
                assert all(0.0 < x <= 0.5 for x in mat[1]) and 0.0 <= mat[2] < 0.5
            else:                                                                                            # :61-68
                counts["glass"] += 1
                assert mat == ("dielectric", 1.5)
        assert r == 0.2 and c0[1] == 0.2
        a, b = math.floor(c0[0] - 0.9), math.floor(c0[2] - 0.9)
        assert -11 <= a < 11 and -11 <= b < 11 and (a, b) not in cells                                    # one per cell, in a-major order
        assert cells == set() or (a, b) > last
        last = (a, b)
        cells.add((a, b))
        assert math.dist(c0, (4.0, 0.2, 0.0)) > 0.9                                                         # :31
    n = len(small)
    assert abs(counts["moving"] / n - 0.80) < 0.06 and abs(counts["metal"] / n - 0.15) < 0.05 and abs(counts["glass"] / n - 0.05) < 0.04
    # mean albedo of the diffuse spheres: E[U*U] = 1/4 per channel
    alb = np.array([it[6][1][1] for it in small if it[0] == "moving_sphere"])
    assert abs(alb.mean() - 0.25) < 0.03


def test_spheres_1m_follows_the_survey_definition(scenes_lib):
    """SURVEY.md section 8(d) config 5 (build-defined, frozen in BASELINE.md): n spheres of radius 1.5, centres uniform in
    x, z in [-400, 400), y in [0, 800); 80 % Lambertian (U*U), 15 % Metal (0.5 (1 - U), fuzz 0.5 U), 5 % Dielectric 1.5; one XZ
    light x, z in [-200, 200) at y = 1000 emitting 7; background (0.7, 0.8, 1.0); camera (0, 400, -1600) -> (0, 400, 0), vfov 40."""
    n = 20000
    world, setup, _ = rb.record(scenes_lib, "spheres_1m", param=n)
    kind, items = world
    lights = [it for it in items if it[0] == "rectangle"]
    assert lights == [rect(XZ, (-200., 200.), (-200., 200.), 1000., ("diffuse_light", solid(7.)))]
    groups = [it for it in items if it[0] == "bvh_tree"]
    spheres = [s for g in groups for s in g[1][1]] + [it for it in items if it[0] == "sphere"]
    assert len(spheres) == n and all(s[2] == 1.5 for s in spheres)
    c = np.array([s[1] for s in spheres])
    assert c[:, 0].min() >= -400 and c[:, 0].max() < 400 and c[:, 2].min() >= -400 and c[:, 2].max() < 400
    assert c[:, 1].min() >= 0 and c[:, 1].max() < 800
    assert np.abs(c.mean(axis=0) - [0, 400, 0]).max() < 8 and np.abs(c.std(axis=0) - 800 / math.sqrt(12)).max() < 5
    kinds = [s[3][0] for s in spheres]
    assert abs(kinds.count("lambertian") / n - 0.80) < 0.02 and abs(kinds.count("metal") / n - 0.15) < 0.02
    assert abs(kinds.count("dielectric") / n - 0.05) < 0.01
    assert all(s[3] == ("dielectric", 1.5) for s in spheres if s[3][0] == "dielectric")
    assert tuple(setup.background) == (0.7, 0.8, 1.0) and tuple(setup.camera.lookfrom) == (0, 400, -1600)
    assert tuple(setup.camera.lookat) == (0, 400, 0) and setup.camera.vertical_fov == 40 and setup.camera.aperture == 0


def test_final_scene_object_list(scenes_lib, earth):
    """scenes.rs:238-334, the fixed part, object for object; the random part (400 heights, 1000 centres) by range."""
    world, _, _ = rb.record(scenes_lib, "final_scene", earth)
    kind, items = world
    assert kind == "list" and len(items) == 11
    floor = items[0]                                                                                         # :244-255
    assert floor[0] == "bvh_tree" and len(floor[1][1]) == 400
    for n, cube in enumerate(floor[1][1]):
        i, j = divmod(n, 20)
        assert cube[0] == "cube" and cube[3] == lamb(solid(0.48, 0.83, 0.53))
        assert cube[1] == (-1000. + i * 100., 0., -1000. + j * 100.)
        assert cube[2][0] == cube[1][0] + 100. and cube[2][2] == cube[1][2] + 100. and 1. <= cube[2][1] < 101.
    assert items[1] == rect(XZ, (123., 423.), (147., 412.), 554., ("diffuse_light", solid(7.)))           # :259-260
    assert items[2] == ("moving_sphere", (400., 400., 400.), (430., 400., 400.), 0., 1., 50., lamb(solid(0.7, 0.3, 0.1)))   # :262-269
    assert items[3] == ("sphere", (260., 150., 45.), 50., ("dielectric", 1.5))                              # :271-275
    assert items[4] == ("sphere", (0., 150., 45.), 50., ("metal", (0.8, 0.8, 0.9), 1.0))                    # :276-280 (fuzz 10 clamped to 1, material.rs:114)
    boundary = ("sphere", (360., 150., 145.), 70., ("dielectric", 1.5))
    assert items[5] == boundary                                                                              # :282-287
    assert items[6] == ("constant_medium", boundary, 0.2, solid(0.2, 0.4, 0.9))                             # :288-292
    assert items[7] == ("constant_medium", ("sphere", (0., 0., 0.), 5000., ("dielectric", 1.5)), 0.0001, solid(1.))   # :293-301
    assert items[8] == ("sphere", (400., 200., 400.), 100., lamb(("image", True, 1200, 600)))              # :303-308
    assert items[9] == ("sphere", (220., 280., 300.), 80., lamb(("noise", 0.1)))                            # :309-314
    t = items[10]                                                                                            # :316-331
    assert t[0] == "translate" and t[2] == (-100., 270., 395.) and t[1][0] == "rotate_y" and t[1][2] == 15.
    cluster = t[1][1]
    assert cluster[0] == "bvh_tree" and len(cluster[1][1]) == 1000
    c = np.array([s[1] for s in cluster[1][1]])
    assert all(s[0] == "sphere" and s[2] == 10. and s[3] == WHITE for s in cluster[1][1])
    assert c.min() >= 0 and c.max() < 165 and np.abs(c.mean(axis=0) - 82.5).max() < 6
