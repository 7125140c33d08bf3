"""T3 — the oracle against the only result artefacts the reference commits: cornel_box.png / image.png.

tests/golden/reference_png_stats.json holds block statistics of those PNGs (made by
tests/golden/make_reference_png_stats.py in the build container).  The reference's RNG is OS-seeded,
so the comparison is statistical: block means of the gamma-encoded 8-bit image at the reference's own
spp (the sqrt makes the 8-bit mean depend on spp), the exact black frame, the saturated light patch.
"""
import json
import os

import numpy as np

import util
from oracle import rto

STATS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_png_stats.json")))


def test_cornell_box_matches_reference_png(oracle, scenes_lib):
    ref = STATS["cornel_box"]
    sc, setup = util.build(oracle, scenes_lib, "cornell_box")
    assert (setup.width, setup.height, setup.spp) == (600, 600, 200)            # main.rs:137-141
    # half resolution, the reference's 200 spp: block means are resolution-independent
    cam, p = util.params_for(setup, 300, 300, 200)
    _, rgba, _ = rto.render(sc, cam, p)
    rgb = rgba[..., :3].astype(np.float64)
    assert (rgba[..., 3] == 255).all()
    bm = util.block_means(rgb, 6)
    rb = np.array(ref["block_means_rgb"])
    # measured at 600x600: max |diff| 0.87, mean 0.24 (8-bit units); 300x300 has 4x fewer pixels per block
    assert np.abs(bm - rb).max() < 2.5, np.abs(bm - rb).max()
    assert np.abs(bm - rb).mean() < 0.6
    assert np.abs(rgb.mean(axis=(0, 1)) - np.array(ref["mean_rgb"])).max() < 0.5
    # Q1 signature: the tall block's front (block row 3-4, col 2) is nearly black; a "fixed" renderer gives ~90-120
    assert bm[0][3][2] < 20 and bm[0][4][2] < 20
    # 14-px black frame at 600 px -> 7 px at 300 px (analytic 7.04); light patch saturated
    black = rgba[..., :3].max(axis=2) == 0
    assert int(np.argmax(~black.all(axis=0))) == 7 and int(np.argmax(~black.all(axis=1))) == 7
    assert ref["first_lit_col"] == 14 and ref["first_lit_row"] == 14
    assert (rgba[43:48, 131:170, :3] == 255).all()
    assert abs(black.mean() - ref["frac_black"]) < 0.01
    assert abs((rgba[..., :3].max(axis=2) == 255).mean() - ref["frac_saturated"]) < 0.002


def test_quirk_off_differs_from_reference_png(oracle, scenes_lib):
    """With Q1 'fixed' the block fronts light up — proving the pin is sensitive to the quirk."""
    ref = np.array(STATS["cornel_box"]["block_means_rgb"])
    sc, setup = util.build(oracle, scenes_lib, "cornell_box")
    cam, p = util.params_for(setup, 120, 120, 200, quirks=0)
    _, rgba, _ = rto.render(sc, cam, p)
    bm = util.block_means(rgba[..., :3].astype(np.float64), 6)
    assert bm[0][3][2] > 30 and np.abs(bm - ref).max() > 25  # with Q1 the reference has 13.1 there


def test_final_scene_global_statistics(oracle, scenes_lib, earth):
    """image.png: geometry is random per run in the reference, so only global statistics can be compared."""
    ref = STATS["image"]
    sc, setup = util.build(oracle, scenes_lib, "final_scene", earth)
    assert (setup.width, setup.height, setup.spp) == (800, 800, 10000)          # main.rs:165-169
    cam, p = util.params_for(setup, 160, 160, 256)
    _, rgba, _ = rto.render(sc, cam, p)
    rgb = rgba[..., :3].astype(np.float64)
    mean = rgb.mean(axis=(0, 1))
    # reference means R 71.87 G 84.70 B 74.88 at 10000 spp; 256 spp biases the gamma-encoded mean down a little
    assert np.abs(mean - np.array(ref["mean_rgb"])).max() < 6.0, mean
    assert mean[1] > mean[0] and mean[1] > mean[2]                               # green-ish ground dominates
    assert abs((rgba[..., :3].max(axis=2) == 255).mean() - ref["frac_saturated"]) < 0.02
    # (image.png has no pure-black pixel at 10000 spp; at 256 spp the dim fog-lit background still has some)
    assert (rgba[..., :3].max(axis=2) == 0).mean() < 0.25
