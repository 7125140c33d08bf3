"""Framebuffer partition closed forms (rttnw_amd/tiles.py = include/rttnw_hip.h rttnw_tile_layout)."""
import numpy as np
import pytest

from rttnw_amd import tiles


@pytest.mark.parametrize("w,h,world", [(800, 800, 8), (64, 64, 2), (45, 37, 4), (8, 8, 1), (17, 9, 3), (1600, 1600, 8)])
def test_partition_is_a_bijection_and_balanced(w, h, world):
    lay = tiles.layout(w, h, world)
    owner, idx = tiles.packed_index(w, h, world)
    assert owner.min() >= 0 and owner.max() < world and idx.max() < lay["pixels_per_rank"]
    key = owner.astype(np.int64) * lay["pixels_per_rank"] + idx
    assert len(np.unique(key)) == w * h                                          # no two pixels share a slot
    counts = np.bincount((owner[::8, ::8]).ravel(), minlength=world)             # tiles per rank
    assert counts.max() - counts.min() <= 1
    if world > 1 and lay["tiles_y"] >= world and lay["tiles_x"] % world == 0:
        # the row rotation spreads one column of tiles over all ranks (no vertical stripes)
        assert len(np.unique(owner[::8, 0])) == world


@pytest.mark.parametrize("w,h,world", [(64, 64, 2), (45, 37, 4), (33, 70, 8)])
def test_pack_untile_roundtrip(w, h, world):
    rng = np.random.default_rng(0)
    img = rng.random((h, w, 3))
    gathered = np.stack([tiles.pack_rank(img, r, world) for r in range(world)])
    assert np.array_equal(tiles.untile_reference(gathered, w, h, world), img)
    for r in range(world):                                                       # pad slots stay zero
        assert (gathered[r][:, 3] <= 1).all()
