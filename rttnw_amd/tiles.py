"""Closed forms of the framebuffer partition (include/rttnw_hip.h `rttnw_tile_layout`).

8x8-pixel tiles; tile (tx, ty) has the permuted id  ty*tiles_x + (tx+ty) % tiles_x  (every tile row is
rotated by its index so a column of tiles spreads over all ranks); owner = permuted % world, and the
owner stores it as its (permuted // world)-th tile.  Packed order = tiles in that local order, 64
pixels per tile, row-major inside the tile, 4 reals (r, g, b, 1) per pixel; every rank is padded to
ceil(n_tiles / world) tiles so that one flat gather has a uniform count.
"""
import numpy as np

TILE = 8


def layout(width, height, world):
    tiles_x = (width + TILE - 1) // TILE
    tiles_y = (height + TILE - 1) // TILE
    n_tiles = tiles_x * tiles_y
    tiles_per_rank = (n_tiles + world - 1) // world
    return dict(tiles_x=tiles_x, tiles_y=tiles_y, n_tiles=n_tiles, tiles_per_rank=tiles_per_rank,
                pixels_per_rank=tiles_per_rank * TILE * TILE)


def packed_index(width, height, world):
    """For every framebuffer pixel (row-major, top row first): (owner rank, index into that rank's packed buffer)."""
    lay = layout(width, height, world)
    y, x = np.mgrid[0:height, 0:width]
    tx, ty = x // TILE, y // TILE
    permuted = ty * lay["tiles_x"] + (tx + ty) % lay["tiles_x"]
    owner = permuted % world
    local = permuted // world
    idx = local * (TILE * TILE) + (y % TILE) * TILE + (x % TILE)
    return owner, idx


def pack_rank(image, rank, world):
    """Pack the pixels `rank` owns out of a full HxWxC image (test/reference helper)."""
    h, w, c = image.shape
    lay = layout(w, h, world)
    owner, idx = packed_index(w, h, world)
    out = np.zeros((lay["pixels_per_rank"], 4), dtype=image.dtype)
    m = owner == rank
    out[idx[m], :c] = image[m]
    out[idx[m], 3] = 1
    return out


def untile_reference(gathered, width, height, world):
    """numpy statement of rttnw_untile_device: gathered [world*pixels_per_rank, 4] -> HxWx3."""
    lay = layout(width, height, world)
    owner, idx = packed_index(width, height, world)
    flat = np.asarray(gathered).reshape(world * lay["pixels_per_rank"], 4)
    return flat[owner * lay["pixels_per_rank"] + idx][..., :3]
