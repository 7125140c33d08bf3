"""Derive statistical pins from the two renders the reference commits (run in the build container
only: it reads /root/reference, which does not exist on the GPU box).

Output: tests/golden/reference_png_stats.json — derived data (means / fractions), not the images.
"""
import json
import os

import numpy as np
from PIL import Image

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_png_stats.json")


def block_means(rgb, n):
    h, w, _ = rgb.shape
    bh, bw = h // n, w // n
    return [[[float(rgb[r * bh:(r + 1) * bh, c * bw:(c + 1) * bw, ch].mean()) for c in range(n)]
             for r in range(n)] for ch in range(3)]


def stats(path, n_blocks):
    im = np.asarray(Image.open(path).convert("RGBA"), dtype=np.uint8)
    rgb = im[..., :3].astype(np.float64)
    h, w, _ = rgb.shape
    black = (im[..., :3].max(axis=2) == 0)
    first_lit_col = int(np.argmax(~black.all(axis=0)))
    first_lit_row = int(np.argmax(~black.all(axis=1)))
    return {
        "file": os.path.basename(path), "width": w, "height": h,
        "alpha_all_255": bool((im[..., 3] == 255).all()),
        "mean_rgb": [float(rgb[..., c].mean()) for c in range(3)],
        "frac_saturated": float((im[..., :3].max(axis=2) == 255).mean()),
        "frac_black": float(black.mean()),
        "first_lit_col": first_lit_col, "first_lit_row": first_lit_row,
        "block_means_rgb": block_means(rgb, n_blocks), "n_blocks": n_blocks,
    }


# ---------------------------------------------------------------------------------------------
# image.png (final_scene, scenes.rs:238-334): the deterministic objects — five spheres, the moving sphere, the
# light, the fog — sit at fixed places (scenes.rs:259-314); only the floor heights, the cluster's sphere centres and the
# Perlin tables are random per run.  Regions are discs INSIDE the projected silhouettes of those spheres (camera of
# main.rs:165-178 projected with camera.rs:32-61), chosen clear of occluders, plus rectangles of fog-only background and
# of the two halves of the sphere cluster.  Stored in image-size-independent form (fractions of width/height).
LOOKFROM, LOOKAT, VFOV = (478.0, 278.0, -600.0), (278.0, 278.0, 0.0), 40.0       # main.rs:165-178
FINAL_SPHERES = {                                                                # scenes.rs:259-314
    "moving": ((415.0, 400.0, 400.0), 50.0),   # MovingSphere centre at mid-shutter (400..430)
    "glass": ((260.0, 150.0, 45.0), 50.0),
    "metal": ((0.0, 150.0, 45.0), 50.0),
    "blue": ((360.0, 150.0, 145.0), 70.0),     # dielectric boundary + density-0.2 blue medium
    "earth": ((400.0, 200.0, 400.0), 100.0),
    "noise": ((220.0, 280.0, 300.0), 80.0),
}
# (sphere, dx, dy, r): a disc of radius r * rho centred (dx, dy) * rho off the projected centre (y down)
DISCS = {
    "earth_left": ("earth", -0.50, 0.00, 0.30), "earth_right": ("earth", 0.45, -0.45, 0.25),
    "earth_top": ("earth", 0.00, -0.60, 0.30), "earth_bottom": ("earth", -0.35, 0.55, 0.25),
    "blue_core": ("blue", 0.10, 0.20, 0.45), "glass_core": ("glass", 0.00, 0.00, 0.50),
    "glass_upper": ("glass", 0.00, -0.40, 0.30), "glass_lower": ("glass", 0.00, 0.40, 0.30),
    "metal_core": ("metal", -0.20, 0.00, 0.45), "moving_core": ("moving", 0.00, 0.00, 0.50),
    "noise_top": ("noise", 0.00, -0.45, 0.35), "noise_bottom": ("noise", 0.00, 0.45, 0.35),
}
RECTS = {  # (x0, y0, x1, y1) as fractions of the image
    "light_patch": (0.30, 0.05, 0.45, 0.10),
    "fog_right": (0.85, 0.19, 0.99, 0.56), "fog_upper_left": (0.02, 0.15, 0.12, 0.22),
    "cluster_left": (0.52, 0.34, 0.595, 0.49), "cluster_right": (0.65, 0.34, 0.76, 0.49),
    "floor_front": (0.55, 0.90, 0.98, 0.99),
}


def project(point, radius):
    """World point -> (s, t_from_top, rho) in fractions of the image (aspect 1)."""
    lf, la = np.array(LOOKFROM), np.array(LOOKAT)
    w = (lf - la) / np.linalg.norm(lf - la)
    u = np.cross([0.0, 1.0, 0.0], w)
    u /= np.linalg.norm(u)
    v = np.cross(w, u)
    d = np.array(point) - lf
    z = -(d @ w)
    hh = np.tan(np.radians(VFOV / 2))
    return 0.5 + (d @ u) / (z * 2 * hh), 0.5 - (d @ v) / (z * 2 * hh), radius / z / (2 * hh)


def final_scene_regions():
    regions = {}
    for name, (sph, dx, dy, r) in DISCS.items():
        cx, cy, rho = project(*FINAL_SPHERES[sph])
        regions[name] = {"disc": [float(cx + dx * rho), float(cy + dy * rho), float(r * rho)]}
    for name, rect in RECTS.items():
        regions[name] = {"rect": list(rect)}
    return regions


def region_mask(region, w, h):
    y, x = np.mgrid[0:h, 0:w]
    fx, fy = (x + 0.5) / w, (y + 0.5) / h
    if "disc" in region:
        cx, cy, r = region["disc"]
        return (fx - cx) ** 2 + (fy - cy) ** 2 <= r * r
    x0, y0, x1, y1 = region["rect"]
    return (fx >= x0) & (fx < x1) & (fy >= y0) & (fy < y1)


def region_stats(path):
    im = np.asarray(Image.open(path).convert("RGBA"), dtype=np.uint8)
    rgb = im[..., :3].astype(np.float64)
    h, w, _ = rgb.shape
    out = {}
    for name, region in final_scene_regions().items():
        m = region_mask(region, w, h)
        out[name] = dict(region, n_px=int(m.sum()), mean_rgb=[float(rgb[..., c][m].mean()) for c in range(3)],
                         frac_saturated=float((im[..., :3].max(axis=2) == 255)[m].mean()))
    return out


if __name__ == "__main__":
    out = {"cornel_box": stats(os.path.join(REF, "cornel_box.png"), 6),
           "image": stats(os.path.join(REF, "image.png"), 8)}
    # the light patch of cornel_box.png is exactly white
    im = np.asarray(Image.open(os.path.join(REF, "cornel_box.png")).convert("RGBA"))
    out["cornel_box"]["light_patch_rows_85_95_cols_260_340_all_255"] = bool(
        (im[85:96, 260:341, :3] == 255).all())
    out["image"]["regions"] = region_stats(os.path.join(REF, "image.png"))
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)
