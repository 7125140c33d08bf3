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


if __name__ == "__main__":
    out = {"cornel_box": stats(os.path.join(REF, "cornel_box.png"), 6),
           "image": stats(os.path.join(REF, "image.png"), 8)}
    # the light patch of cornel_box.png is exactly white
    im = np.asarray(Image.open(os.path.join(REF, "cornel_box.png")).convert("RGBA"))
    out["cornel_box"]["light_patch_rows_85_95_cols_260_340_all_255"] = bool(
        (im[85:96, 260:341, :3] == 255).all())
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)
