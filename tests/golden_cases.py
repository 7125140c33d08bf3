"""The golden cases of tests/golden/make_golden.py, shared by CPU and GPU parity tests."""
import os

import numpy as np

CASES = [  # key, scene, w, h, spp, spp_chunk, param
    ("cornell_64", "cornell_box", 64, 64, 16, 4, 0),
    ("final_64", "final_scene", 64, 64, 16, 4, 0),
    ("final_ragged_45x37", "final_scene", 45, 37, 5, 2, 0),
    ("random_scene_48x27", "random_scene", 48, 27, 8, 8, 0),
    ("smoke_cornell_40", "smoke_cornell_box", 40, 40, 8, 3, 0),
    ("simple_light_48x27", "simple_light", 48, 27, 8, 4, 0),
    ("two_spheres_32x18", "two_spheres", 32, 18, 4, 4, 0),
    ("earth_32x18", "earth", 32, 18, 4, 4, 0),
    ("two_perlin_48x27", "two_perlin_spheres", 48, 27, 8, 4, 0),     # Perlin at scale 4 on a radius-1000 sphere (scenes.rs:110-125)
    ("empty_cornell_40", "empty_cornell_box", 40, 40, 8, 3, 0),      # scenes.rs:157-173
    ("spheres_2k_48", "spheres_1m", 48, 48, 8, 4, 2000),
]


def load():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_images.npz"))
