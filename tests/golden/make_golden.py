"""Generate the committed golden vectors from the CPU oracle (run in the build container).

The reference cannot run (no Rust toolchain, unseedable RNG), so goldens come from the restatement,
which is itself pinned by tests/test_oracle_kat.py and tests/test_oracle_png_pins.py.
Output: tests/golden/golden_images.npz — small f64 linear images + the parameters that made them.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import rto  # noqa: E402
from rttnw_amd import abi, scene as S  # noqa: E402
import util  # noqa: E402

CASES = [  # name, scene, w, h, spp, spp_chunk, param
    ("cornell_64", "cornell_box", 64, 64, 16, 4, 0),
    ("final_64", "final_scene", 64, 64, 16, 4, 0),
    ("final_ragged_45x37", "final_scene", 45, 37, 5, 2, 0),
    ("random_scene_48x27", "random_scene", 48, 27, 8, 8, 0),
    ("smoke_cornell_40", "smoke_cornell_box", 40, 40, 8, 3, 0),
    ("simple_light_48x27", "simple_light", 48, 27, 8, 4, 0),
    ("two_spheres_32x18", "two_spheres", 32, 18, 4, 4, 0),
    ("earth_32x18", "earth", 32, 18, 4, 4, 0),
    ("spheres_2k_48", "spheres_1m", 48, 48, 8, 4, 2000),
]

if __name__ == "__main__":
    b = rto.binding()
    scenes = abi.Binding(C.CDLL(os.path.join(ROOT, "rttnw_amd", "host", "librttnw_scenes.so")), "", abi.SCENES_FUNCS)
    earth = S.load_earth()
    out = {}
    for key, name, w, h, spp, chunk, param in CASES:
        sc, setup = util.build(b, scenes, name, earth, param)
        cam, p = util.params_for(setup, w, h, spp, spp_chunk=chunk)
        lin, rgba, _ = rto.render(sc, cam, p)
        out[key + "_linear"] = lin
        out[key + "_rgba8"] = rgba
        print(key, lin.mean(axis=(0, 1)))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden_images.npz"), **out)
