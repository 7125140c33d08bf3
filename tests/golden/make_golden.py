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

from golden_cases import CASES  # noqa: E402  (key, scene, w, h, spp, spp_chunk, param)

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
