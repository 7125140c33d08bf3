"""Generate tests/golden/golden_windows.npz: windows of the BASELINE-size frames from the CPU oracle at the configs' full sample counts
(tests/golden_cases.py WINDOWS).  About seven minutes of 256 host threads (hours on 8): run where the cores are —
    gpurun -- 'python3 tests/golden/make_golden_windows.py gpurun_out/golden_windows.npz'
and copy the file to tests/golden/.  The reference itself cannot run (no Rust toolchain, unseedable RNG): the windows come from the
restatement, which tests/test_oracle_kat.py and tests/test_oracle_png_pins.py pin; the GPU tests re-render a few pixels of every window live."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import rto  # noqa: E402
from rttnw_amd import abi, scene as S  # noqa: E402
import util  # noqa: E402

from golden_cases import WINDOWS  # noqa: E402

if __name__ == "__main__":
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "golden_windows.npz")
    b = rto.binding()
    scenes = abi.Binding(C.CDLL(os.path.join(ROOT, "rttnw_amd", "host", "librttnw_scenes.so")), "", abi.SCENES_FUNCS)
    earth = S.load_earth()
    out, built = {}, {}
    for key, name, w, h, spp, x0, y0, cw, ch, want_var in WINDOWS:
        if name not in built:
            built[name] = util.build(b, scenes, name, earth)
        sc, setup = built[name]
        cam, p = util.params_for(setup, w, h, spp)
        t0 = time.time()
        lin, rgba, var, _ = rto.render_window(sc, cam, p, x0, y0, x0 + cw, y0 + ch, want_var=want_var)
        out[key + "_linear"] = lin
        out[key + "_rgba8"] = rgba
        if want_var:
            out[key + "_var"] = var
        print("%-14s %s %dx%d spp %d window (%d, %d) %dx%d: mean %s  (%.1f s)" % (key, name, w, h, spp, x0, y0, cw, ch, lin.mean(axis=(0, 1)), time.time() - t0), flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
    np.savez_compressed(dst, **out)
