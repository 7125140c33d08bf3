import sys, os, ctypes as C, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["RTTNW_DEBUG_SCHED"] = "1"
import util
from rttnw_amd import abi, library, render, scene as S
gpu = library.product(); scenes = library.scenes()
for name in ("final_scene", "cornell_box"):
    sg, setup = util.build(gpu, scenes, name, S.load_earth())
    cam, p = util.params_for(setup, 400, 400, 512, precision=abi.F32, collect_counters=1)
    lin, _, st = render.render_host(sg, cam, p)
    print(name, "kernel_ms", st.kernel_ms, "rays/sample", st.rays / st.samples, flush=True)
