# crossover of the lane-owns-path kernel (async shade phases) against the decoupled kernel (3 blocks per CU in f64 since round 5), Msamples/s
import os, sys, time, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import util
from rttnw_amd import abi, library, render, scene as S
gpu = library.product(); scenes = library.scenes()
earth = S.load_earth()
def run(name, param, w, spp, precs):
    sc, setup = util.build(gpu, scenes, name, earth if name == "final_scene" else None, param)
    info = abi.Stats(); gpu.scene_info(sc.handle, info)
    for prec, pn in precs:
        res = {}
        for form in ("plain", "wave"):
            os.environ["RTTNW_KERNEL"] = form
            cam, p = util.params_for(setup, w, w, spp, precision=prec, seed=1)
            r = render.DeviceRenderer(sc, cam, p)
            st = abi.Stats(); r.trace(st); r.trace(st)
            res[form] = w * w * spp / st.kernel_ms / 1e3
        print("%s %6d nodes4 %6d %s: plain %.1f  wave %.1f Msamples/s" % (name, param, info.n_nodes, pn, res["plain"], res["wave"]), flush=True)
P = ((abi.F32, "f32"), (abi.F64, "f64"), (abi.F64_STRICT, "f64strict"))
run("final_scene", 0, 800, 500, P)
run("cornell_box", 0, 800, 500, P)
for n in (1000, 2000, 3000, 4000, 6000, 8000, 12000, 16000, 20000, 30000, 40000, 60000):
    run("spheres_1m", n, 512, 256, P)
del os.environ["RTTNW_KERNEL"]
