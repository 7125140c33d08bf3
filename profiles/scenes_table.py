"""All nine scenes.rs scenes (+ spheres_1m) through the F64 and F32 kernels on one GPU: Msamples/s of the trace kernel at the
reference's aspect ratio, 800 pixels wide, spp 256 (Msamples/s is spp-invariant), with the counted rays / node visits /
record tests per sample.  Run on the MI355X box from the repo root:  python profiles/scenes_table.py > table.md"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from rttnw_amd import abi, library, render, scene as S  # noqa: E402

gpu, scenes, earth = library.product(), library.scenes(), S.load_earth()
names = [scenes.scenes_name(n).decode() for n in range(1, 10)] + ["spheres_1m"]
print("| scene | size | 4-wide nodes | records | kernel form | f64 Msamples/s | f32 Msamples/s | rays / node visits / record tests per sample |")
print("|---|---|---|---|---|---|---|---|")
for name in names:
    sc, setup = S.build(gpu, scenes, name, earth)
    w = 800 if name != "spheres_1m" else 1024
    h = int(w / (setup.width / setup.height))
    info = abi.Stats()
    gpu.scene_info(sc.handle, info)
    rates, form, per = {}, None, None
    for prec in (abi.F64, abi.F32):
        cam, p = S.params_for(setup, w, h, 256, precision=prec, seed=1)
        r = render.DeviceRenderer(sc, cam, p)
        st = abi.Stats()
        r.trace(st)                      # warm-up (uploads the scene of this precision)
        r.trace(st)
        rates[prec] = w * h * 256 / st.kernel_ms / 1e3
        form = "decoupled" if st.reserved else "lane-owns-path"
        if prec == abi.F64:
            camc, pc = S.params_for(setup, w, h, 8, precision=prec, seed=1, collect_counters=1)
            rc = render.DeviceRenderer(sc, camc, pc)
            sc_ = abi.Stats()
            rc.trace(sc_)
            per = (sc_.rays / sc_.samples, sc_.nodes_visited / sc_.samples, sc_.prims_tested / sc_.samples)
    print("| %s | %dx%d | %d | %d | %s | %.0f | %.0f | %.2f / %.1f / %.1f |" % (name, w, h, info.n_nodes, info.n_prims, form, rates[abi.F64], rates[abi.F32], *per))
