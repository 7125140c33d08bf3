"""Host lockstep model (policy 4: asynchronous shade phases) with leaf-step policies (tests/hostsim hostsim_wave_model)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rttnw_amd import abi, scene as S
lib = C.CDLL(os.path.join(ROOT, "tests", "hostsim", "libhostsim.so"))
b = abi.Binding(lib, "rttnw_", abi.BUILDER_FUNCS); b.add([("builder", C.c_void_p, [])])
scenes = abi.Binding(C.CDLL(os.path.join(ROOT, "rttnw_amd", "host", "librttnw_scenes.so")), "", abi.SCENES_FUNCS)
lib.hostsim_wave_model.argtypes = [C.c_void_p, C.POINTER(abi.CameraDesc), C.POINTER(abi.Params), C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p]
name = sys.argv[1] if len(sys.argv) > 1 else "final_scene"
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 8
jobs = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
earth = S.load_earth() if name in ("final_scene", "earth") else None
sc, setup = S.build(b, scenes, name, earth)
cam, p = S.params_for(setup, 800, 800, 1000)
p.precision = abi.F64
# VALU instructions (f64 kernel, rough): node step 60, leaf code by kind: sphere 80, moving 110, rect 45, box 200; a leaf step's fixed part 25; shade phase 1030 + begin 150 + hand-out 100
KC = [80, 110, 45, 200]
def run(leaf_policy, T=56):
    out = np.zeros(64, dtype=np.uint64)
    lib.hostsim_wave_model(sc.handle, C.byref(cam), C.byref(p), 4, 2, T, waves, jobs | (leaf_policy << 28), out.ctypes.data)
    o = [int(x) for x in out]
    phases, nexec, nlanes, lexec, llanes, samples = o[:6]
    leaf_cost = sum(o[6 + m] * (25 + sum(KC[k] for k in range(4) if m >> k & 1)) for m in range(16)) + o[49] * 230
    total = nexec * 60 + leaf_cost + phases * 1030 + o[34] * 150 + phases * 100
    by_kind = [sum(o[6 + m] for m in range(16) if m >> k & 1) for k in range(4)]
    print("leaf policy %d: phases/sample %.3f  node steps/sample %.2f (%.1f lanes)  leaf steps/sample %.2f (%.1f lanes)  executions of kind code / sample: sphere %.2f moving %.2f rect %.2f box %.2f (lanes: %s)  instr/sample: node %.0f leaf %.0f shade+ %.0f total %.0f"
          % (leaf_policy, phases / samples, nexec / samples, nlanes / max(1, nexec), lexec / samples, llanes / max(1, lexec),
             by_kind[0] / samples, by_kind[1] / samples, by_kind[2] / samples, by_kind[3] / samples,
             " ".join("%.1f" % (o[40 + k] / max(1, by_kind[k])) for k in range(4)),
             nexec * 60 / samples, leaf_cost / samples, (phases * 1130 + o[34] * 150) / samples, total / samples))
    return total / samples
base = run(0)
for thr in ("1,6", "2,6", "4,6", "6,6", "8,8", "4,8", "6,10", "12,12", "16,16", "24,16"):
    os.environ["HOSTSIM_LEAF_THR"] = thr
    print("thresholds", thr, end=": ")
    c = run(15)
    print("   against the kernel's: %+.1f%%" % (100 * (c / base - 1)))
