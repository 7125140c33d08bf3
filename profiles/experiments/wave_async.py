"""Host lockstep model: the kernel's synchronous rounds against asynchronous shade phases (policy 4) (tests/hostsim hostsim_wave_model)."""
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
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 24
jobs = int(sys.argv[3]) if len(sys.argv) > 3 else 256
earth = S.load_earth() if name in ("final_scene", "earth") else None
sc, setup = S.build(b, scenes, name, earth)
cam, p = S.params_for(setup, 800, 800, 1000)
p.precision = abi.F64
# unit = one node step (1256 wave clocks on final_scene f64, profiles/r04/phases_final_scene_f64.txt): leaf step 2.46, shade phase 20.4, begin 3.5 (in a phase that begins paths), hand-out 3.8 per phase
CL, CS, CB, CH = 2.46, 20.4, 3.5, 3.8
def run(policy, a, bb):
    out = np.zeros(64, dtype=np.uint64)
    lib.hostsim_wave_model(sc.handle, C.byref(cam), C.byref(p), policy, a, bb, waves, jobs, out.ctypes.data)
    return [int(x) for x in out]
def cost(tag, o, sync):
    rounds, nexec, nlanes, lexec, llanes, samples = o[:6]
    began = o[34] if not sync else 0.77 * rounds
    served = o[33] / max(1, rounds) if not sync else float('nan')
    c = nexec + CL * lexec + CS * rounds + CB * began + CH * rounds
    print("%-26s shade phases/sample %.2f (lanes %.1f)  node exec/sample %.2f (%.1f lanes)  leaf exec/sample %.2f (%.1f lanes)  cost/sample %.1f  [walk %.1f shade+ %.1f]"
          % (tag, rounds / samples, served, nexec / samples, nlanes / max(1, nexec), lexec / samples, llanes / max(1, lexec), c / samples,
             (nexec + CL * lexec) / samples, (CS * rounds + CB * began + CH * rounds) / samples))
    return c / samples
base = cost("kernel: rounds, 2N+L", run(0, 2, 0), True)
for T in (64, 56, 48, 40, 32, 24, 16):
    for a in (2,):
        c = cost("async shade at %d, %dN+L" % (T, a), run(4, a, T), False)
        print("      against the kernel's: %+.1f%%" % (100 * (c / base - 1)))
