"""Lockstep model of the lane-owns-path kernel's walk (tests/hostsim hostsim_wave_model): what a step policy would cost,
counted on the host before anything is built for the device.
Usage: python profiles/experiments/wave_model.py [scene] [waves] [jobs per wave] [f64|f32]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rttnw_amd import abi, scene as S  # noqa: E402

subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostsim")])
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "rttnw_amd", "host")])
lib = C.CDLL(os.path.join(ROOT, "tests", "hostsim", "libhostsim.so"))
b = abi.Binding(lib, "rttnw_", abi.BUILDER_FUNCS)
b.add([("builder", C.c_void_p, [])])
scenes = abi.Binding(C.CDLL(os.path.join(ROOT, "rttnw_amd", "host", "librttnw_scenes.so")), "", abi.SCENES_FUNCS)
lib.hostsim_wave_model.argtypes = [C.c_void_p, C.POINTER(abi.CameraDesc), C.POINTER(abi.Params), C.c_int, C.c_int, C.c_int, C.c_uint32,
                                   C.c_uint32, C.c_void_p]

name = sys.argv[1] if len(sys.argv) > 1 else "final_scene"
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 24
jobs = int(sys.argv[3]) if len(sys.argv) > 3 else 192
prec = sys.argv[4] if len(sys.argv) > 4 else "f64"
earth = S.load_earth() if name in ("final_scene", "earth") else None
sc, setup = S.build(b, scenes, name, earth)
w = h = 800 if name != "cornell_box" else 500
cam, p = S.params_for(setup, w, h, 1000)
p.precision = abi.F32 if prec == "f32" else abi.F64

# cost of one execution, in units of a node step: a leaf step by the record kinds it serves (serialised), rough
KIND_COST = {0: 1.6, 1: 2.2, 2: 1.2, 3: 3.2}  # sphere, moving, rect, box  (bit index = PRIM kind)


def run(policy, a, bb):
    out = np.zeros(64, dtype=np.uint64)
    lib.hostsim_wave_model(sc.handle, C.byref(cam), C.byref(p), policy, a, bb, waves, jobs, out.ctypes.data)
    rounds, nexec, nlanes, lexec, llanes, samples = [int(x) for x in out[:6]]
    by_set = out[6:22]
    return rounds, nexec, nlanes, lexec, llanes, samples, by_set


def report(tag, r):
    rounds, nexec, nlanes, lexec, llanes, samples, by_set = r
    for cl in (1.5, 2.0, 2.5):
        pass
    costs = [nexec + cl * lexec for cl in (1.5, 2.0, 2.5)]
    print("%-22s rounds/sample %.2f  node exec/round %.2f (%.1f lanes)  leaf exec/round %.2f (%.1f lanes)  walk cost/sample %s"
          % (tag, rounds / samples, nexec / rounds, nlanes / max(1, nexec), lexec / rounds, llanes / max(1, lexec),
             " ".join("%.1f" % (c / samples) for c in costs)))
    return costs


base = report("trips 2N+L (kernel)", run(0, 2, 0))
for a in (1, 3):
    report("trips %dN+L" % a, run(0, a, 0))
for a in (2, 3):
    c = report("stash in trip, %dN+L" % a, run(3, a, 0))
    print("      against the kernel's: %s" % " ".join("%+.1f%%" % (100 * (x / y - 1)) for x, y in zip(c, base)))
for thr in (160,):
    c = report("vote leaf>=%d/256" % thr, run(1, 0, thr))
    print("      against the kernel's: %s" % " ".join("%+.1f%%" % (100 * (x / y - 1)) for x, y in zip(c, base)))
for a, bb in ((8, 8),):
    c = report("drain nodes<%d leaves<%d" % (a, bb), run(2, a, bb))
    print("      against the kernel's: %s" % " ".join("%+.1f%%" % (100 * (x / y - 1)) for x, y in zip(c, base)))
