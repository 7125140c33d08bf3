"""TEST INFRASTRUCTURE — ctypes loader for the CPU oracle (oracle/librttnw_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from rttnw_amd import abi
from rttnw_amd.abi import CameraDesc, Params, Stats

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "librttnw_oracle.so")
_dp = C.POINTER(C.c_double)

_PROBES = [
    ("render", C.c_int, [abi.scene_p, C.POINTER(CameraDesc), C.POINTER(Params), C.c_void_p, C.c_void_p,
                         C.POINTER(Stats), C.c_int]),
    ("builder", C.c_void_p, []),
    ("probe_camera", C.c_int, [C.POINTER(CameraDesc), _dp]),
    ("probe_camera_ray", C.c_int, [C.POINTER(CameraDesc), C.c_double, C.c_double, C.c_uint64, C.c_uint64,
                                   C.c_uint64, _dp]),
    ("probe_hit", C.c_int, [abi.scene_p, abi.c_id, _dp, C.c_double, C.c_double, C.c_uint64, C.c_uint64,
                            C.c_uint64, C.c_uint32, C.c_uint32, _dp]),
    ("probe_scatter", C.c_int, [abi.scene_p, abi.c_id, _dp, _dp, C.c_uint64, C.c_uint64, C.c_uint64,
                                C.c_uint32, _dp]),
    ("probe_tex", C.c_int, [abi.scene_p, abi.c_id, C.c_double, C.c_double, _dp, _dp]),
    ("probe_perlin", C.c_int, [abi.scene_p, abi.c_id, _dp, C.c_uint32, _dp]),
    ("probe_perlin_tables", C.c_int, [abi.scene_p, abi.c_id, _dp, C.POINTER(C.c_uint32)]),
    ("probe_aabb", C.c_int, [_dp, _dp, C.c_double, C.c_double]),
    ("probe_bbox", C.c_int, [abi.scene_p, abi.c_id, C.c_double, C.c_double, _dp]),
    ("probe_schlick", C.c_double, [C.c_double, C.c_double]),
    ("probe_refract", None, [_dp, _dp, C.c_double, _dp]),
    ("probe_reflect", None, [_dp, _dp, _dp]),
    ("probe_quantise", C.c_int, [C.c_double]),
    ("probe_uniform", C.c_double, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]),
    ("probe_scene_rng", None, [C.c_uint64, C.c_uint64, C.c_uint32, _dp]),
    ("probe_sample", C.c_int, [abi.scene_p, C.POINTER(CameraDesc), C.POINTER(Params), C.c_uint32, C.c_uint32,
                               C.c_uint32, _dp]),
]

_binding = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def binding():
    """Bound oracle library (built on demand)."""
    global _binding
    if _binding is None:
        if not os.path.exists(_LIB):
            build()
        lib = C.CDLL(_LIB)
        b = abi.Binding(lib, "rto_", abi.BUILDER_FUNCS)
        b.add(_PROBES)
        _binding = b
    return _binding


def darr(*vals):
    a = np.array(vals, dtype=np.float64).ravel()
    return a, a.ctypes.data_as(_dp)


def render(scene, cam, params, n_threads=0, want_rgba8=True):
    """rto_render -> (linear HxWx3 float64, rgba8 HxWx4 uint8 or None, Stats)."""
    b = binding()
    h, w = params.height, params.width
    lin = np.zeros((h, w, 3), dtype=np.float64)
    rgba = np.zeros((h, w, 4), dtype=np.uint8) if want_rgba8 else None
    st = Stats()
    rc = b.render(scene.handle, C.byref(cam), C.byref(params), lin.ctypes.data,
                  rgba.ctypes.data if rgba is not None else None, C.byref(st), int(n_threads))
    abi.check(rc, b, "rto_render")
    return lin, rgba, st
