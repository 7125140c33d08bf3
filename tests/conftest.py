import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _make(path):
    subprocess.run(["make", "-C", os.path.join(ROOT, path), "-s"], check=True)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/librttnw_oracle.so) — the checker, never the thing measured."""
    from oracle import rto
    return rto.binding()


@pytest.fixture(scope="session")
def scenes_lib():
    from rttnw_amd import abi
    so = os.path.join(ROOT, "rttnw_amd", "host", "librttnw_scenes.so")
    if not os.path.exists(so):
        _make("rttnw_amd/host")
    return abi.Binding(C.CDLL(so), "", abi.SCENES_FUNCS)


@pytest.fixture(scope="session")
def hostsim():
    """Host build of the product's tracing core + lowering (tests/hostsim) — test infrastructure."""
    from rttnw_amd import abi
    so = os.path.join(ROOT, "tests", "hostsim", "libhostsim.so")
    _make("tests/hostsim")
    lib = C.CDLL(so)
    b = abi.Binding(lib, "rttnw_", abi.BUILDER_FUNCS)
    b.add([("builder", C.c_void_p, []),
           ("debug_scene_nodes", C.c_int, [abi.scene_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_int32)]),
           ("debug_scene_nodes4", C.c_int, [abi.scene_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_int32)]),
           ("hittable_bounds", C.c_int, [abi.scene_p, abi.c_id, C.c_double, C.c_double, C.POINTER(C.c_double)])])
    lib.hostsim_render.restype = C.c_int
    lib.hostsim_render.argtypes = [C.c_void_p, C.POINTER(abi.CameraDesc), C.POINTER(abi.Params), C.c_void_p,
                                   C.POINTER(abi.Stats), C.c_int]
    lib.hostsim_scene_dims.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    return b


@pytest.fixture(scope="session")
def earth():
    from rttnw_amd import scene
    return scene.load_earth()


@pytest.fixture(scope="session")
def gpu():
    """The product binding on a box with a GPU; the HIP extension must be the thing that runs."""
    from rttnw_amd import library
    b = library.product()
    assert b.device_count() >= 1, "no HIP device visible"
    return b
