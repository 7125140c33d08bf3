"""rttnw_amd — MI355X-native path tracer behind luliic2/rttnw's Scene/Camera/Hittable/Material surface.

Layout: csrc/ = hand-written HIP kernels + the C ABI (include/rttnw_hip.h); host/ = the scenes.rs
mirror (C++); this package = ctypes mirror of the reference's vocabulary + render drivers.
"""
from . import abi, tiles  # noqa: F401
from .abi import F32, F64, QUIRKS_REFERENCE, XY, XZ, YZ  # noqa: F401
from .scene import Scene, camera_desc, load_earth, make_params  # noqa: F401


def product_scene(scene_seed=0x5EED0001):
    """A Scene bound to the HIP library (raises if the extension is not built)."""
    from . import library
    return Scene(library.product(), scene_seed, scenes_binding=library.scenes())
