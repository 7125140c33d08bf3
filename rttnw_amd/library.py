"""Loads the in-tree native libraries.  There is no fallback: a missing or unloadable HIP library is
an error (build with `python -c "import __graft_entry__ as g; g.build()"` or `make -C rttnw_amd/csrc`)."""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.environ.get("RTTNW_HIP_LIB") or os.path.join(_HERE, "csrc", "librttnw_hip.so")  # env: debug builds
SCENES_LIB = os.path.join(_HERE, "host", "librttnw_scenes.so")

_product = None
_scenes = None


def product():
    """Binding of librttnw_hip.so (hand-written HIP kernels + C ABI)."""
    global _product
    if _product is None:
        if not os.path.exists(HIP_LIB):
            raise RuntimeError("rttnw_amd: %s is missing — the HIP extension is not built and there is no "
                               "CPU fallback (run __graft_entry__.build())" % HIP_LIB)
        # PyTorch ships its own copy of the HIP runtime.  Load it first so that this library binds to the same
        # runtime instance (a second instance in one process sees no devices); torch is plumbing only.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(HIP_LIB)
        b = abi.Binding(lib, "rttnw_", abi.BUILDER_FUNCS)
        b.add(abi.PRODUCT_FUNCS)
        if b.abi_version() != abi.ABI_VERSION:
            raise RuntimeError("rttnw_amd: ABI version mismatch")
        _product = b
    return _product


def scenes():
    """Binding of the host-side scene catalogue (scenes.rs mirror)."""
    global _scenes
    if _scenes is None:
        if not os.path.exists(SCENES_LIB):
            raise RuntimeError("rttnw_amd: %s is missing (run __graft_entry__.build())" % SCENES_LIB)
        _scenes = abi.Binding(C.CDLL(SCENES_LIB), "", abi.SCENES_FUNCS)
    return _scenes
