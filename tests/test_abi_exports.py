"""The C-ABI library loads on a CPU-only box and exports every symbol include/rttnw_hip.h declares;
with no GPU every device entry point fails loudly (there is no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from rttnw_amd import abi, library
from rttnw_amd import scene as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\((?!\*)" % prefix, text)))


def test_hip_library_exports_every_declared_symbol():
    names = declared("rttnw_hip.h", "rttnw_")
    assert len(names) >= 30
    lib = C.CDLL(library.HIP_LIB)
    for n in names:
        assert hasattr(lib, n), "librttnw_hip.so does not export %s" % n
    assert set(abi.exported_symbols()) <= set(names) | {"rttnw_builder"}


def test_scenes_library_exports():
    lib = C.CDLL(library.SCENES_LIB)
    for n in declared("rttnw_scenes.h", "rttnw_scenes_"):
        assert hasattr(lib, n)


def test_builder_table_matches_direct_symbols():
    b = library.product()
    assert b.abi_version() == abi.ABI_VERSION == 3
    table = C.cast(b.builder(), C.POINTER(C.c_void_p * len(abi.BUILDER_FUNCS))).contents
    for i, (name, _, _) in enumerate(abi.BUILDER_FUNCS):
        direct = C.cast(getattr(b.lib, "rttnw_" + name), C.c_void_p).value
        assert table[i] == direct, name


def test_argument_validation_without_touching_the_device():
    b = library.product()
    sc = S.Scene(b)
    with pytest.raises(abi.RttnwError):
        sc.lambertian(12345)                       # unknown texture id
    t = sc.solid(0.5, 0.5, 0.5)
    with pytest.raises(abi.RttnwError):
        sc.sphere((0, 0, 0), 1.0, t)               # a texture is not a material
    m = sc.lambertian(t)
    s = sc.sphere((0, 0, 0), 1.0, m)
    lst = sc.list([s])
    with pytest.raises(abi.RttnwError):
        sc.push(lst, lst)                          # a list cannot contain itself
    with pytest.raises(abi.RttnwError):
        sc.rectangle(7, (0, 1), (0, 1), 0.0, m)    # bad plane
    lay = abi.TileLayout()
    assert b.tile_layout_get(800, 800, 8, C.byref(lay)) == 0
    assert (lay.tiles_x, lay.tiles_y, lay.n_tiles, lay.tiles_per_rank, lay.pixels_per_rank) == (100, 100, 10000, 1250, 80000)
    assert b.tile_layout_get(45, 37, 4, C.byref(lay)) == 0
    assert (lay.tiles_x, lay.tiles_y, lay.tiles_per_rank) == (6, 5, 8)


@pytest.mark.skipif(library.product().device_count() > 0, reason="a GPU is present")
def test_no_gpu_means_loud_failure():
    b = library.product()
    sc = S.Scene(b)
    w = sc.list([sc.sphere((0, 0, 0), 1.0, sc.lambertian((0.5, 0.5, 0.5)))])
    sc.set_world(w)
    with pytest.raises(abi.RttnwError, match="no CPU fallback"):
        sc.commit()
