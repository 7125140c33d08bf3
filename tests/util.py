"""Helpers shared by the tests."""
import ctypes as C

import numpy as np

from rttnw_amd import abi
from rttnw_amd import scene as S


def build(binding, scenes_lib, name, earth=None, param=0, seed=0x5EED0001, bvh=None):
    sc = S.Scene(binding, seed, scenes_binding=scenes_lib)
    if bvh is not None:
        sc.set_bvh_builder(bvh)
    setup = sc.build_named(name, earth_rgba=earth, param=param)
    return sc, setup


def params_for(setup, w, h, spp, **kw):
    cam = abi.CameraDesc.from_buffer_copy(setup.camera)
    cam.aspect_ratio = w / h
    p = S.make_params(w, h, spp, background=tuple(setup.background), **kw)
    return cam, p


def hostsim_render(hostsim, sc, cam, p, n_threads=0):
    lin = np.zeros((p.height, p.width, 3))
    st = abi.Stats()
    rc = hostsim.lib.hostsim_render(sc.handle, C.byref(cam), C.byref(p), lin.ctypes.data, C.byref(st), n_threads)
    assert rc == 0
    return lin, st


def block_means(rgb, n):
    h, w, _ = rgb.shape
    bh, bw = h // n, w // n
    return np.array([[[rgb[r * bh:(r + 1) * bh, c * bw:(c + 1) * bw, ch].mean() for c in range(n)]
                      for r in range(n)] for ch in range(3)])
