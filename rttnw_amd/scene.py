"""Python mirror of the reference's Scene / Camera / Hittable / Material vocabulary.

A thin object layer over the C ABI (include/rttnw_hip.h): every method is one ABI call and is
named after the reference constructor it replaces (src/scenes.rs is the model caller).  The
class is generic over a `Binding` so the same scene script can drive any implementation of the
boundary; the product binding comes from `rttnw_amd.library`.
"""
import ctypes as C

import numpy as np

from . import abi
from .abi import CameraDesc, Params, SceneSetup, check, vec3


class Scene:
    """One scene under construction / committed.  Mirrors `List`-of-`Hittable` building in scenes.rs."""

    def __init__(self, binding, scene_seed=0x5EED0001, scenes_binding=None):
        self.b = binding
        self.seed = int(scene_seed)
        self.scenes = scenes_binding
        self.handle = abi.scene_p()
        check(binding.scene_create(self.seed, C.byref(self.handle)), binding, "scene_create")
        self._keep = []  # numpy buffers handed to the ABI during construction
        self.setup = None

    def close(self):
        if self.handle:
            self.b.scene_destroy(self.handle)
            self.handle = abi.scene_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _id(self, rc, what):
        return check(rc, self.b, what)

    # ---- textures (texture.rs)
    def solid(self, r, g=None, b=None):
        if g is None:
            r, g, b = (r, r, r) if np.isscalar(r) else r
        return self._id(self.b.tex_solid(self.handle, r, g, b), "tex_solid")

    def checker(self, odd, even):
        return self._id(self.b.tex_checker(self.handle, odd, even), "tex_checker")

    def noise(self, scale):
        return self._id(self.b.tex_noise(self.handle, scale), "tex_noise")

    def image(self, rgba):
        """rgba: HxWx4 uint8 array (top row first) or None (missing file -> cyan)."""
        if rgba is None:
            return self._id(self.b.tex_image_rgba8(self.handle, None, 0, 0), "tex_image_rgba8")
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        h, w, c = rgba.shape
        assert c == 4
        return self._id(self.b.tex_image_rgba8(self.handle, rgba.ctypes.data, w, h), "tex_image_rgba8")

    # ---- materials (material.rs)
    def lambertian(self, tex_or_rgb):
        tex = tex_or_rgb if isinstance(tex_or_rgb, int) else self.solid(tex_or_rgb)
        return self._id(self.b.mat_lambertian(self.handle, tex), "mat_lambertian")

    def metal(self, rgb, fuzz):
        return self._id(self.b.mat_metal(self.handle, rgb[0], rgb[1], rgb[2], fuzz), "mat_metal")

    def dielectric(self, ri):
        return self._id(self.b.mat_dielectric(self.handle, ri), "mat_dielectric")

    def diffuse_light(self, tex_or_rgb):
        tex = tex_or_rgb if isinstance(tex_or_rgb, int) else self.solid(tex_or_rgb)
        return self._id(self.b.mat_diffuse_light(self.handle, tex), "mat_diffuse_light")

    def isotropic(self, tex_or_rgb):
        tex = tex_or_rgb if isinstance(tex_or_rgb, int) else self.solid(tex_or_rgb)
        return self._id(self.b.mat_isotropic(self.handle, tex), "mat_isotropic")

    # ---- hittables (hittable.rs)
    def sphere(self, center, radius, mat):
        return self._id(self.b.sphere(self.handle, vec3(center), radius, mat), "sphere")

    def moving_sphere(self, c0, c1, t0, t1, radius, mat):
        return self._id(self.b.moving_sphere(self.handle, vec3(c0), vec3(c1), t0, t1, radius, mat),
                        "moving_sphere")

    def rectangle(self, plane, a, b, k, mat):
        return self._id(self.b.rectangle(self.handle, plane, a[0], a[1], b[0], b[1], k, mat), "rectangle")

    def cube(self, box_min, box_max, mat):
        return self._id(self.b.cube(self.handle, vec3(box_min), vec3(box_max), mat), "cube")

    def list(self, items=()):
        lid = self._id(self.b.list(self.handle), "list")
        for it in items:
            self.push(lid, it)
        return lid

    def push(self, lst, item):
        check(self.b.list_push(self.handle, lst, item), self.b, "list_push")

    def bvh_tree(self, lst):
        return self._id(self.b.bvh_tree(self.handle, lst), "bvh_tree")

    def translate(self, item, offset):
        return self._id(self.b.translate(self.handle, item, vec3(offset)), "translate")

    def rotate_y(self, item, degrees):
        return self._id(self.b.rotate_y(self.handle, item, degrees), "rotate_y")

    def constant_medium(self, boundary, density, tex_or_rgb):
        tex = tex_or_rgb if isinstance(tex_or_rgb, int) else self.solid(tex_or_rgb)
        return self._id(self.b.constant_medium(self.handle, boundary, density, tex), "constant_medium")

    def bounding_box(self, hittable, initial_time=0.0, final_time=1.0):
        """`Hittable::bounding_box(initial_time, final_time)` (hittable.rs:50): (min, max) as two arrays of 3, or None (an empty List)."""
        out = np.zeros(6)
        rc = check(self.b.hittable_bounds(self.handle, hittable, initial_time, final_time, out.ctypes.data_as(C.POINTER(C.c_double))), self.b,
                   "hittable_bounds")
        return (out[:3].copy(), out[3:].copy()) if rc == 1 else None

    def set_world(self, lst):
        check(self.b.scene_set_world(self.handle, lst), self.b, "scene_set_world")

    def set_bvh_builder(self, which):
        """abi.BVH_AUTO (default), abi.BVH_HOST_SAH, abi.BVH_DEVICE_LBVH or abi.BVH_DEVICE_SAH; before commit / build_named."""
        check(self.b.scene_set_bvh_builder(self.handle, int(which)), self.b, "scene_set_bvh_builder")

    def build_info(self):
        bi = abi.BuildInfo()
        check(self.b.scene_build_info(self.handle, C.byref(bi)), self.b, "scene_build_info")
        return bi

    def commit(self):
        check(self.b.scene_commit(self.handle), self.b, "scene_commit")

    # ---- the scenes.rs catalogue (host library rttnw_amd/host/scenes.cpp)
    def build_named(self, name, earth_rgba=None, param=0, builder_table=None):
        """Build one of the catalogue scenes into this (empty) scene, set world, commit."""
        assert self.scenes is not None, "no scenes library bound"
        table = builder_table if builder_table is not None else self.b.builder()
        setup = SceneSetup()
        if earth_rgba is not None:
            earth_rgba = np.ascontiguousarray(earth_rgba, dtype=np.uint8)
            h, w, _ = earth_rgba.shape
            ptr = earth_rgba.ctypes.data
        else:
            h = w = 0
            ptr = None
        rc = self.scenes.scenes_build(table, self.handle, name.encode(), self.seed, ptr, w, h, int(param),
                                      C.byref(setup))
        check(rc, self.b, "scenes_build(%s)" % name)
        self.setup = setup
        return setup


def camera_desc(lookfrom, lookat, vfov, aspect, aperture=0.0, view_up=(0.0, 1.0, 0.0), focus=10.0,
                open_time=0.0, close_time=1.0):
    """`CameraDescriptor` with the values main.rs:185-196 hard-codes as defaults."""
    return CameraDesc(vec3(lookfrom), vec3(lookat), vec3(view_up), vfov, aspect, aperture, focus,
                      open_time, close_time)


def make_params(width, height, spp, *, background=(0.0, 0.0, 0.0), seed=1, precision=abi.F64,
                quirks=abi.QUIRKS_REFERENCE, max_depth=50, t_min=1e-3, spp_chunk=0, tile_rank=0,
                tile_world=1, collect_counters=0, sample_begin=0):
    return Params(width, height, spp, max_depth, t_min, vec3(background), seed, precision, quirks,
                  spp_chunk, tile_rank, tile_world, collect_counters, sample_begin, 0)


def build(binding, scenes_lib, name, earth=None, param=0, seed=0x5EED0001, bvh=None):
    """One of the catalogue scenes (host/scenes.cpp, the scenes.rs mirror) built through `binding` and committed:
    (Scene, SceneSetup).  `bvh`: None = the library's default (abi.BVH_AUTO: host SAH below 100 000 leaves per tree, device SAH above),
    or abi.BVH_HOST_SAH, abi.BVH_DEVICE_LBVH, abi.BVH_DEVICE_SAH."""
    sc = Scene(binding, seed, scenes_binding=scenes_lib)
    if bvh is not None:
        sc.set_bvh_builder(bvh)
    setup = sc.build_named(name, earth_rgba=earth, param=param)
    return sc, setup


def params_for(setup, w, h, spp, **kw):
    """(CameraDesc, Params) of a catalogue scene at another size: the scene's own camera (main.rs:66-183) at aspect w / h."""
    cam = CameraDesc.from_buffer_copy(setup.camera)
    cam.aspect_ratio = w / h
    p = make_params(w, h, spp, background=tuple(setup.background), **kw)
    return cam, p


def load_earth():
    """Decode the earth map fixture (the reference's assets/earth.png, scenes.rs:129,303)."""
    import os
    from PIL import Image
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "earth.png")
    return np.asarray(Image.open(path).convert("RGBA"), dtype=np.uint8)
