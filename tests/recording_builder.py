"""A third implementation of the scene-building half of the boundary (include/rttnw_hip.h rttnw_builder_api): it builds
nothing and RECORDS every call.  Test infrastructure: the catalogue (rttnw_amd/host/scenes.cpp) drives the product and the
oracle through the same table, so a wrong constant there is invisible to every product-vs-oracle comparison; driven
through this table instead, the catalogue's calls can be read back and held against numbers typed from the reference's
src/scenes.rs (tests/test_scene_catalogue_independent.py)."""
import ctypes as C

from rttnw_amd import abi


class BuilderApi(C.Structure):
    """rttnw_builder_api: the entry points in header order."""
    _fields_ = [(name, C.CFUNCTYPE(restype, *argtypes)) for name, restype, argtypes in abi.BUILDER_FUNCS]


def _v3(p):
    return (p[0], p[1], p[2])


class Recorder:
    """objs[id] = (kind, fields...); ids are what the table hands back, in creation order."""

    def __init__(self):
        self.objs = []
        self.world = None
        self.committed = False
        self._cbs = []
        self.api = BuilderApi()
        for name, restype, argtypes in abi.BUILDER_FUNCS:
            fn = getattr(self, "_" + name)
            cb = C.CFUNCTYPE(restype, *[C.POINTER(C.c_double) if t is abi.c_double3 else t for t in argtypes])(fn)
            self._cbs.append(cb)
            setattr(self.api, name, C.cast(cb, dict(BuilderApi._fields_)[name]))

    def table(self):
        return C.addressof(self.api)

    def _new(self, *rec):
        self.objs.append(rec)
        return len(self.objs) - 1

    # ---- the table
    def _scene_create(self, seed, out):
        return 0

    def _scene_destroy(self, s):
        return None

    def _tex_solid(self, s, r, g, b):
        return self._new("solid", (r, g, b))

    def _tex_checker(self, s, odd, even):
        return self._new("checker", odd, even)

    def _tex_noise(self, s, scale):
        return self._new("noise", scale)

    def _tex_image_rgba8(self, s, ptr, w, h):
        return self._new("image", bool(ptr), w, h)

    def _mat_lambertian(self, s, tex):
        return self._new("lambertian", tex)

    def _mat_metal(self, s, r, g, b, fuzz):
        return self._new("metal", (r, g, b), fuzz)

    def _mat_dielectric(self, s, ri):
        return self._new("dielectric", ri)

    def _mat_diffuse_light(self, s, tex):
        return self._new("diffuse_light", tex)

    def _mat_isotropic(self, s, tex):
        return self._new("isotropic", tex)

    def _sphere(self, s, c, r, mat):
        return self._new("sphere", _v3(c), r, mat)

    def _moving_sphere(self, s, c0, c1, t0, t1, r, mat):
        return self._new("moving_sphere", _v3(c0), _v3(c1), t0, t1, r, mat)

    def _rectangle(self, s, plane, a0, a1, b0, b1, k, mat):
        return self._new("rectangle", plane, (a0, a1), (b0, b1), k, mat)

    def _cube(self, s, mn, mx, mat):
        return self._new("cube", _v3(mn), _v3(mx), mat)

    def _list(self, s):
        return self._new("list", [])

    def _list_push(self, s, lst, item):
        self.objs[lst][1].append(item)
        return 0

    def _bvh_tree(self, s, lst):
        return self._new("bvh_tree", lst)

    def _translate(self, s, item, off):
        return self._new("translate", item, _v3(off))

    def _rotate_y(self, s, item, deg):
        return self._new("rotate_y", item, deg)

    def _constant_medium(self, s, boundary, density, tex):
        return self._new("constant_medium", boundary, density, tex)

    def _scene_set_world(self, s, lst):
        self.world = lst
        return 0

    def _scene_commit(self, s):
        self.committed = True
        return 0

    def _last_error(self):
        return b""

    # ---- reading the record back as nested tuples (ids resolved), the form the expectations are written in
    def resolve(self, i):
        o = self.objs[i]
        k = o[0]
        if k in ("solid", "noise", "image", "dielectric"):
            return o
        if k == "checker":
            return (k, self.resolve(o[1]), self.resolve(o[2]))
        if k in ("lambertian", "diffuse_light", "isotropic"):
            return (k, self.resolve(o[1]))
        if k == "metal":
            return o
        if k == "sphere":
            return (k, o[1], o[2], self.resolve(o[3]))
        if k == "moving_sphere":
            return (k, o[1], o[2], o[3], o[4], o[5], self.resolve(o[6]))
        if k == "rectangle":
            return (k, o[1], o[2], o[3], o[4], self.resolve(o[5]))
        if k == "cube":
            return (k, o[1], o[2], self.resolve(o[3]))
        if k == "list":
            return (k, [self.resolve(j) for j in o[1]])
        if k == "bvh_tree":
            return (k, self.resolve(o[1]))
        if k in ("translate", "rotate_y"):
            return (k, self.resolve(o[1]), o[2])
        if k == "constant_medium":
            return (k, self.resolve(o[1]), o[2], self.resolve(o[3]))
        raise AssertionError(k)


def record(scenes_lib, name, earth=None, param=0, seed=0x5EED0001):
    """The catalogue's scene `name` as the calls it makes: (world as nested tuples, SceneSetup, Recorder)."""
    import numpy as np
    rec = Recorder()
    setup = abi.SceneSetup()
    ptr, w, h = None, 0, 0
    if earth is not None:
        earth = np.ascontiguousarray(earth, dtype=np.uint8)
        h, w, _ = earth.shape
        ptr = earth.ctypes.data
    token = C.c_int(0)   # the catalogue only passes the handle through
    rc = scenes_lib.scenes_build(rec.table(), C.addressof(token), name.encode(), seed, ptr, w, h, int(param), C.byref(setup))
    assert rc == 0 and rec.committed and rec.world is not None, (name, rc)
    return rec.resolve(rec.world), setup, rec
