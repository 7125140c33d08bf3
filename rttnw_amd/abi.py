"""ctypes declarations of the C ABI in include/rttnw_hip.h and include/rttnw_scenes.h.

Pure declarations: structures, status codes and the signature table used to bind a shared
library that implements the boundary.  No library is loaded here.
"""
import ctypes as C

RTTNW_OK = 0
ERR_NAMES = {-1: "RTTNW_ERR_INVALID", -2: "RTTNW_ERR_STATE", -3: "RTTNW_ERR_UNSUPPORTED",
             -4: "RTTNW_ERR_HIP", -5: "RTTNW_ERR_NOMEM"}

XY, XZ, YZ = 0, 1, 2
F64, F32, F64_STRICT = 0, 1, 2
BVH_HOST_SAH, BVH_DEVICE_LBVH, BVH_DEVICE_SAH, BVH_AUTO = 0, 1, 2, 3
ABI_VERSION = 3  # include/rttnw_hip.h RTTNW_ABI_VERSION
QUIRK_YROTATE_BACKROT = 1
QUIRKS_REFERENCE = QUIRK_YROTATE_BACKROT

c_id = C.c_int32
c_double3 = C.c_double * 3
scene_p = C.c_void_p


class CameraDesc(C.Structure):
    """`CameraDescriptor` — reference src/math/camera.rs:5-15."""
    _fields_ = [("lookfrom", c_double3), ("lookat", c_double3), ("view_up", c_double3),
                ("vertical_fov", C.c_double), ("aspect_ratio", C.c_double), ("aperture", C.c_double),
                ("focus_distance", C.c_double), ("open_time", C.c_double), ("close_time", C.c_double)]


class Params(C.Structure):
    """What `render()` hard-codes or takes as arguments — reference src/main.rs:58,184-197,216,33."""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("max_depth", C.c_uint32),
                ("t_min", C.c_double), ("background", c_double3), ("seed", C.c_uint64),
                ("precision", C.c_uint32), ("quirks", C.c_uint32), ("spp_chunk", C.c_uint32),
                ("tile_rank", C.c_uint32), ("tile_world", C.c_uint32), ("collect_counters", C.c_uint32),
                ("sample_begin", C.c_uint32), ("reserved0", C.c_uint32)]


class BuildInfo(C.Structure):
    _fields_ = [("builder", C.c_uint32), ("n_nodes", C.c_uint32), ("n_prims", C.c_uint32), ("stack_depth", C.c_uint32),
                ("lower_ms", C.c_double), ("device_ms", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("nodes_visited", C.c_uint64),
                ("prims_tested", C.c_uint64), ("texel_fetches", C.c_uint64), ("kernel_ms", C.c_double),
                ("n_nodes", C.c_uint32), ("n_prims", C.c_uint32), ("scene_bytes", C.c_uint32),
                ("reserved", C.c_uint32)]


class TileLayout(C.Structure):
    _fields_ = [("tiles_x", C.c_uint32), ("tiles_y", C.c_uint32), ("n_tiles", C.c_uint32),
                ("tiles_per_rank", C.c_uint32), ("pixels_per_rank", C.c_uint32)]


class SceneSetup(C.Structure):
    """Per-scene camera/size table entry — reference src/main.rs:66-183."""
    _fields_ = [("camera", CameraDesc), ("background", c_double3), ("width", C.c_uint32),
                ("height", C.c_uint32), ("spp", C.c_uint32), ("scene_number", C.c_uint32)]


_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)

# (name, restype, argtypes) of the scene-building entry points, in rttnw_builder_api order.
BUILDER_FUNCS = [
    ("scene_create", C.c_int, [C.c_uint64, C.POINTER(scene_p)]),
    ("scene_destroy", None, [scene_p]),
    ("tex_solid", c_id, [scene_p, C.c_double, C.c_double, C.c_double]),
    ("tex_checker", c_id, [scene_p, c_id, c_id]),
    ("tex_noise", c_id, [scene_p, C.c_double]),
    ("tex_image_rgba8", c_id, [scene_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    ("mat_lambertian", c_id, [scene_p, c_id]),
    ("mat_metal", c_id, [scene_p, C.c_double, C.c_double, C.c_double, C.c_double]),
    ("mat_dielectric", c_id, [scene_p, C.c_double]),
    ("mat_diffuse_light", c_id, [scene_p, c_id]),
    ("mat_isotropic", c_id, [scene_p, c_id]),
    ("sphere", c_id, [scene_p, c_double3, C.c_double, c_id]),
    ("moving_sphere", c_id, [scene_p, c_double3, c_double3, C.c_double, C.c_double, C.c_double, c_id]),
    ("rectangle", c_id, [scene_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, c_id]),
    ("cube", c_id, [scene_p, c_double3, c_double3, c_id]),
    ("list", c_id, [scene_p]),
    ("list_push", C.c_int, [scene_p, c_id, c_id]),
    ("bvh_tree", c_id, [scene_p, c_id]),
    ("translate", c_id, [scene_p, c_id, c_double3]),
    ("rotate_y", c_id, [scene_p, c_id, C.c_double]),
    ("constant_medium", c_id, [scene_p, c_id, C.c_double, c_id]),
    ("scene_set_world", C.c_int, [scene_p, c_id]),
    ("scene_commit", C.c_int, [scene_p]),
    ("last_error", C.c_char_p, []),
]

# render / introspection entry points of the product library (prefix rttnw_)
PRODUCT_FUNCS = [
    ("tile_layout_get", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(TileLayout)]),
    ("render", C.c_int, [scene_p, C.POINTER(CameraDesc), C.POINTER(Params), C.c_void_p, C.c_void_p,
                         C.POINTER(Stats)]),
    ("render_multi", C.c_int, [scene_p, C.POINTER(CameraDesc), C.POINTER(Params), C.c_uint32, C.POINTER(C.c_int32), C.c_void_p,
                               C.c_void_p, C.c_void_p]),
    ("render_tiles_device", C.c_int, [scene_p, C.POINTER(CameraDesc), C.POINTER(Params), C.c_void_p,
                                      C.c_void_p, C.POINTER(Stats)]),
    ("untile_device", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p]),
    ("abi_version", C.c_int, []),
    ("shutdown", None, []),
    ("device_count", C.c_int, []),
    ("scene_info", C.c_int, [scene_p, C.POINTER(Stats)]),
    ("scene_set_bvh_builder", C.c_int, [scene_p, C.c_uint32]),
    ("scene_build_info", C.c_int, [scene_p, C.POINTER(BuildInfo)]),
    ("hittable_bounds", C.c_int, [scene_p, c_id, C.c_double, C.c_double, _dp]),
    ("debug_scene_nodes", C.c_int, [scene_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_int32)]),
    ("debug_scene_nodes4", C.c_int, [scene_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_int32)]),
    ("builder", C.c_void_p, []),
    ("debug_probe_path", C.c_int, [scene_p, C.POINTER(CameraDesc), C.POINTER(Params), C.c_uint32, C.c_uint32,
                                   C.c_uint32, C.c_void_p, C.c_uint32]),
]

SCENES_FUNCS = [
    ("rttnw_scenes_build", C.c_int, [C.c_void_p, scene_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint32,
                                     C.c_uint32, C.c_uint32, C.POINTER(SceneSetup)]),
    ("rttnw_scenes_name", C.c_char_p, [C.c_uint32]),
    ("rttnw_scenes_rng_f64", None, [C.c_uint64, C.c_uint64, C.c_uint32, _dp]),
]


class Binding:
    """Functions of one implementation of the boundary, bound by prefix (`rttnw_` = product)."""

    def __init__(self, lib, prefix, funcs):
        self.lib = lib
        self.prefix = prefix
        for name, restype, argtypes in funcs:
            fn = getattr(lib, prefix + name)
            fn.restype = restype
            fn.argtypes = argtypes
            setattr(self, name.replace("rttnw_", ""), fn)

    def add(self, funcs, prefix=None):
        for name, restype, argtypes in funcs:
            fn = getattr(self.lib, (self.prefix if prefix is None else prefix) + name)
            fn.restype = restype
            fn.argtypes = argtypes
            setattr(self, name.replace("rttnw_", ""), fn)


def exported_symbols(prefix="rttnw_"):
    """Every symbol include/rttnw_hip.h declares (used by the CPU-side export test)."""
    return [prefix + n for n, _, _ in BUILDER_FUNCS] + [prefix + n for n, _, _ in PRODUCT_FUNCS]


def vec3(x, y=None, z=None):
    if y is None:
        x, y, z = x
    return c_double3(float(x), float(y), float(z))


class RttnwError(RuntimeError):
    pass


def check(rc, binding, what=""):
    if rc is not None and rc < 0:
        msg = binding.last_error()
        raise RttnwError("%s failed: %s (%s)" % (what or "call", ERR_NAMES.get(rc, rc),
                                                 msg.decode() if msg else ""))
    return rc
