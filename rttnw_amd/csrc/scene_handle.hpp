// scene_handle.hpp — what `rttnw_scene*` points at.  The recording half (capi_builder.cpp) is plain
// C++; the device half (render_api.cpp, render_f32.hip / render_f64.hip) hangs its state off `device`.
#pragma once
#include "scene_lower.hpp"

#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace rt {
struct DeviceState; // defined in render_common.hpp
// Called by rttnw_scene_commit after lowering; uploads to the current HIP device.
int device_commit(struct ::rttnw_scene* s, std::string& err);
void device_release(DeviceState* d);
// The device BVH builder bound to scene `s` (accumulates its kernel time in s->build_kernel_ms); fails without a
// usable HIP device.  Defined in render_api.cpp (and as a failing stub in the host-only test build).
int device_bvh_builder(struct ::rttnw_scene* s, DeviceBvhApi& out, std::string& err);
// Bring the records of the device-built trees into f.nodes4 / f.nodes (behind their host-built part), once; a no-op for scenes
// without device trees.  Inspection calls and the upload to a second device use it; the render path on the building device
// never does.  Defined in render_api.cpp (a no-op stub in the host-only test build, which has no device trees).
int materialize_host_nodes(FlatScene& f, std::string& err);
void set_last_error(const std::string& msg);
} // namespace rt

struct rttnw_scene {
    rt::SceneGraph graph;
    rt::FlatScene flat;
    // The same graph lowered with the spheres of transformed groups LEFT in their groups' trees — every object tested in the frame the
    // reference tests it in: what RTTNW_F64_STRICT renders (render_api.cpp reference_frame_scene: made at the first such render of a scene
    // whose `flat` holds world-space copies; null otherwise).
    std::unique_ptr<rt::FlatScene> flat_ref;
    bool committed = false;
    uint32_t n_media = 0;
    uint32_t bvh_builder = 3;      // RTTNW_BVH_* (RTTNW_BVH_AUTO)
    double lower_ms = 0;           // host wall time of rttnw_scene_commit's lowering (BVH builds included)
    double build_kernel_ms = 0;    // device time of the BVH build kernels (device builder only)
    rt::DeviceState* device = nullptr;              // state on the device that was current at commit
    std::vector<rt::DeviceState*> more_devices;     // states on further devices (rttnw_render_multi), created on first use
    std::mutex rebuild_mutex;                       // render entry points: the one post-commit change of `flat` (a wider shutter, render.hip validate)
};
