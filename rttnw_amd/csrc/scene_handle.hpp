// scene_handle.hpp — what `rttnw_scene*` points at.  The recording half (capi_builder.cpp) is plain
// C++; the device half (render.hip) hangs its state off `device`.
#pragma once
#include "scene_lower.hpp"

#include <string>

namespace rt {
struct DeviceState; // defined in render.hip
// Called by rttnw_scene_commit after lowering; uploads to the current HIP device.
int device_commit(struct ::rttnw_scene* s, std::string& err);
void device_release(DeviceState* d);
void set_last_error(const std::string& msg);
} // namespace rt

struct rttnw_scene {
    rt::SceneGraph graph;
    rt::FlatScene flat;
    bool committed = false;
    uint32_t n_media = 0;
    rt::DeviceState* device = nullptr;
};
