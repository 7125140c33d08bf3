// bvh_build.hpp — device-side BVH construction (the step BEFORE the hot path).
//
// The reference builds its tree on the CPU, recursively, one random axis per level and a full sort per level
// (BvhTree::from / build, hittable.rs:300-353); the library's default is the host binned-SAH builder of
// scene_lower.cpp.  This is the alternative for large or frequently rebuilt scenes: a linear BVH (Morton order +
// Karras 2012 hierarchy + bottom-up box fit) built by HIP kernels, about two orders of magnitude faster to
// build, somewhat slower to traverse.  Which tree is used never changes a result: the closest hit does not
// depend on topology and exact ties are resolved by the records' sequence numbers (rt_types.hpp).
#pragma once
#include "rt_types.hpp"

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace rt {

// One leaf of the tree to build: its bounds (f32, already rounded outward) and its child-slot code
// (make_leaf(kind, 1, record index)).
struct BuildPrim {
    float lo[3];
    int32_t leaf;
    float hi[3];
    int32_t pad;
};
static_assert(sizeof(BuildPrim) == 32, "BuildPrim must be 32 bytes");

// Build a tree over prims (size >= 2).  Appends its size-1 inner nodes to `nodes` (child slots already offset),
// sets `root` (index into `nodes`) and `levels` (inner-node levels on the longest root-to-leaf path).
using BvhBuilder = std::function<int(const std::vector<BuildPrim>& prims, std::vector<BvhNode>& nodes, int32_t& root,
                                     uint32_t& levels, std::string& err)>;

// The HIP implementation (bvh_build.hip); runs on the current device, synchronous.  `kernel_ms` (optional)
// accumulates the device time of the build kernels + sort.
int lbvh_build_device(const std::vector<BuildPrim>& prims, std::vector<BvhNode>& nodes, int32_t& root, uint32_t& levels,
                      double* kernel_ms, std::string& err);

} // namespace rt
