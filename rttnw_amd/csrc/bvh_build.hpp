// bvh_build.hpp — device-side BVH construction (the step BEFORE the hot path).
//
// The reference builds its tree on the CPU, recursively, one random axis per level and a full sort per level
// (BvhTree::from / build, hittable.rs:300-353); the library's default is the host binned-SAH builder of
// scene_lower.cpp.  These are the alternatives for large or frequently rebuilt scenes, built by HIP kernels: a linear BVH
// (Morton order + Karras 2012 hierarchy + bottom-up box fit: two orders of magnitude faster to build than the host's, 4-7 %
// slower to traverse) and the binned-SAH tree (level-synchronous kernels: ten times faster to build, traverses like the host's).  Which tree is used never changes a result: the closest hit does not
// depend on topology and exact ties are resolved by the records' sequence numbers (rt_types.hpp).
#pragma once
#include "rt_types.hpp"

#include <cstdint>
#include <functional>
#include <memory>
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

// A tree the device builder made and LEFT ON THE DEVICE ("the step before the path" hands its result
// to the path without a round trip through the host): the 4-wide records the kernels walk (rt_types.hpp Bvh4Node, root =
// record 0, records in LEVEL order — the inner children of a record side by side: bvh_build.hip) and, for inspection, the builder's binary
// tree (BvhNode, pre-order, root = 0).  Child indices are LOCAL to the tree until `rebase` has added the tree's place in
// the scene's node array.  The buffers are owned by the shared pointers (freed on the device they live on).
struct DeviceTree {
    std::shared_ptr<void> nodes4, nodes2;
    uint32_t count4 = 0, count2 = 0; // records in nodes4 / nodes2
    uint32_t need = 0;               // stack entries a walk of this tree can have pending (exact, as scene_lower.cpp collapse4 counts)
    uint32_t levels = 0;             // inner levels of the binary tree
    uint32_t base4 = 0, base2 = 0;   // where the tree sits in the scene's 4-wide / binary node arrays (set by the lowering)
    int device = -1;
};

// The device builder as the lowering sees it (scene_lower.cpp is plain C++: it calls through these).
struct DeviceBvhApi {
    // build a tree over prims[0, n) (n >= 2) on the current device; centroid_bounds (optional): lo[3], hi[3] of the leaves' box
    // centres as the builders compute them (0.5f * (lo + hi) in f32) — a caller that walks the leaves anyway saves the builder a pass
    std::function<int(const BuildPrim* prims, size_t n, const float* centroid_bounds, DeviceTree& out, std::string& err)> build;
    // add base4 to the inner child indices of the tree's 4-wide records (and remember base4 / base2 in the tree)
    std::function<int(DeviceTree& tree, uint32_t base4, uint32_t base2, std::string& err)> rebase;
    // trees of fewer leaves are built on the host (RTTNW_BVH_AUTO: include/rttnw_hip.h RTTNW_BVH_AUTO_DEVICE_LEAVES; the explicit device builders: 2)
    size_t min_leaves = 2;
};
// Copy a device tree's records to the host (inspection, other devices): out4 / out2 may be null.
int device_tree_download(const DeviceTree& tree, Bvh4Node* out4, BvhNode* out2, std::string& err);

// The HIP implementation (bvh_build.hip); runs on the current device, synchronous.  `kernel_ms` (optional) accumulates the
// device time of the build kernels + sort + collapse.
// (`sah`: the binned-SAH hierarchy instead of the linear one, RTTNW_BVH_DEVICE_SAH)
int lbvh_build_device_tree(const BuildPrim* prims, size_t n, const float* centroid_bounds, bool sah, DeviceTree& out, double* kernel_ms, std::string& err);
int device_tree_rebase(DeviceTree& tree, uint32_t base4, uint32_t base2, std::string& err);

// The scene's f32 4-wide records (on the current device) -> the quantised records the f64 decoupled kernel walks (rt_types.hpp
// Bvh4QNode, bvh_quant.hpp), index for index, into d_out[0, n).  Synchronous.
int quant4_build_device(const Bvh4Node* d_nodes4, uint32_t n, Bvh4QNode* d_out, std::string& err);

// The INTERLEAVED buffer of the decoupled kernels (round 6, big clouds: FlatScene::sphere_mat_is_index): a quantised node record is followed, in ONE
// buffer, by the sphere records of its sphere leaves — a record that has any starts on a 128-byte line, so that the walk that has fetched the node
// finds its first two (f64; all four in f32) spheres in the line it already holds.  Positions are counted in 16-byte UNITS: `noff[i]` = where record
// i starts (the host lays the records out from the per-record sphere counts of interleave_count_device).  A node is addressed as buffer +
// (noff / 4) x 64 bytes, a sphere as buffer + index x sphere_bytes: child slots and leaf bits of the records are rewritten to those numbers; the
// spheres' sequence numbers and materials (slot i holds i: material index = sphere index) move to the same, sparse, indices.  Layout only.
int interleave_count_device(const Bvh4Node* d_nodes4, uint32_t n, uint8_t* d_sphere_count, std::string& err);
struct InterleaveArgs {
    const Bvh4Node* nodes4; uint32_t n4;
    const uint32_t* noff;                       // [n4], 16-byte units
    const void* spheres; uint32_t sphere_bytes;  // 16 (f32) / 32 (f64)
    const int32_t* sphere_seq;
    const void* mats; uint32_t mat_bytes;        // sizeof(MaterialRec<R>)
    void* buffer;                                // out: the interleaved node + sphere buffer
    int32_t* seq_out;                            // out: sequence numbers by the new sphere index
    void* mats_out;                              // out: materials by the new sphere index
};
int interleave_build_device(const InterleaveArgs& a, std::string& err);

} // namespace rt
