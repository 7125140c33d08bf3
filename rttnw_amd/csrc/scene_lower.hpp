// scene_lower.hpp — host side of the boundary: records the scene graph the caller describes
// through the C ABI (the `Box/Arc<dyn Hittable>` graph of the reference, hittable.rs) and lowers
// it into the flat, index-linked arrays of rt_types.hpp.  Pure C++; no HIP in here.
#pragma once
#include "bvh_build.hpp"
#include "rt_types.hpp"

#include <cstdint>
#include <string>
#include <memory>
#include <utility>
#include <vector>

namespace rt {

// ---------------------------------------------------------------- recorded graph
struct GraphObj {
    enum Kind : int { TEX_SOLID_K, TEX_CHECKER_K, TEX_NOISE_K, TEX_IMAGE_K,
                      MAT_K,
                      SPHERE_K, MOVING_K, RECT_K, CUBE_K, LIST_K, BVH_K, TRANSLATE_K, ROTATE_K, MEDIUM_K };
    Kind kind;
    double v[10] = {0}; // numeric payload (meaning per kind, see scene_lower.cpp)
    int32_t a = -1, b = -1, c = -1; // object references / small ints
    std::vector<int32_t> items;     // LIST_K / BVH_K members
    bool consumed = false;          // a list moved into a BvhTree (BvhTree::from takes it by value)
};

struct SceneGraph {
    uint64_t seed = 0;
    std::vector<GraphObj> objs;
    std::vector<std::vector<uint8_t>> image_data; // RGBA8 per image texture (index = GraphObj::a)
    std::vector<uint32_t> image_w, image_h;
    int32_t world = -1;
    uint32_t n_noise = 0;

    bool is_texture(int32_t id) const;
    bool is_material(int32_t id) const;
    bool is_hittable(int32_t id) const;
};

// ---------------------------------------------------------------- lowered scene (f64 master copy)
// A vector whose resize() leaves the new elements UNINITIALISED: the big record arrays (10^6 spheres, materials, items) are sized
// once and then filled by host threads in parallel; value-initialising them first is a sequential pass over hundreds of
// megabytes of fresh pages (20-30 ms of a 10^6-sphere commit).  Element types are trivially copyable and destructible.
// Big blocks (>= 4 MiB) are 2 MiB-aligned and advised as transparent huge pages: the first touch of 250 MB of fresh 4 KiB pages by
// 32 host threads is 65 000 page faults taken one at a time.
void* big_block_alloc(size_t bytes, size_t align);
void big_block_free(void* p, size_t bytes, size_t align);
template <typename T> struct NoInitAlloc : std::allocator<T> {
    template <typename U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <typename U> NoInitAlloc(const NoInitAlloc<U>&) noexcept {}
    T* allocate(size_t n) { return static_cast<T*>(big_block_alloc(n * sizeof(T), alignof(T))); }
    void deallocate(T* p, size_t n) noexcept { big_block_free(p, n * sizeof(T), alignof(T)); }
    template <typename U> void construct(U*) noexcept {} // resize(): nothing — the caller writes every element
    template <typename U, typename A0, typename... A> void construct(U* p, A0&& a0, A&&... a) { ::new ((void*)p) U(std::forward<A0>(a0), std::forward<A>(a)...); }
};
template <typename T> using RecVec = std::vector<T, NoInitAlloc<T>>;

struct FlatScene {
    std::vector<BvhNode> nodes;   // the builders' binary trees (kept for inspection: rttnw_debug_scene_nodes); not uploaded
    std::vector<Bvh4Node> nodes4; // what the kernels walk: the same trees collapsed to 4-wide records
    int32_t top_root2 = 0;        // root of the top-level binary tree in `nodes`
    // Trees the DEVICE builder made stay on the device (bvh_build.hpp DeviceTree): the scene's node array is the n_host4
    // host-built records of `nodes4` followed by the device trees in order (tree.base4 = where each starts; likewise base2 /
    // n_host2 for the binary records).  `nodes4` / `nodes` hold only their host-built part until someone asks to look at
    // the rest (materialize_host_nodes(), scene_handle.hpp: inspection calls, a second device).
    std::vector<DeviceTree> device_trees;
    uint32_t n_host4 = 0, n_host2 = 0;
    uint32_t total_nodes4() const { uint32_t n = n_host4; for (const auto& t : device_trees) n += t.count4; return n; }
    uint32_t total_nodes2() const { uint32_t n = n_host2; for (const auto& t : device_trees) n += t.count2; return n; }
    RecVec<SphereRec<double>> spheres;
    RecVec<int32_t> sphere_mat;
    RecVec<int32_t> sphere_seq;
    RecVec<MovingSphereRec<double>> moving;
    RecVec<RectRec<double>> rects;
    RecVec<BoxRec<double>> boxes;
    std::vector<InstanceRec<double>> insts;
    std::vector<MediumRec<double>> media;
    std::vector<int32_t> medium_refs; // boundary primitives of the media (MediumRec::b_first / b_count)
    double time0 = 0.0, time1 = 1.0;  // shutter interval the moving spheres' boxes were built for (BvhTree::from_time, hittable.rs:261)
    RecVec<MaterialRec<double>> mats;
    RecVec<TextureRec<double>> texs;
    std::vector<ImageRec> images;
    std::vector<uint32_t> texels;
    std::vector<double> perlin_vec;   // [n][256][3]
    std::vector<uint8_t> perlin_perm; // [n][3][256]
    bool needs_general = false;       // a chain of more than FAST_INSTANCE_OPS wrappers, a multi-member medium boundary or a medium
                                      //   inside a transformed group: rendered by the GENERAL instantiation of the kernels
    int32_t top_root = 0;
    uint32_t stack_depth = 4;         // entries a lane's traversal stack can need (exact bound for the 4-wide trees)
    uint32_t n_prims_in_bvh = 0;
    bool has_instance_leaves = false; // some tree holds an instance leaf at all (a wrapped group, or a single wrapped record tested in place)
    bool walk_changes_frames = false; // some tree holds an instance with a tree of its own (a wrapped group the walk enters): false = the walk never leaves world space
    uint32_t n_world_copies = 0;      // spheres of transformed groups that the walk tests as world-space copies in the top tree (scene_lower.cpp collect)
    // A LEAN scene: no MovingSphere, no ConstantMedium, every material a solid colour (lowered into its record: tex < 0) — the kernels have
    // instantiations without the code of any of them (rt_core.hpp SHAPES_NONE_NT / SHAPES_SINGLE_NT).
    // (decided ONCE, at the end of lower_scene: spheres_1m has 10^6 materials, and the render path asks this before every launch)
    bool is_lean = false;
    bool lean() const { return is_lean; }
    void decide_lean() {
        is_lean = moving.empty() && media.empty();
        for (size_t i = 0; is_lean && i < mats.size(); ++i) is_lean = mats[i].tex < 0;
        // A big cloud whose spheres were created material-then-sphere, in order (spheres_1m: 10^6 of each), keeps that order through the lowering
        // when its tree is built on the device (one record per leaf, records in creation order): sphere i's material slot holds i.  The kernels
        // then take the index itself (SceneView::sphere_mat == nullptr, rt_core.hpp make_record) — on a tree that lives in HBM the slot is an
        // L2-miss line per hit of its own (tests/hostsim/cache_model.hpp: 4.4 of 92 per sample).  Small scenes keep their slots (LDS-staged).
        sphere_mat_is_index = spheres.size() >= 65536;
        for (size_t i = 0; sphere_mat_is_index && i < sphere_mat.size(); ++i) sphere_mat_is_index = sphere_mat[i] == int32_t(i);
        // ... and where a builder has reordered the records (the host SAH build emits them in leaf order), a LEAN cloud gets the order back the
        // other way round: the materials follow the spheres — material i = sphere i's, the scene's own materials behind them (the other kinds'
        // references move along).  Nothing but indices changes; 64 B per sphere.
        if (!sphere_mat_is_index && is_lean && spheres.size() >= 65536 && insts.empty() && mats.size() < (size_t(1) << 28)) {
            bool plain = true; // (no slot is a world-space copy's way home, none asks for (u, v))
            for (size_t i = 0; plain && i < sphere_mat.size(); ++i) plain = (sphere_mat[i] & ~MAT_INDEX_MASK) == 0 && (sphere_mat[i] & MAT_HOME_FLAG) == 0;
            if (plain) {
                const size_t ns = spheres.size(), nm = mats.size();
                RecVec<MaterialRec<double>> moved;
                moved.resize(ns + nm);
                for (size_t i = 0; i < ns; ++i) moved[i] = mats[size_t(sphere_mat[i])];
                for (size_t i = 0; i < nm; ++i) moved[ns + i] = mats[i];
                mats.swap(moved);
                for (size_t i = 0; i < ns; ++i) sphere_mat[i] = int32_t(i);
                for (auto& r : rects) r.mat += int32_t(ns);   // (the flag bits above the index stay where they are)
                for (auto& b : boxes) b.mat += int32_t(ns);
                sphere_mat_is_index = true;
            }
        }
    }
    bool sphere_mat_is_index = false;
};

// Lower `g` into `out`.  Returns 0 or a negative rttnw_status; `err` receives a message.  `device` (optional)
// replaces the host binned-SAH build for every tree of two or more leaves by the device builder (bvh_build.hpp); those
// trees stay on the device.
// `world_spheres`: 1 / 0 = test the spheres of transformed groups as world-space copies in the top tree / leave them in their groups' trees, in
// the frame the reference tests them in; 2 = what RTTNW_F64_STRICT renders: the copies' world-space boxes in the top tree (culling never shapes
// a result), the sphere test itself in the group's frame, through the group's wrappers as in hittable.rs:599-606,686-699 (leaf kind
// PRIM_SPHERE_WC); -1 = the default (1, or RTTNW_WORLD_SPHERES).
int lower_scene(const SceneGraph& g, FlatScene& out, std::string& err, const DeviceBvhApi* device = nullptr, double time0 = 0.0,
                double time1 = 1.0, int world_spheres = -1);

// Camera::new — camera.rs:32-61 (computed once on the host, in f64)
void make_camera(const double lookfrom[3], const double lookat[3], const double view_up[3], double vfov_deg,
                 double aspect, double aperture, double focus_distance, double open_time, double close_time,
                 CameraRec<double>& out);

// Scene-construction stream of the library (Perlin tables): DESIGN.md "RNG"
struct SceneRng {
    uint64_t s;
    SceneRng(uint64_t seed, uint64_t stream);
    uint64_t next_u64();
    double next_f64();
    double range(double a, double b) { return a + (b - a) * next_f64(); }
    uint32_t below(uint32_t n) { return uint32_t(next_f64() * double(n)); }
};

} // namespace rt
