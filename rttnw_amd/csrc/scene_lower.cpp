// scene_lower.cpp — flatten the recorded scene graph and build the flat BVHs (host, C++).
//
// Reference shapes and what they lower to:
//   world List (main.rs:47-55)                 -> one top-level BVH over every solid in it
//   List / BvhTree::from(list) (hittable.rs:134,254) -> flattened into the enclosing BVH (grouping only;
//                                                 closest-hit results do not depend on topology, Q12)
//   Sphere / MovingSphere / Rectangle / Cube   -> one record each in its own array (Cube = ONE box
//                                                 record whose hit reproduces its six rectangles)
//   x.rotate_y(a).translate(v) chains (hittable.rs:51-65) -> an instance: ops + a sub-BVH over x
//   ConstantMedium (hittable.rs:724-801)       -> a medium record evaluated after the BVH walk
// The reference's own builder (hittable.rs:265-321: random axis, whole-vector re-sort, O(n^2 log n))
// is NOT reproduced; this one is a binned-SAH build producing 64-byte two-box nodes.
#include "scene_lower.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <array>
#include <atomic>
#include <mutex>
#include <thread>
#include <sys/mman.h>
#include <cstdlib>
#include <new>

namespace rt {

namespace {
constexpr int ERR_INVALID = -1, ERR_UNSUPPORTED = -3;
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr uint64_t GAMMA = 0x9E3779B97F4A7C15ull;
uint64_t mix64h(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
} // namespace

SceneRng::SceneRng(uint64_t seed, uint64_t stream)
    : s(mix64h(seed + GAMMA) ^ mix64h((stream + 1) * 0xD1B54A32D192ED03ull)) {}
uint64_t SceneRng::next_u64() { s += GAMMA; return mix64h(s); }
double SceneRng::next_f64() { return double(next_u64() >> 11) * (1.0 / 9007199254740992.0); }

bool SceneGraph::is_texture(int32_t id) const {
    return id >= 0 && size_t(id) < objs.size() && objs[id].kind <= GraphObj::TEX_IMAGE_K;
}
bool SceneGraph::is_material(int32_t id) const {
    return id >= 0 && size_t(id) < objs.size() && objs[id].kind == GraphObj::MAT_K;
}
bool SceneGraph::is_hittable(int32_t id) const {
    return id >= 0 && size_t(id) < objs.size() && objs[id].kind >= GraphObj::SPHERE_K && !objs[id].consumed;
}

// Camera::new — camera.rs:32-61
void make_camera(const double lookfrom[3], const double lookat[3], const double vup[3], double vfov_deg, double aspect,
                 double aperture, double focus, double open_time, double close_time, CameraRec<double>& c) {
    auto unit3 = [](double* v) {
        double k = 1.0 / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        v[0] *= k; v[1] *= k; v[2] *= k;
    };
    auto cross3 = [](const double* a, const double* b, double* o) {
        o[0] = a[1] * b[2] - a[2] * b[1];
        o[1] = -(a[0] * b[2] - a[2] * b[0]);
        o[2] = a[0] * b[1] - a[1] * b[0];
    };
    c.lens_radius = aperture / 2.0;
    double theta = vfov_deg * kPi / 180.0;
    double half_height = std::tan(theta / 2.0);
    double half_width = aspect * half_height;
    double w[3] = {lookfrom[0] - lookat[0], lookfrom[1] - lookat[1], lookfrom[2] - lookat[2]};
    unit3(w);
    double u[3];
    cross3(vup, w, u);
    unit3(u);
    double v[3];
    cross3(w, u, v);
    for (int k = 0; k < 3; ++k) {
        c.origin[k] = lookfrom[k];
        c.u[k] = u[k];
        c.v[k] = v[k];
        // origin - half_width*focus*u - half_height*focus*v - focus*w  (left to right, camera.rs:41-44)
        c.lower_left_corner[k] = ((lookfrom[k] - half_width * focus * u[k]) - half_height * focus * v[k]) - focus * w[k];
        c.horizontal[k] = 2.0 * half_width * focus * u[k];
        c.vertical[k] = 2.0 * half_height * focus * v[k];
    }
    c.open_time = open_time;
    c.close_time = close_time;
}

constexpr size_t BIG_BLOCK = size_t(4) << 20;
void* big_block_alloc(size_t bytes, size_t align) {
    if (bytes < BIG_BLOCK) return ::operator new(bytes ? bytes : 1, std::align_val_t(align)); // (record types are aligned to 16 or 32 bytes)
    void* p = nullptr;
    if (posix_memalign(&p, size_t(2) << 20, bytes) != 0 || !p) throw std::bad_alloc();
    (void)madvise(p, bytes, MADV_HUGEPAGE); // advice only: where transparent huge pages are off, nothing changes
    return p;
}
void big_block_free(void* p, size_t bytes, size_t align) {
    if (bytes < BIG_BLOCK) ::operator delete(p, std::align_val_t(align));
    else std::free(p);
}

namespace {

// Host threads for the two loops of the lowering that touch every object of a big flat list (10^6 graph objects are 130 MB
// to read): chunks of `grain` indices, at most 32 threads; small ranges run inline.
template <typename F> void parallel_for(size_t n, size_t grain, F&& fn) {
    const size_t hw = std::max<size_t>(1, std::thread::hardware_concurrency());
    const size_t n_threads = std::min<size_t>(std::min<size_t>(hw, 32), (n + grain - 1) / grain);
    if (n_threads <= 1) { fn(size_t(0), n); return; }
    const size_t per = (n + n_threads - 1) / n_threads;
    std::vector<std::thread> th;
    for (size_t t = 1; t < n_threads; ++t) th.emplace_back([&, t] { fn(std::min(n, t * per), std::min(n, (t + 1) * per)); });
    fn(size_t(0), std::min(n, per));
    for (auto& x : th) x.join();
}
// x moved `steps` representable floats towards -inf / +inf (what a chain of nextafterf calls does, without the calls:
// 18 libm calls per box were 90 ms of a 10^6-leaf commit).  Finite x; stops at +-FLT_MAX.
inline float float_step(float x, int steps, bool up) {
    int32_t i;
    std::memcpy(&i, &x, 4);
    // map the sign-magnitude bit pattern to a monotone integer line: negative floats -> negative integers
    int64_t k = i >= 0 ? int64_t(i) : -int64_t(i & 0x7FFFFFFF);
    k += up ? steps : -steps;
    const int64_t top = 0x7F7FFFFF; // FLT_MAX
    k = std::max<int64_t>(-top, std::min<int64_t>(top, k));
    const int32_t o = k >= 0 ? int32_t(k) : int32_t(uint32_t(-k) | 0x80000000u);
    float r;
    std::memcpy(&r, &o, 4);
    return r;
}

struct Box3 {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    void grow(const Box3& o) {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], o.lo[k]); hi[k] = std::max(hi[k], o.hi[k]); }
    }
    void grow_pt(const double p[3]) {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); }
    }
    double area() const {
        double d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        if (d[0] < 0 || d[1] < 0 || d[2] < 0) return 0.0;
        return 2.0 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    }
};

struct Item {
    uint32_t kind;  // PRIM_*
    int32_t obj;    // graph object id (instances: index into FlatScene::insts)
    Box3 box;
    int32_t seq = 0; // position in the reference's depth-first traversal of the world List
    int32_t world_copy = -1; // spheres only: index into Lowering::world_spheres (the record to emit is that world-space copy)
};
using ItemVec = RecVec<Item>; // (resize() does not initialise: collect() fills a big flat list's items in parallel)
// A sphere of a rigidly transformed group, in world space (collect(), "spheres of transformed groups")
struct WorldSphere { double c[3], r; int32_t home, inst; };

struct Lowering {
    const SceneGraph& g;
    FlatScene& fs;
    std::string& err;
    RecVec<int32_t> tex_index, mat_index; // graph id -> flat index
    int rc = 0;
    const DeviceBvhApi* builder = nullptr; // the device builder, or null: host binned SAH
    size_t max_leaf = 4; // records per leaf of the host SAH build (lower_scene picks it)
    double leaf_cost = 1.0; // what testing a leaf's record costs in node fetches (RTTNW_SAH_LEAF_COST, experiments: < 1 folds more leaves)
    double time0 = 0.0, time1 = 1.0; // shutter interval the moving spheres' boxes must cover

    Lowering(const SceneGraph& graph, FlatScene& flat, std::string& error, const DeviceBvhApi* bvh_builder, size_t leaf_records, double t0, double t1)
        : g(graph), fs(flat), err(error), builder(bvh_builder), max_leaf(leaf_records), time0(t0), time1(t1) {}

    int fail(int code, const std::string& m) { if (!rc) { rc = code; err = m; } return code; }

    // ---- bounds of leaf objects (Sphere :125-130, MovingSphere :233-244 over shutter 0..1,
    //      Rectangle :532-546, Cube :585-591)
    Box3 bounds_of(const GraphObj& o) const {
        Box3 b;
        switch (o.kind) {
        case GraphObj::SPHERE_K:
            for (int k = 0; k < 3; ++k) { b.lo[k] = o.v[k] - std::fabs(o.v[3]); b.hi[k] = o.v[k] + std::fabs(o.v[3]); }
            break;
        case GraphObj::MOVING_K: {
            double r = std::fabs(o.v[8]);
            for (double time : {time0, time1}) { // the centre moves linearly: the box over an interval is the hull of its ends
                double f = (time - o.v[6]) / (o.v[7] - o.v[6]);
                double c[3];
                for (int k = 0; k < 3; ++k) c[k] = o.v[k] + f * (o.v[3 + k] - o.v[k]);
                double p0[3] = {c[0] - r, c[1] - r, c[2] - r}, p1[3] = {c[0] + r, c[1] + r, c[2] + r};
                b.grow_pt(p0); b.grow_pt(p1);
            }
            break;
        }
        case GraphObj::RECT_K: {
            int a0 = o.c == 2 ? 1 : 0, a1 = o.c == 0 ? 1 : 2, ka = o.c == 0 ? 2 : (o.c == 1 ? 1 : 0);
            b.lo[a0] = std::min(o.v[0], o.v[1]); b.hi[a0] = std::max(o.v[0], o.v[1]);
            b.lo[a1] = std::min(o.v[2], o.v[3]); b.hi[a1] = std::max(o.v[2], o.v[3]);
            b.lo[ka] = o.v[4] - 0.0001; b.hi[ka] = o.v[4] + 0.0001;
            break;
        }
        case GraphObj::CUBE_K:
            for (int k = 0; k < 3; ++k) { b.lo[k] = std::min(o.v[k], o.v[3 + k]); b.hi[k] = std::max(o.v[k], o.v[3 + k]); }
            break;
        default: break;
        }
        return b;
    }

    // ---- record emission
    // The records of items[0, n) — one per item, instances excepted — appended to their kinds' arrays in item order, in
    // parallel: idx_out[i] = the record index of item i (an instance's own index for instances).
    void emit_all(const ItemVec& items, std::vector<uint32_t>& idx_out, bool defer_writes = false) {
        const size_t n = items.size();
        idx_out.resize(n);
        if (n < 16384) {
            defer_writes = false;
            for (size_t i = 0; i < n; ++i) idx_out[i] = items[i].kind == PRIM_INSTANCE ? uint32_t(items[i].obj) : emit(items[i]);
            return;
        }
        // per-chunk counts by kind -> where every chunk's records start
        const size_t grain = 8192, n_chunks = (n + grain - 1) / grain;
        std::vector<std::array<uint32_t, 4>> start(n_chunks + 1); // sphere, moving, rect, box
        parallel_for(n_chunks, 1, [&](size_t a, size_t b) {
            for (size_t c = a; c < b; ++c) {
                std::array<uint32_t, 4> k{0, 0, 0, 0};
                for (size_t i = c * grain; i < std::min(n, (c + 1) * grain); ++i)
                    if (rec_array(items[i].kind) < 4u) ++k[rec_array(items[i].kind)];
                start[c + 1] = k;
            }
        });
        start[0] = {uint32_t(fs.spheres.size()), uint32_t(fs.moving.size()), uint32_t(fs.rects.size()), uint32_t(fs.boxes.size())};
        for (size_t c = 1; c <= n_chunks; ++c)
            for (int k = 0; k < 4; ++k) start[c][k] += start[c - 1][k];
        fs.spheres.resize(start[n_chunks][0]); fs.sphere_mat.resize(start[n_chunks][0]); fs.sphere_seq.resize(start[n_chunks][0]);
        fs.moving.resize(start[n_chunks][1]); fs.rects.resize(start[n_chunks][2]); fs.boxes.resize(start[n_chunks][3]);
        parallel_for(n_chunks, 1, [&](size_t a, size_t b) { // where every record goes ...
            for (size_t c = a; c < b; ++c) {
                std::array<uint32_t, 4> at = start[c];
                for (size_t i = c * grain; i < std::min(n, (c + 1) * grain); ++i) {
                    const Item& it = items[i];
                    idx_out[i] = it.kind == PRIM_INSTANCE ? uint32_t(it.obj) : at[rec_array(it.kind)]++;
                }
            }
        });
        if (defer_writes) return; // ... and (write_records(), possibly on another thread beside the device build) the records themselves
        write_records(items, idx_out);
    }
    void write_records(const ItemVec& items, const std::vector<uint32_t>& idx) {
        parallel_for(items.size(), 8192, [&](size_t a, size_t b) {
            for (size_t i = a; i < b; ++i)
                if (items[i].kind != PRIM_INSTANCE) emit_at(items[i], idx[i]);
        });
    }
    // which of the four record arrays (sphere, moving, rect, box) an item's record goes to; 4: none (an instance)
    static uint32_t rec_array(uint32_t kind) { return kind == PRIM_SPHERE_WC ? uint32_t(PRIM_SPHERE) : (kind <= PRIM_BOX ? kind : 4u); }
    uint32_t emit(const Item& it) {
        switch (it.kind) {
        case PRIM_SPHERE_WC:
        case PRIM_SPHERE: fs.spheres.emplace_back(); fs.sphere_mat.emplace_back(); fs.sphere_seq.emplace_back(); return emit_at(it, uint32_t(fs.spheres.size() - 1));
        case PRIM_MOVING_SPHERE: fs.moving.emplace_back(); return emit_at(it, uint32_t(fs.moving.size() - 1));
        case PRIM_RECT: fs.rects.emplace_back(); return emit_at(it, uint32_t(fs.rects.size() - 1));
        case PRIM_BOX: fs.boxes.emplace_back(); return emit_at(it, uint32_t(fs.boxes.size() - 1));
        default: return 0;
        }
    }
    // write item `it`'s record into slot `at` of its kind's array (which already has that slot)
    uint32_t emit_at(const Item& it, uint32_t at) {
        const GraphObj& o = g.objs[it.obj];
        switch (it.kind) {
        case PRIM_SPHERE_WC:
        case PRIM_SPHERE: {
            if (it.world_copy >= 0) {
                const WorldSphere& w = world_spheres[it.world_copy];
                fs.spheres[at] = {w.c[0], w.c[1], w.c[2], w.r};
                fs.sphere_mat[at] = MAT_HOME_FLAG | (w.inst << MAT_HOME_INST_SHIFT) | w.home; // not a material: where its record is made
            } else {
                fs.spheres[at] = {o.v[0], o.v[1], o.v[2], o.v[3]};
                fs.sphere_mat[at] = mat_index[o.a];
            }
            fs.sphere_seq[at] = it.seq;
            return at;
        }
        case PRIM_MOVING_SPHERE: {
            MovingSphereRec<double> m{};
            for (int k = 0; k < 3; ++k) { m.c0[k] = o.v[k]; m.c1[k] = o.v[3 + k]; }
            m.t0 = o.v[6]; m.t1 = o.v[7]; m.r = o.v[8]; m.mat = mat_index[o.a]; m.seq = it.seq;
            fs.moving[at] = m;
            return at;
        }
        case PRIM_RECT: {
            RectRec<double> r{o.v[0], o.v[1], o.v[2], o.v[3], o.v[4], o.c, mat_index[o.a], it.seq};
            fs.rects[at] = r;
            return at;
        }
        case PRIM_BOX: {
            BoxRec<double> bx{};
            for (int k = 0; k < 3; ++k) { bx.mn[k] = o.v[k]; bx.mx[k] = o.v[3 + k]; }
            bx.mat = mat_index[o.a];
            bx.seq = it.seq;
            fs.boxes[at] = bx;
            return at;
        }
        default: return 0;
        }
    }

    static void set_box(float* lo, float* hi, const Box3& b) {
        for (int k = 0; k < 3; ++k) {
            float l = float(b.lo[k]), h = float(b.hi[k]);
            // round outward, then pad two ulps: covers the f32 narrowing of the primitives themselves
            if (std::isfinite(l)) l = float_step(l, double(l) > b.lo[k] ? 3 : 2, false);
            if (std::isfinite(h)) h = float_step(h, double(h) < b.hi[k] ? 3 : 2, true);
            lo[k] = l; hi[k] = h;
        }
    }
    static void set_empty(float* lo, float* hi) {
        for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
    }

    static bool groupable(uint32_t kind) { return kind == PRIM_SPHERE || kind == PRIM_RECT || kind == PRIM_BOX || kind == PRIM_SPHERE_WC; }

    // ---- binned-SAH build over items[lo, hi), in two phases.
    // split(): decides the topology — bins, partitions `items` in place (afterwards the items stand in leaf order) and notes
    // every inner node under the position of its split, which names it uniquely — with the two halves of a large range built
    // by two threads (disjoint parts of `items` and of `topo`): the 10^6 spheres of BASELINE config 5 took 1.0-1.2 s on one thread.
    // number(): one cheap sequential walk that gives the nodes their pre-order places in fs.nodes and the records their
    // places in the kinds' arrays — the output is the one a sequential recursive build would have produced.
    struct Topo {
        int32_t left = 0, right = 0; // >= 0: the inner node whose split position this is; < 0: ~(first item of a leaf)
        float lo0[3], hi0[3], lo1[3], hi1[3]; // the children's boxes, already rounded outward (set_box)
    };
    std::vector<Topo> topo;         // [split position]
    std::vector<uint32_t> leaf_n;   // [first item of a leaf] -> its item count
    static constexpr size_t PAR_SPLIT_MIN = 32768; // smallest range whose halves are worth two threads
    static constexpr size_t PAR_SCAN_MIN = 131072; // smallest range whose own passes (bounds, bins) are shared out

    int32_t split(ItemVec& items, size_t lo, size_t hi, uint32_t depth, Box3& out_box) {
        const size_t n = hi - lo;
        Box3 box, cbox;
        bool same_kind = true;
        const bool wide = n >= PAR_SCAN_MIN; // the few ranges at the top of a big build: their passes over the items are shared out too
        std::mutex merge;
        auto scan = [&](size_t a, size_t b) {
            Box3 bx, cb;
            bool same = true;
            for (size_t i = a; i < b; ++i) {
                bx.grow(items[i].box);
                double c[3];
                for (int k = 0; k < 3; ++k) c[k] = 0.5 * (items[i].box.lo[k] + items[i].box.hi[k]);
                cb.grow_pt(c);
                same = same && items[i].kind == items[lo].kind;
            }
            std::lock_guard<std::mutex> g(merge); // min / max / and: the order of the merges does not matter
            box.grow(bx); cbox.grow(cb); same_kind = same_kind && same;
        };
        if (wide) parallel_for(n, PAR_SCAN_MIN / 4, [&](size_t a, size_t b) { scan(lo + a, lo + b); });
        else scan(lo, hi);
        out_box = box;
        auto make_leaf_here = [&]() -> int32_t { leaf_n[lo] = uint32_t(n); return ~int32_t(lo); };
        if (n == 1) return make_leaf_here();
        const bool can_leaf = same_kind && groupable(items[lo].kind) && n <= max_leaf;

        // best binned split
        constexpr int NB = 32; // (16 / 32 / 64 / 256 bins: 38.9 / 37.9 / 38.1 / 38.0 node visits per sample on final_scene; 32: +1 % there)
        double best_cost = std::numeric_limits<double>::infinity();
        int best_axis = -1, best_bin = -1;
        for (int ax = 0; ax < 3; ++ax) {
            double ext = cbox.hi[ax] - cbox.lo[ax];
            if (!(ext > 0)) continue;
            Box3 bb[NB];
            size_t bc[NB] = {0};
            double scale = NB / ext;
            auto fill = [&](size_t a, size_t b_end, Box3* tb, size_t* tc) {
                for (size_t i = a; i < b_end; ++i) {
                    double c = 0.5 * (items[i].box.lo[ax] + items[i].box.hi[ax]);
                    int b = std::min(NB - 1, std::max(0, int((c - cbox.lo[ax]) * scale)));
                    tb[b].grow(items[i].box);
                    tc[b]++;
                }
            };
            if (wide) {
                parallel_for(n, PAR_SCAN_MIN / 4, [&](size_t a, size_t b_end) {
                    Box3 tb[NB];
                    size_t tc[NB] = {0};
                    fill(lo + a, lo + b_end, tb, tc);
                    std::lock_guard<std::mutex> g(merge);
                    for (int b = 0; b < NB; ++b) { bb[b].grow(tb[b]); bc[b] += tc[b]; }
                });
            } else {
                fill(lo, hi, bb, bc);
            }
            double right_area[NB];
            size_t right_cnt[NB];
            Box3 acc;
            size_t cnt = 0;
            for (int b = NB - 1; b > 0; --b) { acc.grow(bb[b]); cnt += bc[b]; right_area[b] = acc.area(); right_cnt[b] = cnt; }
            acc = Box3();
            cnt = 0;
            for (int b = 0; b < NB - 1; ++b) {
                acc.grow(bb[b]); cnt += bc[b];
                if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                double cost = acc.area() * double(cnt) + right_area[b + 1] * double(right_cnt[b + 1]);
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = b; }
            }
        }
        if (can_leaf) {
            // SAH: leaf cost n * A vs. split cost A (one node fetch) + children
            double a = box.area();
            if (best_axis < 0 || !(best_cost + a < double(n) * a * leaf_cost)) return make_leaf_here();
        }
        size_t mid;
        if (best_axis >= 0) {
            double ext = cbox.hi[best_axis] - cbox.lo[best_axis], scale = NB / ext, clo = cbox.lo[best_axis];
            int ax = best_axis, bin = best_bin;
            auto it = std::partition(items.begin() + lo, items.begin() + hi, [&](const Item& x) {
                double c = 0.5 * (x.box.lo[ax] + x.box.hi[ax]);
                int b = std::min(NB - 1, std::max(0, int((c - clo) * scale)));
                return b <= bin;
            });
            mid = size_t(it - items.begin());
        } else if (!same_kind) { // coincident centroids of different kinds: separate the kinds
            uint32_t k0 = items[lo].kind;
            auto it = std::partition(items.begin() + lo, items.begin() + hi, [&](const Item& x) { return x.kind == k0; });
            mid = size_t(it - items.begin());
        } else {
            mid = lo + n / 2;
        }
        if (mid == lo || mid == hi || depth > 40) {
            // degenerate or runaway split: fall back to an object-median cut on the widest axis
            int ax = 0;
            for (int k = 1; k < 3; ++k)
                if (cbox.hi[k] - cbox.lo[k] > cbox.hi[ax] - cbox.lo[ax]) ax = k;
            mid = lo + n / 2;
            std::nth_element(items.begin() + lo, items.begin() + mid, items.begin() + hi, [ax](const Item& x, const Item& y) {
                return x.box.lo[ax] + x.box.hi[ax] < y.box.lo[ax] + y.box.hi[ax];
            });
        }

        Box3 b0, b1;
        int32_t c0, c1;
        if (n >= PAR_SPLIT_MIN) {
            std::thread left([&] { c0 = split(items, lo, mid, depth + 1, b0); });
            c1 = split(items, mid, hi, depth + 1, b1);
            left.join();
        } else {
            c0 = split(items, lo, mid, depth + 1, b0);
            c1 = split(items, mid, hi, depth + 1, b1);
        }
        Topo& t = topo[mid];
        t.left = c0; t.right = c1;
        set_box(t.lo0, t.hi0, b0);
        set_box(t.lo1, t.hi1, b1);
        return int32_t(mid);
    }
    // child code (node index or leaf bits) of the subtree `id` of split(); rec[i] = the record (or instance) index of item i
    int32_t number(const ItemVec& items, const std::vector<uint32_t>& rec, int32_t id, uint32_t depth, uint32_t& max_depth) {
        max_depth = std::max(max_depth, depth);
        if (id < 0) {
            const size_t lo = size_t(~id);
            fs.n_prims_in_bvh += leaf_n[lo];
            return make_leaf(items[lo].kind, leaf_n[lo], rec[lo]); // a leaf's records are neighbours: emitted in item order
        }
        const Topo& t = topo[size_t(id)];
        const int32_t me = int32_t(fs.nodes.size());
        fs.nodes.emplace_back();
        const int32_t c0 = number(items, rec, t.left, depth + 1, max_depth);
        const int32_t c1 = number(items, rec, t.right, depth + 1, max_depth);
        BvhNode& nd = fs.nodes[me];
        for (int k = 0; k < 3; ++k) { nd.lo0[k] = t.lo0[k]; nd.hi0[k] = t.hi0[k]; nd.lo1[k] = t.lo1[k]; nd.hi1[k] = t.hi1[k]; }
        nd.child0 = c0; nd.child1 = c1; nd.pad0 = nd.pad1 = 0;
        return me;
    }
    int32_t build(ItemVec& items, uint32_t& max_depth, Box3& out_box) {
        topo.assign(items.size() + 1, Topo());
        leaf_n.assign(items.size() + 1, 0u);
        auto T0 = std::chrono::steady_clock::now();
        const int32_t root = split(items, 0, items.size(), 1, out_box);
        auto T1 = std::chrono::steady_clock::now();
        std::vector<uint32_t> rec;
        emit_all(items, rec); // the items now stand in leaf order
        auto T2 = std::chrono::steady_clock::now();
        fs.nodes.reserve(fs.nodes.size() + items.size());
        const int32_t code = number(items, rec, root, 1, max_depth);
        auto T3 = std::chrono::steady_clock::now();
        if (getenv("RTTNW_DEBUG_LOWER") && items.size() > 100000) fprintf(stderr, "[sah] split %.1f emit %.1f number %.1f ms\n", std::chrono::duration<double, std::milli>(T1 - T0).count(), std::chrono::duration<double, std::milli>(T2 - T1).count(), std::chrono::duration<double, std::milli>(T3 - T2).count());
        topo = std::vector<Topo>();
        leaf_n = std::vector<uint32_t>();
        return code;
    }

    // Build a BVH whose root is always a node record.  Returns the root index; `depth_out` = levels of inner nodes on
    // the longest root-to-leaf path (what bounds the traversal stack: one pending sibling per inner level).
    static constexpr int32_t DEVICE_ROOT = 0x40000000; // build_root's result for a tree the device builder made: DEVICE_ROOT + its index in fs.device_trees
    int32_t build_root(ItemVec& items, uint32_t& depth_out, Box3& box_out) {
        uint32_t md = 0;
        if (items.empty()) {
            BvhNode nd{};
            set_empty(nd.lo0, nd.hi0); set_empty(nd.lo1, nd.hi1);
            nd.child0 = nd.child1 = CHILD_EMPTY;
            fs.nodes.push_back(nd);
            depth_out = 1;
            box_out = Box3();
            return int32_t(fs.nodes.size() - 1);
        }
        if (builder && items.size() >= std::max<size_t>(2, builder->min_leaves)) {
            // external builder (device LBVH): every item is a one-record leaf; records are emitted in item order
            RecVec<BuildPrim> prims(items.size()); // (not initialised: every leaf is written below)
            std::vector<uint32_t> idx;
            const bool beside = items.size() >= 16384; // a big list: its records are written by host threads WHILE the device builds the tree
            emit_all(items, idx, beside);
            // the leaves, and in the same pass their bounds and the bounds of their centres (which the builder would otherwise walk
            // the leaves for once more): min / max merged under a mutex, order-independent
            float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY}, cb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
            std::mutex merge;
            parallel_for(items.size(), 8192, [&](size_t a, size_t b) {
                float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, c[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                for (size_t i = a; i < b; ++i) {
                    BuildPrim& q = prims[i];
                    set_box(q.lo, q.hi, items[i].box);
                    q.leaf = make_leaf(items[i].kind, 1, idx[i]);
                    q.pad = 0;
                    for (int k = 0; k < 3; ++k) {
                        lo[k] = std::min(lo[k], q.lo[k]); hi[k] = std::max(hi[k], q.hi[k]);
                        const float ck = 0.5f * (q.lo[k] + q.hi[k]);
                        c[k] = std::min(c[k], ck); c[3 + k] = std::max(c[3 + k], ck);
                    }
                }
                std::lock_guard<std::mutex> g(merge);
                for (int k = 0; k < 3; ++k) {
                    blo[k] = std::min(blo[k], lo[k]); bhi[k] = std::max(bhi[k], hi[k]);
                    cb[k] = std::min(cb[k], c[k]); cb[3 + k] = std::max(cb[3 + k], c[3 + k]);
                }
            });
            box_out = Box3(); // (the root's box is only used for a wrapping instance's bounds: the f32 boxes cover the f64 ones)
            for (int k = 0; k < 3; ++k) { box_out.lo[k] = double(blo[k]); box_out.hi[k] = double(bhi[k]); }
            fs.n_prims_in_bvh += uint32_t(items.size());
            DeviceTree tree;
            std::string berr;
            std::thread writer;
            if (beside) writer = std::thread([&] { write_records(items, idx); });
            const int brc = builder->build(prims.data(), prims.size(), cb, tree, berr);
            if (writer.joinable()) writer.join();
            if (brc) { fail(brc, berr); depth_out = 1; return 0; }
            depth_out = tree.levels;
            fs.device_trees.push_back(std::move(tree));
            return DEVICE_ROOT + int32_t(fs.device_trees.size() - 1); // a handle: the tree's place in the node array is known only at the end (run())
        }
        // reserve the root slot first so that it precedes its subtree
        int32_t code = build(items, md, box_out);
        depth_out = std::max(1u, md - 1); // build() counts the leaf level too; a lone leaf gets the wrapper node below
        if (code >= 0) return code;
        BvhNode nd{};
        set_box(nd.lo0, nd.hi0, box_out);
        set_empty(nd.lo1, nd.hi1);
        nd.child0 = code; nd.child1 = CHILD_EMPTY;
        fs.nodes.push_back(nd);
        return int32_t(fs.nodes.size() - 1);
    }

    // ---- transform chains
    // Walk Translate/YRotate wrappers from `id` inwards; ops[0] = outermost.  Returns the wrapped object.
    int32_t peel_ops(int32_t id, InstanceRec<double>& in) {
        in = InstanceRec<double>{};
        int32_t cur = id;
        while (g.objs[cur].kind == GraphObj::TRANSLATE_K || g.objs[cur].kind == GraphObj::ROTATE_K) {
            const GraphObj& o = g.objs[cur];
            if (in.n_ops >= MAX_INSTANCE_OPS) { fail(ERR_UNSUPPORTED, "more than 8 nested translate/rotate_y wrappers around one object"); return -1; }
            auto& op = in.ops[in.n_ops++];
            if (o.kind == GraphObj::TRANSLATE_K) {
                op.type = OP_TRANSLATE;
                op.v[0] = o.v[0]; op.v[1] = o.v[1]; op.v[2] = o.v[2];
            } else { // YRotate::new — hittable.rs:641-643
                op.type = OP_ROTATE_Y;
                double radians = o.v[0] * (kPi / 180.0);
                op.v[0] = std::sin(radians); op.v[1] = std::cos(radians); op.v[2] = 0;
            }
            cur = o.a;
        }
        return cur;
    }
    // object -> world for a point (the CORRECT inverse; instance bounds must be conservative, Q2)
    static void to_world(const InstanceRec<double>& in, double p[3]) {
        for (int i = in.n_ops - 1; i >= 0; --i) {
            if (in.ops[i].type == OP_TRANSLATE) {
                p[0] += in.ops[i].v[0]; p[1] += in.ops[i].v[1]; p[2] += in.ops[i].v[2];
            } else {
                double s = in.ops[i].v[0], c = in.ops[i].v[1];
                double x = c * p[0] + s * p[2], z = -s * p[0] + c * p[2];
                p[0] = x; p[2] = z;
            }
        }
    }

    // ---- collection of solids
    // `outer`: the Translate / YRotate wrappers around the list being collected, outermost first (empty at world level).
    // A transformed object inside a transformed group becomes an instance of its own at the TOP level whose chain is
    // outer + its own wrappers — the closest hit does not care how the reference nests its lists — so instances never nest.
    struct Chain { int32_t n = 0; InstanceRec<double>::Op ops[MAX_INSTANCE_OPS]; };
    int32_t next_seq = 0;
    int nesting = 0; // recursion guard: a list that (transitively) contains itself
    ItemVec* top_items = nullptr;
    std::vector<WorldSphere> world_spheres;
    bool move_spheres = true;         // false keeps them in their groups' trees (lower_scene's world_spheres; RTTNW_WORLD_SPHERES)
    bool test_in_group_frame = false; // lower_scene's world_spheres == 2: the copies are leaves of kind PRIM_SPHERE_WC — culled in world space, tested in their group's frame
    int world_spheres_arg = -1;       // lower_scene's argument: -1 = the default above or the environment's word

    bool append_ops(Chain& c, const InstanceRec<double>& in) {
        for (int i = 0; i < in.n_ops; ++i) {
            if (c.n >= MAX_INSTANCE_OPS) { fail(ERR_UNSUPPORTED, "more than 8 nested translate/rotate_y wrappers around one object"); return false; }
            c.ops[c.n++] = in.ops[i];
        }
        return true;
    }
    static void chain_to_inst(const Chain& c, InstanceRec<double>& in) {
        in = InstanceRec<double>{};
        in.n_ops = c.n;
        for (int i = 0; i < c.n; ++i) in.ops[i] = c.ops[i];
    }
    // boundary of a medium: spheres and cubes, directly or in (nested) lists / BvhTrees, all in one space
    bool boundary_members(int32_t id, std::vector<int32_t>& refs) {
        const GraphObj& o = g.objs[id];
        if (o.kind == GraphObj::SPHERE_K) { refs.push_back(make_ref(PRIM_SPHERE, emit({PRIM_SPHERE, id, Box3(), 0}))); return true; }
        if (o.kind == GraphObj::CUBE_K) { refs.push_back(make_ref(PRIM_BOX, emit({PRIM_BOX, id, Box3(), 0}))); return true; }
        if (o.kind == GraphObj::LIST_K || o.kind == GraphObj::BVH_K) {
            if (++nesting > 64) { fail(ERR_UNSUPPORTED, "scene graph nests deeper than 64 levels (cycle?)"); return false; }
            for (int32_t it : o.items)
                if (!boundary_members(it, refs)) { --nesting; return false; }
            --nesting;
            return true;
        }
        fail(ERR_UNSUPPORTED, "constant_medium boundary must be a sphere, a cube or a list / bvh_tree of those (optionally translated/rotated as a whole)");
        return false;
    }

    void collect(int32_t id, ItemVec& out, const Chain& outer) {
        if (rc) return;
        struct Guard { int& n; Guard(int& x) : n(x) { ++n; } ~Guard() { --n; } } guard(nesting);
        if (nesting > 64) { fail(ERR_UNSUPPORTED, "scene graph nests deeper than 64 levels (cycle?)"); return; }
        const GraphObj& o = g.objs[id];
        switch (o.kind) {
        case GraphObj::SPHERE_K: out.push_back({PRIM_SPHERE, id, bounds_of(o), next_seq++}); break;
        case GraphObj::MOVING_K: out.push_back({PRIM_MOVING_SPHERE, id, bounds_of(o), next_seq++}); break;
        case GraphObj::RECT_K: out.push_back({PRIM_RECT, id, bounds_of(o), next_seq++}); break;
        case GraphObj::CUBE_K: out.push_back({PRIM_BOX, id, bounds_of(o), next_seq++}); break;
        case GraphObj::LIST_K:
        case GraphObj::BVH_K: {
            // a big flat list of plain leaves (10^6 spheres): its items in parallel — same Items, same sequence numbers.  ONE pass:
            // the items are written on the assumption that every member is a plain leaf, which the same pass verifies; a list that
            // holds anything else is walked member by member as usual (what was written is dropped)
            const size_t n = o.items.size();
            if (n >= 16384) {
                const size_t base = out.size();
                const int32_t seq0 = next_seq;
                out.resize(base + n);
                std::atomic<bool> all_leaves{true};
                parallel_for(n, 8192, [&](size_t a, size_t b) {
                    for (size_t i = a; i < b; ++i) {
                        const int32_t it = o.items[i];
                        const GraphObj& q = g.objs[it];
                        uint32_t kind;
                        switch (q.kind) {
                        case GraphObj::SPHERE_K: kind = PRIM_SPHERE; break;
                        case GraphObj::MOVING_K: kind = PRIM_MOVING_SPHERE; break;
                        case GraphObj::RECT_K: kind = PRIM_RECT; break;
                        case GraphObj::CUBE_K: kind = PRIM_BOX; break;
                        default: all_leaves = false; return;
                        }
                        Item item{kind, it, bounds_of(q), seq0 + int32_t(i)};
                        out[base + i] = item;
                    }
                });
                if (all_leaves) { next_seq += int32_t(n); break; }
                out.resize(base);
            }
            for (int32_t it : o.items) collect(it, out, outer);
            break;
        }
        case GraphObj::TRANSLATE_K:
        case GraphObj::ROTATE_K: {
            InstanceRec<double> own;
            int32_t inner = peel_ops(id, own);
            if (rc) return;
            Chain full = outer;
            if (!append_ops(full, own)) return;
            InstanceRec<double> in;
            chain_to_inst(full, in);
            ItemVec sub;
            collect(inner, sub, full);
            if (rc) return;
            // Spheres of transformed groups: Translate and YRotate are rigid (the forward half of YRotate::hit is a proper
            // rotation, hittable.rs:687-694; quirk Q1 sits in the way BACK), so a sphere under them is a sphere in world space:
            // centre carried through the chain, same radius, same t along the same ray.  The walk tests that world-space
            // copy in the TOP tree — no ray transform, no second tree, one SAH build over everything (final_scene's 1000-sphere
            // cluster) — and the hit record is still made the reference's way, in object space through the chain
            // (make_record: SceneView::sphere_home leads from the copy to the object-space record and the chain).
            const int32_t inst_index = int32_t(fs.insts.size()); // of the record pushed below
            bool moved = false;
            if (move_spheres) {
                ItemVec keep;
                for (const Item& it : sub) {
                    // (the copy's material slot holds where its record is made: 20 bits of sphere index, 9 of chain index)
                    if (it.kind != PRIM_SPHERE || inst_index > MAT_HOME_INST_MAX || fs.spheres.size() >= MAT_HOME_SPHERE_MAX) { keep.push_back(it); continue; }
                    const GraphObj& so = g.objs[it.obj];
                    WorldSphere w{{so.v[0], so.v[1], so.v[2]}, so.v[3], int32_t(emit(it)), inst_index};
                    to_world(in, w.c);
                    Box3 wb;
                    for (int k = 0; k < 3; ++k) { // (+ the rounding of the carried centre)
                        const double pad = std::fabs(w.r) + 1e-9 * std::max(1.0, std::fabs(w.c[k]) + std::fabs(w.r));
                        wb.lo[k] = w.c[k] - pad; wb.hi[k] = w.c[k] + pad;
                    }
                    world_spheres.push_back(w);
                    Item copy{test_in_group_frame ? uint32_t(PRIM_SPHERE_WC) : uint32_t(PRIM_SPHERE), it.obj, wb, it.seq};
                    copy.world_copy = int32_t(world_spheres.size() - 1);
                    top_items->push_back(copy);
                    moved = true;
                }
                sub.swap(keep);
            }
            if (sub.empty()) {
                if (moved) { in.root = -1; in.single_leaf = 0; fs.insts.push_back(in); } // the chain alone: the moved spheres' records go through it
                break; // (else: nothing but nested instances / media inside: they went to the top level themselves)
            }
            uint32_t depth = 0;
            Box3 ob;
            in.root = build_root(sub, depth, ob);
            // a wrapped single object (the Cornell blocks: `cube.rotate_y(a).translate(v)`): build_root made a node whose
            // only child is that record's leaf; the walk tests the record in place instead of entering a one-node tree
            in.single_leaf = 0;
            if (sub.size() == 1 && fs.nodes[in.root].child0 < 0 && fs.nodes[in.root].child1 == CHILD_EMPTY &&
                leaf_count(fs.nodes[in.root].child0) == 1)
                in.single_leaf = fs.nodes[in.root].child0;
            Box3 wb;
            for (int c = 0; c < 8; ++c) {
                double p[3] = {(c & 1) ? ob.hi[0] : ob.lo[0], (c & 2) ? ob.hi[1] : ob.lo[1], (c & 4) ? ob.hi[2] : ob.lo[2]};
                to_world(in, p);
                wb.grow_pt(p);
            }
            // guard the sin/cos rounding of the corner transform
            for (int k = 0; k < 3; ++k) {
                double pad = 1e-9 * std::max(1.0, std::max(std::fabs(wb.lo[k]), std::fabs(wb.hi[k])));
                wb.lo[k] -= pad; wb.hi[k] += pad;
            }
            fs.insts.push_back(in);
            top_items->push_back({PRIM_INSTANCE, int32_t(fs.insts.size() - 1), wb, 0}); // always a leaf of the TOP tree
            break;
        }
        case GraphObj::MEDIUM_K: {
            InstanceRec<double> own;
            int32_t base = peel_ops(o.a, own);
            if (rc) return;
            Chain full = outer;
            if (!append_ops(full, own)) return;
            MediumRec<double> md{};
            std::vector<int32_t> refs;
            if (!boundary_members(base, refs)) return;
            if (refs.empty()) { fail(ERR_UNSUPPORTED, "constant_medium with an empty boundary"); return; }
            md.b_first = int32_t(fs.medium_refs.size());
            md.b_count = int32_t(refs.size());
            md.ref0 = refs[0];
            fs.medium_refs.insert(fs.medium_refs.end(), refs.begin(), refs.end());
            md.inst = -1;
            md.n_outer = outer.n;
            if (full.n > 0) {
                InstanceRec<double> in;
                chain_to_inst(full, in);
                in.root = -1;
                fs.insts.push_back(in);
                md.inst = int32_t(fs.insts.size() - 1);
            }
            md.mat = mat_index[o.b] & MAT_INDEX_MASK;
            md.neg_inv_density = -1. / o.v[0]; // hittable.rs:733
            // media keep their creation order (= RNG slot, DESIGN.md "RNG")
            if (size_t(o.c) >= fs.media.size()) fs.media.resize(size_t(o.c) + 1, MediumRec<double>{-1, 0, -1, 0, 0, make_ref(PRIM_NONE, 0), 0.0});
            fs.media[o.c] = md;
            break;
        }
        default: fail(ERR_INVALID, "world contains a non-hittable object"); break;
        }
    }

    void lower_textures_materials() {
        // Texture and material records in graph-id order.  Scenes give every sphere its own material (and colour: 3 x 10^6 graph
        // objects for spheres_1m), so both passes run in parallel over id chunks: count, prefix, fill.
        const size_t n_obj = g.objs.size(), grain = 16384, n_chunks = (n_obj + grain - 1) / grain;
        tex_index.resize(n_obj); // (not initialised by resize: the counting pass below writes the -1s, in parallel)
        mat_index.resize(n_obj);
        std::vector<uint32_t> tex_start(n_chunks + 1, 0), mat_start(n_chunks + 1, 0);
        parallel_for(n_chunks, 1, [&](size_t ca, size_t cb) {
            for (size_t c = ca; c < cb; ++c) {
                uint32_t nt = 0, nm = 0;
                for (size_t id = c * grain; id < std::min(n_obj, (c + 1) * grain); ++id) {
                    tex_index[id] = -1; mat_index[id] = -1;
                    const GraphObj::Kind k = g.objs[id].kind;
                    nt += k <= GraphObj::TEX_IMAGE_K;
                    nm += k == GraphObj::MAT_K;
                }
                tex_start[c + 1] = nt; mat_start[c + 1] = nm;
            }
        });
        for (size_t c = 1; c <= n_chunks; ++c) { tex_start[c] += tex_start[c - 1]; mat_start[c] += mat_start[c - 1]; }
        fs.texs.resize(tex_start[n_chunks]);
        fs.mats.resize(mat_start[n_chunks]);
        parallel_for(n_chunks, 1, [&](size_t ca, size_t cb) { // indices first: a material may name a texture of another chunk
            for (size_t c = ca; c < cb; ++c) {
                uint32_t at = tex_start[c], am = mat_start[c];
                for (size_t id = c * grain; id < std::min(n_obj, (c + 1) * grain); ++id) {
                    const GraphObj::Kind k = g.objs[id].kind;
                    if (k <= GraphObj::TEX_IMAGE_K) tex_index[id] = int32_t(at++);
                    else if (k == GraphObj::MAT_K) mat_index[id] = int32_t(am++);
                }
            }
        });
        parallel_for(n_chunks, 1, [&](size_t ca, size_t cb) {
            for (size_t c = ca; c < cb; ++c) {
                for (size_t id = c * grain; id < std::min(n_obj, (c + 1) * grain); ++id) {
                    const GraphObj& o = g.objs[id];
                    if (o.kind == GraphObj::MAT_K) {
                        MaterialRec<double> m{};
                        m.type = o.c;
                        m.tex = -1;
                        m.albedo[0] = o.v[0]; m.albedo[1] = o.v[1]; m.albedo[2] = o.v[2];
                        m.param = o.v[3];
                        if (o.a >= 0) { // textured: inline a solid colour (read off the graph object), else point at the texture record
                            const GraphObj& t = g.objs[o.a];
                            if (t.kind == GraphObj::TEX_SOLID_K) { m.albedo[0] = t.v[0]; m.albedo[1] = t.v[1]; m.albedo[2] = t.v[2]; }
                            else m.tex = tex_index[o.a];
                            // an image texture, possibly under a checker, reads the hit's (u, v) (a missing image is TEX_CYAN and reads nothing)
                            if ((t.kind == GraphObj::TEX_IMAGE_K && t.a >= 0) || t.kind == GraphObj::TEX_CHECKER_K) mat_index[id] |= MAT_UV_FLAG;
                        }
                        fs.mats[size_t(mat_index[id] & MAT_INDEX_MASK)] = m;
                        continue;
                    }
                    if (o.kind > GraphObj::TEX_IMAGE_K) continue;
                    TextureRec<double> t{};
                    switch (o.kind) {
                    case GraphObj::TEX_SOLID_K: t.type = TEX_SOLID; t.color[0] = o.v[0]; t.color[1] = o.v[1]; t.color[2] = o.v[2]; break;
                    case GraphObj::TEX_CHECKER_K: t.type = TEX_CHECKER; t.a = o.a; t.b = o.b; break; // remapped below
                    case GraphObj::TEX_NOISE_K: t.type = TEX_NOISE; t.a = o.a; t.scale = o.v[0]; break;
                    default: // image
                        if (o.a < 0) { t.type = TEX_CYAN; break; }
                        t.type = TEX_IMAGE; t.a = o.a;
                        break;
                    }
                    fs.texs[size_t(tex_index[id])] = t;
                }
            }
        });
        for (auto& t : fs.texs)
            if (t.type == TEX_CHECKER) { t.a = tex_index[t.a]; t.b = tex_index[t.b]; }
        // images
        for (size_t i = 0; i < g.image_data.size(); ++i) {
            ImageRec im{uint32_t(fs.texels.size()), g.image_w[i], g.image_h[i], 0};
            const auto& d = g.image_data[i];
            for (size_t p = 0; p + 3 < d.size(); p += 4)
                fs.texels.push_back(uint32_t(d[p]) | (uint32_t(d[p + 1]) << 8) | (uint32_t(d[p + 2]) << 16) | (uint32_t(d[p + 3]) << 24));
            fs.images.push_back(im);
        }
        // Perlin tables — Perlin::new (noise.rs:40-47): points, then the x, y, z permutations
        fs.perlin_vec.resize(size_t(g.n_noise) * 768);
        fs.perlin_perm.resize(size_t(g.n_noise) * 768);
        for (uint32_t n = 0; n < g.n_noise; ++n) {
            SceneRng rng(g.seed, 0x100 + n);
            for (int i = 0; i < 768; ++i) fs.perlin_vec[size_t(n) * 768 + i] = rng.range(-1., 1.); // noise.rs:15-19, not normalised
            for (int t = 0; t < 3; ++t) { // noise.rs:21-29 (Fisher-Yates, from the back)
                uint8_t* p = &fs.perlin_perm[size_t(n) * 768 + size_t(t) * 256];
                for (int i = 0; i < 256; ++i) p[i] = uint8_t(i);
                for (uint32_t i = 255; i >= 1; --i) std::swap(p[i], p[rng.below(i + 1)]);
            }
        }
    }

    // ---- binary tree -> 4-wide records (rt_types.hpp Bvh4Node).  A node takes its two children and then, while it has
    // fewer than four, replaces the inner child of largest surface area by that child's two children.  Returns the
    // index of the record; `need` = stack entries a walk below this record can have pending at once: a visit pushes
    // every hit child but the one it descends into, so need = (children - 1) + the largest need among the children.
    struct Slot { float lo[3], hi[3]; int32_t child; };
    static double slot_area(const Slot& s) {
        const double d[3] = {double(s.hi[0]) - s.lo[0], double(s.hi[1]) - s.lo[1], double(s.hi[2]) - s.lo[2]};
        if (d[0] < 0 || d[1] < 0 || d[2] < 0) return 0.0;
        return 2.0 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    }
    static void child_slots(const BvhNode& nd, Slot* out, int& n) { // the non-empty children of a binary node
        if (nd.child0 != CHILD_EMPTY) { Slot& s = out[n++]; std::memcpy(s.lo, nd.lo0, 12); std::memcpy(s.hi, nd.hi0, 12); s.child = nd.child0; }
        if (nd.child1 != CHILD_EMPTY) { Slot& s = out[n++]; std::memcpy(s.lo, nd.lo1, 12); std::memcpy(s.hi, nd.hi1, 12); s.child = nd.child1; }
    }
    int32_t collapse4(int32_t b, uint32_t& need) {
        Slot slots[5];
        int n = 0;
        child_slots(fs.nodes[b], slots, n);
        while (n < 4) {
            int pick = -1;
            double best = -1.0;
            for (int i = 0; i < n; ++i) // (picking by the area the expansion saves instead — area minus the children's — is far worse:
                if (slots[i].child >= 0 && slot_area(slots[i]) > best) { best = slot_area(slots[i]); pick = i; } // 6.1 trips per walk against 4.2)
            if (pick < 0) break;
            const BvhNode inner = fs.nodes[slots[pick].child];
            for (int i = pick; i + 1 < n; ++i) slots[i] = slots[i + 1]; // keep the order of the others
            --n;
            child_slots(inner, slots, n);
        }
        const int32_t me = int32_t(fs.nodes4.size());
        fs.nodes4.emplace_back();
        Bvh4Node out{};
        uint32_t deepest = 0;
        for (int c = 0; c < 4; ++c) {
            if (c < n) {
                for (int a = 0; a < 3; ++a) { out.lo[a][c] = slots[c].lo[a]; out.hi[a][c] = slots[c].hi[a]; }
                if (slots[c].child >= 0) {
                    uint32_t sub = 0;
                    out.child[c] = collapse4(slots[c].child, sub);
                    deepest = std::max(deepest, sub);
                } else {
                    out.child[c] = slots[c].child;
                }
            } else {
                for (int a = 0; a < 3; ++a) { out.lo[a][c] = INFINITY; out.hi[a][c] = -INFINITY; }
                out.child[c] = CHILD_EMPTY;
            }
        }
        fs.nodes4[me] = out;
        need = uint32_t(n > 0 ? n - 1 : 0) + deepest;
        return me;
    }

    int run() {
        if (g.world < 0 || g.objs[g.world].kind != GraphObj::LIST_K) return fail(-2, "commit: world not set");
        const bool timing = getenv("RTTNW_DEBUG_LOWER") != nullptr; // phase times of the lowering on stderr
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        const auto t_start = now();
        lower_textures_materials();
        const auto t_mats = now();
        if (const char* e = getenv("RTTNW_SAH_LEAF_COST")) { const double v = std::atof(e); if (v > 0) leaf_cost = v; }
        if (world_spheres_arg >= 0) { move_spheres = world_spheres_arg != 0; test_in_group_frame = world_spheres_arg == 2; }
        // (the environment chooses between 0 and 1 only: leaves of kind PRIM_SPHERE_WC — 2 — exist for the relowering RTTNW_F64_STRICT renders, asked for
        // by argument; the contracted kernels are compiled without that kind, and a default lowering that emitted it would lose those spheres)
        // (the host test build of the core — tests/hostsim, RT_HOST_TEST_BUILD — knows the kind in every precision and takes 2 from the environment)
        else if (const char* e = getenv("RTTNW_WORLD_SPHERES")) {
            move_spheres = std::atoi(e) != 0;
#if defined(RT_HOST_TEST_BUILD)
            test_in_group_frame = std::atoi(e) == 2;
#endif
        }
        ItemVec top;
        top_items = &top;
        collect(g.world, top, Chain{});
        if (rc) return rc;
        for (const auto& md : fs.media)
            if (md.b_first < 0) return fail(ERR_UNSUPPORTED, "a constant_medium was created but is not in the world list");
        if (fs.media.size() > 16) return fail(ERR_UNSUPPORTED, "more than 16 constant media (the free-flight draw of medium m uses RNG slot m < 16)");
        uint32_t top_depth = 0;
        Box3 wb;
        const auto t_collected = now();
        fs.top_root2 = build_root(top, top_depth, wb);
        if (rc) return rc;
        const auto t_built = now();
        // The kernels walk 4-wide records: collapse the host-built trees here; the device builder has collapsed its own.
        // The node array is [host-built records][device tree 0][device tree 1]...: the device trees learn their place now.
        uint32_t top_need = 0, inst_need = 0;
        if (fs.top_root2 < DEVICE_ROOT) fs.top_root = collapse4(fs.top_root2, top_need);
        bool any_tree = false;
        for (auto& in : fs.insts) {
            if (in.root < 0) continue; // a medium's transform chain: no tree
            any_tree = true;
            if (in.root >= DEVICE_ROOT) continue;
            uint32_t need = 0;
            in.root = collapse4(in.root, need);
            inst_need = std::max(inst_need, need);
        }
        fs.n_host4 = uint32_t(fs.nodes4.size());
        fs.n_host2 = uint32_t(fs.nodes.size());
        {
            uint32_t base4 = fs.n_host4, base2 = fs.n_host2;
            for (DeviceTree& t : fs.device_trees) {
                std::string berr;
                if (int brc = builder->rebase(t, base4, base2, berr)) return fail(brc, berr);
                base4 += t.count4;
                base2 += t.count2;
            }
        }
        if (fs.top_root2 >= DEVICE_ROOT) {
            const DeviceTree& t = fs.device_trees[size_t(fs.top_root2 - DEVICE_ROOT)];
            fs.top_root = int32_t(t.base4); // record 0 of a device tree is its root
            fs.top_root2 = int32_t(t.base2);
            top_need = t.need;
        }
        for (auto& in : fs.insts)
            if (in.root >= DEVICE_ROOT) {
                const DeviceTree& t = fs.device_trees[size_t(in.root - DEVICE_ROOT)];
                in.root = int32_t(t.base4);
                inst_need = std::max(inst_need, t.need);
            }
        // Entries a lane's stack can hold at once: the pending children of the top tree and, while inside an instance,
        // one sentinel plus the pending children of the instance's tree.  +1 spare.
        fs.stack_depth = top_need + (any_tree ? 1u + inst_need : 0u) + 1u;
        if (timing)
            fprintf(stderr, "[lower] textures + materials %.1f ms, collect %.1f ms, top tree (%zu items) %.1f ms, 4-wide collapse (%zu -> %zu records) %.1f ms\n", ms(t_start, t_mats), ms(t_mats, t_collected),
                    top.size(), ms(t_collected, t_built), size_t(fs.total_nodes2()), size_t(fs.total_nodes4()), ms(t_built, now()));
        for (const auto& in : fs.insts) fs.needs_general = fs.needs_general || in.n_ops > FAST_INSTANCE_OPS;
        // (a record without a tree is a bare chain — the way back for the hit records of world-space copies: no leaf refers to it — or a single
        // wrapped record, which the leaf step tests in place: neither takes the walk out of its frame)
        for (const auto& in : fs.insts) fs.walk_changes_frames = fs.walk_changes_frames || (in.root != -1 && in.single_leaf == 0);
        for (const auto& in : fs.insts) fs.has_instance_leaves = fs.has_instance_leaves || in.root != -1 || in.single_leaf != 0;
        for (const auto& md : fs.media) fs.needs_general = fs.needs_general || md.b_count > 1 || md.n_outer > 0;
        return 0;
    }
};

} // namespace

// Leaf size of the host SAH build.  A record test costs 1.5-2.5 node steps in the kernels and its code runs for few
// lanes at a time, so ONE record per leaf is best (cornell_box: 18 instead of 46 record tests per sample, +14 %) —
// unless the extra nodes push the tree out of the LDS-resident form of the trace kernel (final_scene: 1406 nodes
// instead of 891 no longer fit beside the stacks, -14 %).  So: the finest of 1 / 2 / 4 records per leaf whose tree
// still fits; big scenes (which never fit) take 4 and save a third of the node memory.
static bool fits_lds_form(const FlatScene& f) {
    return lds_form_bytes(f.total_nodes4(), f.stack_depth, 1024) <= 160 * 1024;
}
int lower_scene(const SceneGraph& g, FlatScene& out, std::string& err, const DeviceBvhApi* builder, double time0, double time1, int world_spheres) {
    // (small scenes: the finest leaf size whose tree still fits the LDS form, below — host-built trees only, i.e. no device builder or RTTNW_BVH_AUTO's,
    // which leaves every tree of such a scene to the host)
    const bool small = g.objs.size() <= 8192 && (builder == nullptr || builder->min_leaves > 8192);
    const char* forced = getenv("RTTNW_MAX_LEAF"); // experiments: force the leaf size of the host SAH build (1, 2 or 4)
    for (size_t max_leaf : {size_t(1), size_t(2), size_t(4)}) {
        if (forced && *forced && size_t(atoi(forced)) != max_leaf && max_leaf != 4) continue;
        if (!small && max_leaf != 4) continue;
        out = FlatScene();
        out.time0 = time0; out.time1 = time1;
        Lowering lw(g, out, err, builder, max_leaf, time0, time1);
        lw.world_spheres_arg = world_spheres;
        const int rc = lw.run();
        out.n_world_copies = uint32_t(lw.world_spheres.size());
        out.decide_lean();
        if (rc != 0 || max_leaf == 4 || fits_lds_form(out)) return rc;
    }
    return 0;
}

} // namespace rt
