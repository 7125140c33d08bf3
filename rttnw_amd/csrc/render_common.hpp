// render_common.hpp — host-side state of the device half of include/rttnw_hip.h, shared by its translation units:
//   render_api.cpp   the extern "C" entry points, device state, rttnw_render_multi + RCCL (host code only)
//   render_f32.hip   the F32 instantiation of the kernels (trace_kernels.hpp) and of their launch code (render_tiles.hpp)
//   render_f64.hip   the F64 instantiation — a translation unit of its own because its code wants other compiler settings
//                    than the f32 code (Makefile: machine LICM off, 1024-thread blocks) and because the two halves build in
//                    parallel.
#pragma once
#include "../../include/rttnw_hip.h"
#include "rt_core.hpp"
#include "scene_handle.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <memory>
#include <chrono>
#include <mutex>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace rt {

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_));             \
            return RTTNW_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)

template <typename T> struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    template <typename A> int upload(const std::vector<T, A>& v) {
        release();
        n = v.size();
        const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 31) / 32 * 32; // (the LDS staging of small record arrays copies whole 32-byte units)
        HIP_TRY(hipMalloc((void**)&p, bytes));
        if (n) HIP_TRY(hipMemcpy(p, v.data(), n * sizeof(T), hipMemcpyHostToDevice));
        return 0;
    }
    void release() {
        if (p && !adopted) (void)hipFree(p);
        adopted.reset();
        p = nullptr;
        n = 0;
    }
    std::shared_ptr<void> adopted; // set: p is a buffer someone else made on this device (a device-built tree); kept alive, not freed here
    void adopt(const std::shared_ptr<void>& buf, size_t count) {
        release();
        adopted = buf;
        p = (T*)buf.get();
        n = count;
    }
};

template <typename R> struct DeviceScene {
    bool ready = false;
    DevBuf<Bvh4Node> nodes;
    DevBuf<Bvh4QNode> nodes4q;   // the same records with quantised boxes (made on first use by the decoupled kernels, ensure_quant4)
    DevBuf<SphereRec<R>> spheres;
    DevBuf<int32_t> sphere_mat, sphere_seq;
    DevBuf<MovingSphereRec<R>> moving;
    DevBuf<RectRec<R>> rects;
    DevBuf<BoxRec<R>> boxes;
    DevBuf<InstanceRec<R>> insts;
    DevBuf<MediumRec<R>> media;
    DevBuf<int32_t> medium_refs;
    DevBuf<MaterialRec<R>> mats;
    DevBuf<TextureRec<R>> texs;
    DevBuf<ImageRec> images;
    DevBuf<uint32_t> texels;
    DevBuf<R> perlin_vec;
    DevBuf<uint8_t> perlin_perm;
    SceneView<R> view{};
    size_t bytes = 0;
    // The decoupled kernels' own view of a big cloud (round 6, ensure_quant4): node records and the sphere records of their leaves INTERLEAVED in one
    // buffer (bvh_build.hpp interleave_build_device), the spheres' sequence numbers and materials at the same, sparse, indices; the few other records'
    // material references moved behind them.  Layout only: what `view` describes, elsewhere.
    bool interleaved = false;
    SceneView<R> view_q{};
    DevBuf<int32_t> seq_q;
    DevBuf<MaterialRec<R>> mats_q;
    DevBuf<RectRec<R>> rects_q;
    DevBuf<BoxRec<R>> boxes_q;
    const SceneView<R>& decoupled_view() const { return interleaved ? view_q : view; }

    // The scene's node array on the current device: the host-built records followed by the device-built trees.  A scene whose
    // nodes are ONE device-built tree on this very device (spheres_1m: 57 MB) simply adopts the builder's buffer; otherwise the
    // pieces are put together — device-to-device for trees built here, through the host copy (materialize_host_nodes) for
    // trees built on another device.
    int upload_nodes(const FlatScene& f) {
        int dev = -1;
        HIP_TRY(hipGetDevice(&dev));
        if (f.device_trees.empty()) return nodes.upload(f.nodes4);
        if (f.n_host4 == 0 && f.device_trees.size() == 1 && f.device_trees[0].device == dev) {
            nodes.adopt(f.device_trees[0].nodes4, f.device_trees[0].count4);
            return 0;
        }
        nodes.release();
        const size_t total = f.total_nodes4();
        HIP_TRY(hipMalloc((void**)&nodes.p, std::max<size_t>(total, 1) * sizeof(Bvh4Node)));
        nodes.n = total;
        if (f.n_host4) HIP_TRY(hipMemcpy(nodes.p, f.nodes4.data(), size_t(f.n_host4) * sizeof(Bvh4Node), hipMemcpyHostToDevice));
        for (const DeviceTree& t : f.device_trees) {
            if (t.device == dev) {
                HIP_TRY(hipMemcpy(nodes.p + t.base4, t.nodes4.get(), size_t(t.count4) * sizeof(Bvh4Node), hipMemcpyDeviceToDevice));
            } else {
                std::string err;
                if (int mrc = materialize_host_nodes(const_cast<FlatScene&>(f), err)) { set_last_error(err); return mrc; }
                HIP_TRY(hipMemcpy(nodes.p + t.base4, f.nodes4.data() + t.base4, size_t(t.count4) * sizeof(Bvh4Node), hipMemcpyHostToDevice));
            }
        }
        return 0;
    }

    // The decoupled kernels' node records, made on this device from the f32 ones the first time such a kernel is chosen.
    int ensure_quant4(const FlatScene& f) {
        if (nodes4q.p) return 0;
        {
            const char* e = getenv("RTTNW_INTERLEAVE"); // (0: the separate arrays of rounds 1-5, for A/B runs and tests)
            if (f.sphere_mat_is_index && f.insts.empty() && f.moving.empty() && f.media.empty() && !(e && e[0] == '0')) return build_interleaved(f);
        }
        const uint32_t n = f.total_nodes4();
        HIP_TRY(hipMalloc((void**)&nodes4q.p, std::max<size_t>(n, 1) * sizeof(Bvh4QNode)));
        nodes4q.n = n;
        std::string err;
        if (int rc = quant4_build_device(nodes.p, n, nodes4q.p, err)) { set_last_error(err); nodes4q.release(); return rc; }
        view.nodes4q = nodes4q.p;
        return 0;
    }


    // A big cloud (FlatScene::sphere_mat_is_index: >= 65 536 spheres, slot i holds i, no instance, medium or moving sphere): the quantised node
    // records and the spheres of their leaves in ONE buffer.  Per-record sphere counts come from the device (the tree may live only there), the
    // layout is a sequential pass on the host (a record with sphere leaves starts on a 128-byte line, any other on a 64-byte boundary), the
    // records are written by one kernel.  tests/hostsim/cache_model.hpp priced it: nodes + spheres 73.9 -> 64.7 read-miss lines per sample in
    // f32 (all four 16-byte spheres of a record share its line), 80.1 -> 73.7 in f64.
    int build_interleaved(const FlatScene& f) {
        const uint32_t n4 = f.total_nodes4();
        std::string err;
        uint8_t* d_cnt = nullptr;
        uint32_t* d_off = nullptr;
        struct Free { uint8_t*& a; uint32_t*& b; ~Free() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); } } free_tmp{d_cnt, d_off};
        HIP_TRY(hipMalloc((void**)&d_cnt, std::max<uint32_t>(n4, 1)));
        if (int rc = interleave_count_device(nodes.p, n4, d_cnt, err)) { set_last_error(err); return rc; }
        std::vector<uint8_t> cnt(n4);
        if (n4) HIP_TRY(hipMemcpy(cnt.data(), d_cnt, n4, hipMemcpyDeviceToHost));
        constexpr uint32_t su = uint32_t(sizeof(SphereRec<R>) / 16);
        std::vector<uint32_t> off(n4);
        uint64_t at = 0;
        for (uint32_t i = 0; i < n4; ++i) {
            const uint32_t align = cnt[i] ? 8u : 4u;
            at = (at + align - 1) / align * align;
            off[i] = uint32_t(at);
            at += 4u + uint32_t(cnt[i]) * su;
        }
        at = (at + 7) / 8 * 8;
        const uint64_t n_sparse = at / su;               // sphere indices of the buffer run up to here
        if (at >= (1ull << 32) || n_sparse >= (1ull << 26)) return ensure_quant4_plain(f); // (beyond the leaf bits' 26-bit record index: the separate arrays)
        HIP_TRY(hipMalloc((void**)&d_off, std::max<size_t>(n4, 1) * 4));
        if (n4) HIP_TRY(hipMemcpy(d_off, off.data(), size_t(n4) * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&nodes4q.p, size_t(at) * 16));
        nodes4q.n = size_t(at) / 4;
        HIP_TRY(hipMemset(nodes4q.p, 0, size_t(at) * 16));
        HIP_TRY(hipMalloc((void**)&seq_q.p, size_t(n_sparse) * 4)); seq_q.n = size_t(n_sparse);
        HIP_TRY(hipMemset(seq_q.p, 0, size_t(n_sparse) * 4));
        const size_t nm = f.mats.size();
        HIP_TRY(hipMalloc((void**)&mats_q.p, (size_t(n_sparse) + nm) * sizeof(MaterialRec<R>))); mats_q.n = size_t(n_sparse) + nm;
        HIP_TRY(hipMemset(mats_q.p, 0, size_t(n_sparse) * sizeof(MaterialRec<R>)));
        if (nm) HIP_TRY(hipMemcpy(mats_q.p + n_sparse, mats.p, nm * sizeof(MaterialRec<R>), hipMemcpyDeviceToDevice)); // the scene's materials, behind the spheres'
        InterleaveArgs a{};
        a.nodes4 = nodes.p; a.n4 = n4; a.noff = d_off; a.spheres = spheres.p; a.sphere_bytes = uint32_t(sizeof(SphereRec<R>)); a.sphere_seq = sphere_seq.p;
        a.mats = mats.p; a.mat_bytes = uint32_t(sizeof(MaterialRec<R>)); a.buffer = nodes4q.p; a.seq_out = seq_q.p; a.mats_out = mats_q.p;
        if (int rc = interleave_build_device(a, err)) { set_last_error(err); return rc; }
        // the other kinds' records keep their places; their material references move behind the sparse block
        std::vector<RectRec<R>> rq;
        for (auto& r : f.rects) rq.push_back({R(r.a0), R(r.a1), R(r.b0), R(r.b1), R(r.k), r.plane, r.mat + int32_t(n_sparse), r.seq});
        std::vector<BoxRec<R>> bq;
        for (auto& b : f.boxes) {
            BoxRec<R> o{};
            for (int k = 0; k < 3; ++k) { o.mn[k] = R(b.mn[k]); o.mx[k] = R(b.mx[k]); }
            o.mat = b.mat + int32_t(n_sparse); o.seq = b.seq;
            bq.push_back(o);
        }
        if (int rc = rects_q.upload(rq)) return rc;
        if (int rc = boxes_q.upload(bq)) return rc;
        view_q = view;
        view_q.nodes4q = nodes4q.p;
        view_q.spheres = reinterpret_cast<const SphereRec<R>*>(nodes4q.p);
        view_q.sphere_mat = nullptr;
        view_q.sphere_seq = seq_q.p;
        view_q.mats = mats_q.p;
        view_q.rects = rects_q.p;
        view_q.boxes = boxes_q.p;
        view_q.top_root = int32_t(off[size_t(f.top_root)] >> 2);
        view.nodes4q = nodes4q.p; // (never walked through `view`: the records' child slots are the interleaved buffer's)
        interleaved = true;
        return 0;
    }
    int ensure_quant4_plain(const FlatScene& f) {
        const uint32_t n = f.total_nodes4();
        HIP_TRY(hipMalloc((void**)&nodes4q.p, std::max<size_t>(n, 1) * sizeof(Bvh4QNode)));
        nodes4q.n = n;
        std::string err;
        if (int rc = quant4_build_device(nodes.p, n, nodes4q.p, err)) { set_last_error(err); nodes4q.release(); return rc; }
        view.nodes4q = nodes4q.p;
        return 0;
    }

    int upload(const FlatScene& f) {
        std::vector<SphereRec<R>> sp;
        for (auto& s : f.spheres) sp.push_back({R(s.cx), R(s.cy), R(s.cz), R(s.r)});
        std::vector<MovingSphereRec<R>> mv;
        for (auto& m : f.moving) {
            MovingSphereRec<R> o{};
            for (int k = 0; k < 3; ++k) { o.c0[k] = R(m.c0[k]); o.c1[k] = R(m.c1[k]); }
            o.r = R(m.r); o.t0 = R(m.t0); o.t1 = R(m.t1); o.mat = m.mat; o.seq = m.seq;
            mv.push_back(o);
        }
        std::vector<RectRec<R>> rc_;
        for (auto& r : f.rects) rc_.push_back({R(r.a0), R(r.a1), R(r.b0), R(r.b1), R(r.k), r.plane, r.mat, r.seq});
        std::vector<BoxRec<R>> bx;
        for (auto& b : f.boxes) {
            BoxRec<R> o{};
            for (int k = 0; k < 3; ++k) { o.mn[k] = R(b.mn[k]); o.mx[k] = R(b.mx[k]); }
            o.mat = b.mat; o.seq = b.seq;
            bx.push_back(o);
        }
        std::vector<InstanceRec<R>> in;
        for (auto& i : f.insts) {
            InstanceRec<R> o{};
            o.n_ops = i.n_ops; o.root = i.root; o.single_leaf = i.single_leaf;
            for (int k = 0; k < MAX_INSTANCE_OPS; ++k) {
                o.ops[k].type = i.ops[k].type;
                for (int c = 0; c < 3; ++c) o.ops[k].v[c] = R(i.ops[k].v[c]);
            }
            in.push_back(o);
        }
        std::vector<MediumRec<R>> md;
        for (auto& m : f.media) md.push_back({m.b_first, m.b_count, m.inst, m.n_outer, m.mat, m.ref0, R(m.neg_inv_density)});
        std::vector<MaterialRec<R>> mt;
        for (auto& m : f.mats) mt.push_back({m.type, m.tex, {R(m.albedo[0]), R(m.albedo[1]), R(m.albedo[2])}, R(m.param)});
        std::vector<TextureRec<R>> tx;
        for (auto& t : f.texs) tx.push_back({t.type, t.a, t.b, 0, {R(t.color[0]), R(t.color[1]), R(t.color[2])}, R(t.scale)});
        std::vector<R> pv;
        for (double v : f.perlin_vec) pv.push_back(R(v));

        int rc;
        if ((rc = upload_nodes(f)) || (rc = spheres.upload(sp)) || (!f.sphere_mat_is_index && (rc = sphere_mat.upload(f.sphere_mat))) ||
            (rc = sphere_seq.upload(f.sphere_seq)) || (rc = moving.upload(mv)) || (rc = rects.upload(rc_)) ||
            (rc = boxes.upload(bx)) || (rc = insts.upload(in)) || (rc = media.upload(md)) || (rc = medium_refs.upload(f.medium_refs)) || (rc = mats.upload(mt)) ||
            (rc = texs.upload(tx)) || (rc = images.upload(f.images)) || (rc = texels.upload(f.texels)) ||
            (rc = perlin_vec.upload(pv)) || (rc = perlin_perm.upload(f.perlin_perm)))
            return rc;
        view.nodes = nodes.p; view.nodes4q = nullptr; view.spheres = spheres.p; view.sphere_mat = f.sphere_mat_is_index ? nullptr : sphere_mat.p; view.sphere_seq = sphere_seq.p;
        view.moving = moving.p; view.rects = rects.p; view.boxes = boxes.p; view.insts = insts.p; view.media = media.p; view.medium_refs = medium_refs.p;
        view.mats = mats.p; view.texs = texs.p; view.images = images.p; view.texels = texels.p;
        view.perlin_vec = perlin_vec.p; view.perlin_perm = perlin_perm.p;
        view.top_root = f.top_root;
        view.n_media = int32_t(f.media.size());
        bytes = size_t(f.total_nodes4()) * sizeof(Bvh4Node) + sp.size() * sizeof(SphereRec<R>) + mv.size() * sizeof(MovingSphereRec<R>) +
                rc_.size() * sizeof(RectRec<R>) + bx.size() * sizeof(BoxRec<R>) + in.size() * sizeof(InstanceRec<R>);
        ready = true;
        return 0;
    }
    void release() {
        seq_q.release(); mats_q.release(); rects_q.release(); boxes_q.release(); interleaved = false;
        nodes.release(); nodes4q.release(); spheres.release(); sphere_mat.release(); sphere_seq.release(); moving.release(); rects.release();
        boxes.release(); insts.release(); media.release(); medium_refs.release(); mats.release(); texs.release(); images.release();
        texels.release(); perlin_vec.release(); perlin_perm.release();
        ready = false;
    }
};

struct DeviceState {
    int device = -1;
    int num_cus = 0;
    uint64_t chunk_budget = 0; // bytes of chunk sums a launch may hold on this device (rt_types.hpp launch_chunks): total HBM / 12, 4 .. 24 GiB
    DeviceScene<float> s32;
    DeviceScene<double> s64;
    DeviceScene<double> s64_ref; // rttnw_scene::flat_ref on this device (RTTNW_F64_STRICT renders of scenes with world-space copies)
    // workspace, grown on demand and kept
    void* partial = nullptr;
    size_t partial_bytes = 0;
    void* pool_r = nullptr; size_t pool_r_bytes = 0; // path-slot state (reals / words), SoA over all slots
    void* pool_u = nullptr; size_t pool_u_bytes = 0;
    void* spill = nullptr; size_t spill_bytes = 0;   // traversal-stack entries beyond LDS_STACK_ENTRIES, per thread of the launch
    unsigned long long* job_counter = nullptr; // [0] job counter, then DeviceCounters
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t stream = nullptr;              // rttnw_render_multi: this device's launch stream
    void* multi_packed = nullptr; size_t multi_packed_bytes = 0; // packed tiles of the logical ranks living on this device
    void* gathered = nullptr; size_t gathered_bytes = 0;         // root device: every rank's packed tiles
    // scratch for the blocking host-output render()
    void* packed = nullptr; size_t packed_bytes = 0;
    void* linear = nullptr; size_t linear_bytes = 0;
    uint8_t* rgba = nullptr; size_t rgba_bytes = 0;
};

int grow(void** p, size_t* have, size_t want); // a workspace buffer kept at its high-water mark (render_api.cpp)
void debug_print_sched(const DeviceCounters& hc, bool plain, uint32_t profile, uint64_t samples); // RTTNW_DEBUG_SCHED=1 only (debug_sched.cpp)
int device_state_create(DeviceState*& out, std::string& err);
// The lowering an RTTNW_F64_STRICT render walks: s->flat, or — when that holds world-space copies of transformed groups' spheres — the same
// graph lowered without them (made once, under the scene's mutex).  render_api.cpp.
int reference_frame_scene(::rttnw_scene* s, const FlatScene*& flat);

template <typename R> DeviceScene<R>& scene_of(DeviceState* d);
template <> inline DeviceScene<float>& scene_of<float>(DeviceState* d) { return d->s32; }
template <> inline DeviceScene<double>& scene_of<double>(DeviceState* d) { return d->s64; }

template <typename R> CameraRec<R> narrow_camera(const CameraRec<double>& c) {
    CameraRec<R> o;
    for (int k = 0; k < 3; ++k) {
        o.origin[k] = R(c.origin[k]); o.lower_left_corner[k] = R(c.lower_left_corner[k]);
        o.horizontal[k] = R(c.horizontal[k]); o.vertical[k] = R(c.vertical[k]); o.u[k] = R(c.u[k]); o.v[k] = R(c.v[k]);
    }
    o.lens_radius = R(c.lens_radius); o.open_time = R(c.open_time); o.close_time = R(c.close_time);
    return o;
}

void fill_layout(uint32_t w, uint32_t h, uint32_t world, rttnw_tile_layout& L);
int validate(::rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p);

// The launch code of one precision (render_tiles.hpp), instantiated in render_f32.hip / render_f64.hip — and, for double, a second
// time in render_f64_strict.hip in the namespace rt::ieee_strict (rt_core.hpp: the two builds of the f64 arithmetic).
inline namespace RT_ARITH_NS {
template <typename R>
int render_tiles_t(::rttnw_scene* s, DeviceState* d, const rttnw_camera_desc* cam, const rttnw_params* p, void* d_packed, hipStream_t stream,
                   rttnw_stats* stats, bool sync_for_stats = true, bool prepare_only = false);
template <typename R>
int probe_path_t(::rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, uint32_t px, uint32_t row, uint32_t sample,
                 double* out, uint32_t max_out);
template <typename R>
int untile_launch(uint32_t width, uint32_t height, uint32_t world, const void* d_gathered, void* d_linear_rgb, uint8_t* d_rgba8, hipStream_t stream);
extern template int render_tiles_t<float>(::rttnw_scene*, DeviceState*, const rttnw_camera_desc*, const rttnw_params*, void*, hipStream_t, rttnw_stats*, bool, bool);
extern template int render_tiles_t<double>(::rttnw_scene*, DeviceState*, const rttnw_camera_desc*, const rttnw_params*, void*, hipStream_t, rttnw_stats*, bool, bool);
extern template int probe_path_t<float>(::rttnw_scene*, const rttnw_camera_desc*, const rttnw_params*, uint32_t, uint32_t, uint32_t, double*, uint32_t);
extern template int probe_path_t<double>(::rttnw_scene*, const rttnw_camera_desc*, const rttnw_params*, uint32_t, uint32_t, uint32_t, double*, uint32_t);
extern template int untile_launch<float>(uint32_t, uint32_t, uint32_t, const void*, void*, uint8_t*, hipStream_t);
extern template int untile_launch<double>(uint32_t, uint32_t, uint32_t, const void*, void*, uint8_t*, hipStream_t);
} // namespace RT_ARITH_NS

#if !defined(RT_STRICT_F64)
// what render_api.cpp calls for precision RTTNW_F64_STRICT (defined by render_f64_strict.hip)
namespace ieee_strict {
template <typename R>
int render_tiles_t(::rttnw_scene* s, DeviceState* d, const rttnw_camera_desc* cam, const rttnw_params* p, void* d_packed, hipStream_t stream,
                   rttnw_stats* stats, bool sync_for_stats, bool prepare_only);
template <typename R>
int probe_path_t(::rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, uint32_t px, uint32_t row, uint32_t sample,
                 double* out, uint32_t max_out);
extern template int render_tiles_t<double>(::rttnw_scene*, DeviceState*, const rttnw_camera_desc*, const rttnw_params*, void*, hipStream_t, rttnw_stats*, bool, bool);
extern template int probe_path_t<double>(::rttnw_scene*, const rttnw_camera_desc*, const rttnw_params*, uint32_t, uint32_t, uint32_t, double*, uint32_t);
} // namespace ieee_strict
#endif

} // namespace rt
