// TEST INFRASTRUCTURE — host build of the product's per-lane tracing core (rttnw_amd/csrc/rt_core.hpp)
// and lowering, used by tests to debug/verify the kernel LOGIC in a container without a GPU and to
// cross-check the device counters.  It is never loaded by the rttnw_amd package: the product path has
// no CPU fallback.  Job order and per-job accumulation mirror trace_kernels.hpp's trace kernel exactly.
#include "../../include/rttnw_hip.h"
#include "../../rttnw_amd/csrc/rt_core.hpp"
#include "../../rttnw_amd/csrc/bvh_quant.hpp"
#include "../../rttnw_amd/csrc/scene_handle.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace rt {
int device_commit(::rttnw_scene*, std::string&) { return 0; } // no device in the host build
void device_release(DeviceState*) {}
int device_bvh_builder(::rttnw_scene*, DeviceBvhApi&, std::string& err) {
    err = "the host test build has no device BVH builder";
    return RTTNW_ERR_HIP;
}
int materialize_host_nodes(FlatScene&, std::string&) { return 0; } // no device trees without a device builder
} // namespace rt

namespace {
using namespace rt;

std::atomic<int> g_max_stack{0}; // deepest traversal-stack index written since the last hostsim_max_stack() call

struct HostStack { // shaped like the device's: 16 entries + a spare slot "in LDS", the rest in a spill strip
    static constexpr int WIDE = NODES_F32X4;
    static constexpr int SLAB_F32 = SLAB_EXACT;
    static constexpr int SPARE = 16;
    int32_t lds[17];
    int32_t spill[256];
    void set(int i, int32_t v) {
        (i < 16 ? lds[i] : spill[i - 16]) = v;
        int seen = g_max_stack.load(std::memory_order_relaxed);
        while (i + 1 > seen && !g_max_stack.compare_exchange_weak(seen, i + 1, std::memory_order_relaxed)) {}
    }
    int32_t get(int i) const { return i < 16 ? lds[i] : spill[i - 16]; }
    bool room_for_three(int i) const { return i + 3 <= 16; } // both push forms of the node step get exercised
    void set_fast(int i, int32_t v) { if (i == SPARE) lds[16] = v; else set(i, v); }
    uint32_t plane_off(uint32_t q) const { return q; }
    template <typename R> void fetch(const SceneView<R>& sc, int32_t i, const uint32_t* near_off, Planes4& out) const {
        const Bvh4Node& nd = sc.nodes[i];
        for (uint32_t a = 0; a < 3; ++a) {
            const float* nr = near_off[a] == a ? nd.lo[a] : nd.hi[a];
            const float* fr = near_off[a] == a ? nd.hi[a] : nd.lo[a];
            for (int c = 0; c < 4; ++c) { out.nr[a][c] = nr[c]; out.fr[a][c] = fr[c]; }
        }
        for (int c = 0; c < 4; ++c) out.child[c] = nd.child[c];
    }
};

struct HostStackAll : HostStack { // HOSTSIM_QUANT=3: the f32 records with every used slot's box opened to the whole space — a walk that never culls:
    // the reference the conservative box tests are held against (a box test may only ever open more nodes: same hits, same image)
    template <typename R> void fetch(const SceneView<R>& sc, int32_t i, const uint32_t* near_off, Planes4& out) const {
        const Bvh4Node& nd = sc.nodes[i];
        for (uint32_t a = 0; a < 3; ++a)
            for (int c = 0; c < 4; ++c) {
                const bool used = nd.child[c] != CHILD_EMPTY;
                const float lo = used ? -1e30f : nd.lo[a][c], hi = used ? 1e30f : nd.hi[a][c];
                out.nr[a][c] = near_off[a] == a ? lo : hi;
                out.fr[a][c] = near_off[a] == a ? hi : lo;
            }
        for (int c = 0; c < 4; ++c) out.child[c] = nd.child[c];
    }
};

struct HostStack4Q : HostStack { // the f64 decoupled kernel's: walks the quantised records (bvh_quant.hpp, made on the host here)
    static constexpr int WIDE = NODES_Q8X4;
    template <typename R> void fetch4q(const SceneView<R>& sc, int32_t i, uint32_t* w) const { std::memcpy(w, &sc.nodes4q[i], 64); }
};

template <typename R> struct HostScene {
    std::vector<SphereRec<R>> spheres;
    std::vector<MovingSphereRec<R>> moving;
    std::vector<RectRec<R>> rects;
    std::vector<BoxRec<R>> boxes;
    std::vector<InstanceRec<R>> insts;
    std::vector<MediumRec<R>> media;
    std::vector<MaterialRec<R>> mats;
    std::vector<TextureRec<R>> texs;
    std::vector<R> perlin_vec;
    SceneView<R> view;
    std::vector<Bvh4QNode> nodes4q; // HOSTSIM_QUANT=1: the f64 decoupled kernel's records, made by the same per-record function as on the device
    void make_quant4(const FlatScene& f) {
        nodes4q.resize(f.nodes4.size());
        for (size_t i = 0; i < f.nodes4.size(); ++i) quant4_make(f.nodes4.data(), int32_t(i), nodes4q[i]);
        view.nodes4q = nodes4q.data();
    }
    explicit HostScene(const FlatScene& f) {
        for (auto& s : f.spheres) spheres.push_back({R(s.cx), R(s.cy), R(s.cz), R(s.r)});
        for (auto& m : f.moving) {
            MovingSphereRec<R> o{};
            for (int k = 0; k < 3; ++k) { o.c0[k] = R(m.c0[k]); o.c1[k] = R(m.c1[k]); }
            o.r = R(m.r); o.t0 = R(m.t0); o.t1 = R(m.t1); o.mat = m.mat; o.seq = m.seq;
            moving.push_back(o);
        }
        for (auto& r : f.rects) rects.push_back({R(r.a0), R(r.a1), R(r.b0), R(r.b1), R(r.k), r.plane, r.mat, r.seq});
        for (auto& b : f.boxes) {
            BoxRec<R> o{};
            for (int k = 0; k < 3; ++k) { o.mn[k] = R(b.mn[k]); o.mx[k] = R(b.mx[k]); }
            o.mat = b.mat; o.seq = b.seq;
            boxes.push_back(o);
        }
        for (auto& i : f.insts) {
            InstanceRec<R> o{};
            o.n_ops = i.n_ops; o.root = i.root; o.single_leaf = i.single_leaf;
            for (int k = 0; k < MAX_INSTANCE_OPS; ++k) {
                o.ops[k].type = i.ops[k].type;
                for (int c = 0; c < 3; ++c) o.ops[k].v[c] = R(i.ops[k].v[c]);
            }
            insts.push_back(o);
        }
        for (auto& m : f.media) media.push_back({m.b_first, m.b_count, m.inst, m.n_outer, m.mat, m.ref0, R(m.neg_inv_density)});
        for (auto& m : f.mats) mats.push_back({m.type, m.tex, {R(m.albedo[0]), R(m.albedo[1]), R(m.albedo[2])}, R(m.param)});
        for (auto& t : f.texs) texs.push_back({t.type, t.a, t.b, 0, {R(t.color[0]), R(t.color[1]), R(t.color[2])}, R(t.scale)});
        for (double v : f.perlin_vec) perlin_vec.push_back(R(v));
        view.nodes = f.nodes4.data();
        view.nodes4q = nullptr;
        view.spheres = spheres.data(); view.sphere_mat = f.sphere_mat_is_index ? nullptr : f.sphere_mat.data(); view.sphere_seq = f.sphere_seq.data();
        view.moving = moving.data(); view.rects = rects.data(); view.boxes = boxes.data();
        view.insts = insts.data(); view.media = media.data(); view.medium_refs = f.medium_refs.data(); view.mats = mats.data(); view.texs = texs.data();
        view.images = f.images.data(); view.texels = f.texels.data();
        view.perlin_vec = perlin_vec.data(); view.perlin_perm = f.perlin_perm.data();
        view.top_root = f.top_root; view.n_media = int32_t(media.size());
    }
};

template <typename R> CameraRec<R> narrow_camera(const CameraRec<double>& c) {
    CameraRec<R> o;
    for (int k = 0; k < 3; ++k) {
        o.origin[k] = R(c.origin[k]); o.lower_left_corner[k] = R(c.lower_left_corner[k]);
        o.horizontal[k] = R(c.horizontal[k]); o.vertical[k] = R(c.vertical[k]); o.u[k] = R(c.u[k]); o.v[k] = R(c.v[k]);
    }
    o.lens_radius = R(c.lens_radius); o.open_time = R(c.open_time); o.close_time = R(c.close_time);
    return o;
}

template <typename R, typename StackT>
int render_t(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, double* out_linear, rttnw_stats* stats,
             int n_threads) {
    HostScene<R> hs(s->flat);
    if (StackT::WIDE == NODES_Q8X4) hs.make_quant4(s->flat);
    CameraRec<double> cam64;
    make_camera(cam->lookfrom, cam->lookat, cam->view_up, cam->vertical_fov, cam->aspect_ratio, cam->aperture,
                cam->focus_distance, cam->open_time, cam->close_time, cam64);
    CameraRec<R> camr = narrow_camera<R>(cam64);
    RenderConsts rc{};
    rc.width = p->width; rc.height = p->height; rc.spp = p->spp; rc.max_depth = p->max_depth;
    rc.sample_begin = p->sample_begin;
    plan_chunks(rc, p->spp, p->spp_chunk); // the render's chunk schedule: a function of spp alone (rt_types.hpp)
    rc.quirks = p->quirks; rc.seed = p->seed; rc.stack_depth = s->flat.stack_depth;
    V3<R> background(R(p->background[0]), R(p->background[1]), R(p->background[2]));
    const R t_min = R(p->t_min);
    if (n_threads <= 0) n_threads = int(std::thread::hardware_concurrency());
    std::atomic<uint32_t> next_row{0};
    std::vector<LaneCounters> counters(n_threads);
    auto worker = [&](int tid) {
        StackT stack;
        LaneCounters& cnt = counters[tid];
        for (;;) {
            uint32_t row = next_row.fetch_add(1);
            if (row >= rc.height) break;
            for (uint32_t px = 0; px < rc.width; ++px) {
                V3<R> total; // one chain per pixel: the chunk sums added in chunk order, however the device splits the render into launches
                for (uint32_t c = 0; c < rc.n_chunks; ++c) { // one job = (pixel, chunk), folded sequentially
                    V3<R> acc;
                    uint32_t s0, s1;
                    chunk_samples(rc, c, s0, s1);
                    for (uint32_t si = s0; si < s1; ++si) {
                        PathState<R> ps;
                        path_begin(ps, camr, rc, px, row, si);
                        while (path_step(ps, hs.view, rc, background, t_min, stack, cnt)) {}
                        acc = acc + ps.radiance;
                    }
                    total = total + acc;
                }
                V3<R> mean = total / R(rc.spp);
                size_t o = (size_t(row) * rc.width + px) * 3;
                out_linear[o] = double(mean.x); out_linear[o + 1] = double(mean.y); out_linear[o + 2] = double(mean.z);
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto& t : th) t.join();
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->samples = uint64_t(rc.width) * rc.height * rc.spp;
        for (auto& c : counters) {
            stats->rays += c.rays; stats->nodes_visited += c.nodes; stats->prims_tested += c.prims; stats->texel_fetches += c.texels;
        }
        stats->n_nodes = uint32_t(s->flat.nodes4.size());
        stats->n_prims = s->flat.n_prims_in_bvh;
    }
    return RTTNW_OK;
}
} // namespace

template <typename R>
static int probe_path_t(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, uint32_t px, uint32_t row,
                        uint32_t sample, double* out, uint32_t max_out) {
    HostScene<R> hs(s->flat);
    CameraRec<double> cam64;
    make_camera(cam->lookfrom, cam->lookat, cam->view_up, cam->vertical_fov, cam->aspect_ratio, cam->aperture,
                cam->focus_distance, cam->open_time, cam->close_time, cam64);
    CameraRec<R> camr = narrow_camera<R>(cam64);
    RenderConsts rc{};
    rc.width = p->width; rc.height = p->height; rc.spp = p->spp; rc.max_depth = p->max_depth; rc.quirks = p->quirks; rc.seed = p->seed;
    PathState<R> ps;
    path_begin(ps, camr, rc, px, row, sample);
    HostStack stack; NoCounters cnt;
    uint32_t n = 0;
    while (n < max_out) { // same record layout as rttnw_debug_probe_path (include/rttnw_hip.h): 20 doubles per bounce
        HitRecord<R> rec;
        const Ray<R> ray = ps.ray;
        if (!world_hit(hs.view, ps.ray, R(p->t_min), ps.key, ps.bounce, rc.quirks, rec, stack, cnt)) break;
        double* o = out + size_t(n) * 20;
        o[0] = rec.t; o[1] = rec.p.x; o[2] = rec.p.y; o[3] = rec.p.z; o[4] = rec.normal.x; o[5] = rec.normal.y; o[6] = rec.normal.z;
        o[7] = double(rec.mat); o[8] = rec.u; o[9] = rec.v; o[10] = rec.front_face ? 1.0 : 0.0;
        o[11] = ray.o.x; o[12] = ray.o.y; o[13] = ray.o.z; o[14] = ray.d.x; o[15] = ray.d.y; o[16] = ray.d.z; o[17] = ray.time;
        ++n;
        V3<R> att, em;
        const bool cont = shade(hs.view, rec, ps.key, ps.bounce, ps.ray, att, em, cnt);
        o[18] = em.x; o[19] = cont ? att.x : -1.0;
        if (!cont) break;
        ps.bounce += 1;
        if (ps.bounce >= rc.max_depth) break;
    }
    return int(n);
}

// Walk-length statistics of a render: hist[k] = walks that took k trips round the walk loop of `node_steps` node steps + a
// leaf step (what a wave's lanes do in lockstep); out2 = {walks, node visits, record tests}.  Experiment support.
static void walk_histogram(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, int node_steps, uint64_t* hist,
                           uint64_t* out2) {
    HostScene<float> hs(s->flat);
    CameraRec<double> cam64;
    make_camera(cam->lookfrom, cam->lookat, cam->view_up, cam->vertical_fov, cam->aspect_ratio, cam->aperture,
                cam->focus_distance, cam->open_time, cam->close_time, cam64);
    CameraRec<float> camr = narrow_camera<float>(cam64);
    RenderConsts rc{};
    rc.width = p->width; rc.height = p->height; rc.spp = p->spp; rc.max_depth = p->max_depth; rc.quirks = p->quirks; rc.seed = p->seed;
    V3<float> background(float(p->background[0]), float(p->background[1]), float(p->background[2]));
    HostStack stack; LaneCounters cnt;
    const int mode = node_steps / 100; // experiment: 1 = test the previous hit's primitive first when it was inside an instance
    node_steps %= 100;
    for (uint32_t row = 0; row < rc.height; ++row)
        for (uint32_t px = 0; px < rc.width; ++px)
            for (uint32_t si = 0; si < rc.spp; ++si) {
                PathState<float> ps;
                path_begin(ps, camr, rc, px, row, si);
                HitRef prev; prev.prim = make_ref(PRIM_NONE, 0); prev.inst = -1; prev.aux = 0;
                for (;;) {
                    Trav<float> tr;
                    trav_begin(tr, hs.view, ps.ray, stack);
                    if (mode == 1 && prev.inst >= 0 && ref_kind(prev.prim) != PRIM_NONE) {
                        const Ray<float> saved = tr.ray;
                        tr.ray = to_object<true>(hs.view.insts[prev.inst], ps.ray);
                        tr.cur_inst = prev.inst;
                        trav_test_record(tr, hs.view, ref_kind(prev.prim), ref_index(prev.prim), float(p->t_min), tr.ray, tr.cur_inst);
                        tr.ray = saved; tr.cur_inst = -1;
                    }
                    uint32_t trips = 0;
                    while (tr.node != TRAV_DONE) {
                        ++trips;
                        for (int k = 0; k < node_steps; ++k)
                            if (tr.node >= 0) trav_node_step(tr, hs.view, ps.ray, float(p->t_min), stack, cnt);
                        if (tr.node < 0 && tr.node != TRAV_DONE) trav_leaf_step(tr, hs.view, ps.ray, float(p->t_min), stack, cnt);
                    }
                    hist[trips < 63 ? trips : 63]++;
                    out2[0]++;
                    if (tr.found && tr.best.inst >= 0) { out2[3] += trips; out2[4]++; }
                    prev = tr.best; if (!tr.found) prev.inst = -1;
                    if (!path_shade(ps, hs.view, rc, background, float(p->t_min), tr.found, tr.closest, tr.best, cnt)) break;
                }
            }
    out2[1] = cnt.nodes; out2[2] = cnt.prims;
}


// Lockstep model of the lane-owns-path kernel (trace_kernels.hpp trace_kernel_plain): `n_waves` waves of 64 lanes, each folding
// `jobs_per_wave` consecutive jobs the way the kernel does (job hand-out, path begin, one walk + shade per round), the walk driven by a
// POLICY that decides, wave-uniformly, which kind of step runs next.  Experiment support: it counts what a policy would cost before
// anything is built for the device.  out: [0] rounds, [1] node-step executions, [2] node lane-steps, [3] leaf-step executions,
// [4] leaf lane-steps, [5] samples, [6..6+16) leaf executions by the SET of record kinds served (bit k = kind k), [32] walks.
// policy 0: trips of `a` node steps + one leaf step (the kernel's).  policy 1: vote — a leaf step when leaf lanes * 256 >= b * (node + leaf
// lanes) or no lane is at a node, else a node step.  policy 2: node steps until fewer than `a` lanes are at nodes (or none), then leaf steps
// until fewer than `b` lanes are at leaves.
template <typename R>
static void wave_model(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, int policy, int a, int b, uint32_t n_waves,
                       uint32_t jobs_per_wave, uint64_t* out) {
    HostScene<R> hs(s->flat);
    CameraRec<double> cam64;
    make_camera(cam->lookfrom, cam->lookat, cam->view_up, cam->vertical_fov, cam->aspect_ratio, cam->aperture,
                cam->focus_distance, cam->open_time, cam->close_time, cam64);
    CameraRec<R> camr = narrow_camera<R>(cam64);
    RenderConsts rc{};
    rc.width = p->width; rc.height = p->height; rc.spp = p->spp; rc.max_depth = p->max_depth; rc.quirks = p->quirks; rc.seed = p->seed;
    rc.sample_begin = p->sample_begin;
    plan_chunks(rc, p->spp, p->spp_chunk);
    rc.tiles_x = (rc.width + 7) / 8; rc.tiles_y = (rc.height + 7) / 8; rc.n_tiles = rc.tiles_x * rc.tiles_y;
    rc.tile_rank = 0; rc.tile_world = 1; rc.my_tiles = rc.n_tiles;
    rc.div_tiles_x = make_fastdiv(std::max<uint32_t>(1u, rc.tiles_x));
    plan_jobs(rc);
    V3<R> background(R(p->background[0]), R(p->background[1]), R(p->background[2]));
    const R t_min = R(p->t_min);
    struct Lane {
        HostStack stack; Trav<R> tr; PathState<R> ps;
        bool alive = false, done = false;
        uint32_t px = 0, row = 0, s = 0, s_end = 0;
    };
    NoCounters cnt;
    std::vector<Lane> L(64);
    const uint32_t leaf_policy = jobs_per_wave >> 28;
    jobs_per_wave &= 0x0FFFFFFFu;
    if (policy == 4) {
        // ASYNCHRONOUS SHADE (round 5): a lane whose walk is finished WAITS; the wave runs its shade phase (path_shade for the waiting lanes, then job
        // hand-out + path begin for those whose path ended, then trav_begin) only once `b` lanes wait or no lane walks — the others keep walking through
        // it.  Between shade phases: the kernel's trips of `a` node steps + a leaf step.  out[0] counts SHADE PHASES, out[33] the lanes they served,
        // out[34] shade phases that began paths.
        std::vector<uint8_t> walking(64);
        for (uint32_t w = 0; w < n_waves; ++w) {
            uint64_t next = uint64_t(w) * (rc.n_jobs / n_waves) / 64 * 64, end = next + jobs_per_wave;
            for (auto& l : L) { l.alive = false; l.done = false; l.s = l.s_end = 0; }
            std::fill(walking.begin(), walking.end(), uint8_t(0));
            for (;;) {
                uint32_t n_walk = 0, n_wait = 0;
                for (uint32_t i = 0; i < 64; ++i) {
                    if (L[i].done) continue;
                    if (walking[i] && L[i].tr.node == TRAV_DONE) walking[i] = 0; // its walk is over: waits for the shade phase
                    if (walking[i]) ++n_walk; else ++n_wait;
                }
                if (n_walk + n_wait == 0) break;
                if (n_walk == 0 || n_wait >= uint32_t(b)) {
                    bool began = false;
                    uint32_t served = 0, phase_cls = 0;
                    for (uint32_t i = 0; i < 64; ++i) {
                        Lane& l = L[i];
                        if (l.done || walking[i]) continue;
                        ++served;
                        if (l.alive) {
                            // which material branches of shade() this phase runs (out[50 + class]: phases with the class present, out[56 + class]: lanes):
                            // 0 lambertian / isotropic / light with a solid colour, 1 metal, 2 dielectric, 3 noise texture, 4 image texture, 5 checker
                            if (l.tr.found) {
                                HitRecord<R> rec;
                                NoCounters c2;
                                PathState<R> tmp = l.ps;
                                if (world_hit_finish(hs.view, tmp.ray, t_min, tmp.key, tmp.bounce, rc.quirks, l.tr.found, l.tr.closest, l.tr.best, rec, c2)) {
                                    const MaterialRec<R>& m = hs.view.mats[rec.mat];
                                    uint32_t cls = m.type == MAT_METAL ? 1u : (m.type == MAT_DIELECTRIC ? 2u : 0u);
                                    if (cls == 0u && m.tex >= 0) { const int32_t tt = hs.view.texs[m.tex].type; cls = tt == TEX_NOISE ? 3u : (tt == TEX_IMAGE ? 4u : (tt == TEX_CHECKER ? 5u : 0u)); }
                                    if (!(phase_cls >> cls & 1u)) { phase_cls |= 1u << cls; out[50 + cls]++; }
                                    out[56 + cls]++;
                                }
                            }
                            l.alive = path_shade(l.ps, hs.view, rc, background, t_min, l.tr.found, l.tr.closest, l.tr.best, cnt);
                            if (!l.alive) { ++l.s; out[5]++; }
                        }
                        while (!l.alive && l.s >= l.s_end) {
                            if (next >= end || next >= rc.n_jobs) { l.done = true; break; }
                            const JobInfo ji = job_decode(rc, uint32_t(next++));
                            l.px = ji.px; l.row = ji.row; l.s = ji.s; l.s_end = ji.s_end;
                        }
                        if (l.done) continue;
                        if (!l.alive) { path_begin(l.ps, camr, rc, l.px, l.row, l.s); l.alive = true; began = true; }
                        trav_begin(l.tr, hs.view, l.ps.ray, l.stack);
                        walking[i] = 1;
                        out[32]++;
                    }
                    out[0]++; out[33] += served; out[34] += began;
                    continue;
                }
                for (int k = 0; k < a; ++k) {
                    uint32_t n = 0;
                    for (uint32_t i = 0; i < 64; ++i) if (!L[i].done && walking[i] && L[i].tr.node >= 0) {
                        trav_node_step(L[i].tr, hs.view, L[i].ps.ray, t_min, L[i].stack, cnt); ++n;
                        // HOSTSIM_SPHERE_PRECULL=<pct> (experiment): the one-sphere leaf a node step has just DESCENDED into is pre-tested against the sphere its
                        // box implies (modelled: the record's sphere, radius inflated by pct / 1000) and skipped when the ray misses it; out[35] counts the skips.
                        // (=<pct>,all: leaves taken off the stack are pre-tested too — what a test in the leaf step's place could do with the box at hand)
                        static const char* pre = getenv("HOSTSIM_SPHERE_PRECULL");
                        if (pre) {
                            static const double infl = 1.0 + std::atof(pre) * 1e-3;
                            static const bool all = std::strstr(pre, "all") != nullptr;
                            Trav<R>& tr = L[i].tr;
                            while (tr.node < 0 && tr.node != TRAV_DONE && tr.node != CHILD_EMPTY && leaf_kind(tr.node) == PRIM_SPHERE && leaf_count(tr.node) == 1) {
                                const SphereRec<R> sp = hs.view.spheres[leaf_first(tr.node)];
                                R tt;
                                if (sphere_t(V3<R>(sp.cx, sp.cy, sp.cz), sp.r * R(infl), L[i].ps.ray, t_min, tr.closest, tt)) break;
                                out[35]++;
                                trav_pop(tr, L[i].ps.ray, L[i].stack);
                                if (!all) break;
                            }
                        }
                    }
                    if (n) { out[1]++; out[2] += n; }
                }
                // the leaf step.  jobs_per_wave's top bits select a LEAF POLICY (experiments): 0 every lane at a leaf is served (the kernel's); 1 only the
                // record kind with the most lanes waiting is served (the others wait for a later trip) unless no lane is at a node any more; 2 a kind is
                // served when >= 8 lanes wait at it, or no lane is at a node, else it waits.  out[6 + set]: leaf steps by the set of kinds they served,
                // out[40 + k]: lane-steps served of kind k.
                uint32_t cnt_k[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n_node_lanes = 0;
                for (uint32_t i = 0; i < 64; ++i) {
                    if (L[i].done || !walking[i]) continue;
                    const int32_t nd = L[i].tr.node;
                    if (nd >= 0) ++n_node_lanes;
                    else if (nd != TRAV_DONE) ++cnt_k[nd == CHILD_EMPTY ? 7u : (leaf_kind(nd) & 7u)];
                }
                uint32_t serve = 0xFFu;
                if (leaf_policy == 1 && n_node_lanes != 0) {
                    uint32_t best = 0;
                    for (uint32_t k = 1; k < 8; ++k) if (cnt_k[k] > cnt_k[best]) best = k;
                    serve = 1u << best;
                } else if (leaf_policy == 2 && n_node_lanes != 0) {
                    serve = 0;
                    for (uint32_t k = 0; k < 8; ++k) if (cnt_k[k] >= 8u || k == 7u) serve |= 1u << k;
                } else if (leaf_policy == 15 && n_node_lanes != 0) { // thresholds for the two classes from HOSTSIM_LEAF_THR="cheap,expensive"
                    uint32_t ta = 1, tb = 6;
                    if (const char* e = getenv("HOSTSIM_LEAF_THR")) sscanf(e, "%u,%u", &ta, &tb);
                    serve = 1u << 7;
                    if (cnt_k[PRIM_SPHERE] + cnt_k[PRIM_RECT] >= ta) serve |= (1u << PRIM_SPHERE) | (1u << PRIM_RECT);
                    if (cnt_k[PRIM_MOVING_SPHERE] + cnt_k[PRIM_BOX] + cnt_k[PRIM_INSTANCE] >= tb) serve |= (1u << PRIM_MOVING_SPHERE) | (1u << PRIM_BOX) | (1u << PRIM_INSTANCE);
                } else if (leaf_policy >= 3 && n_node_lanes != 0) {
                    // two classes: spheres, rectangles and empty slots are always served; the EXPENSIVE kinds (moving sphere, cube, instance) together once
                    // `leaf_policy` lanes wait at them (or no lane is at a node)
                    serve = (1u << PRIM_SPHERE) | (1u << PRIM_RECT) | (1u << 7);
                    if (cnt_k[PRIM_MOVING_SPHERE] + cnt_k[PRIM_BOX] + cnt_k[PRIM_INSTANCE] >= leaf_policy) serve |= (1u << PRIM_MOVING_SPHERE) | (1u << PRIM_BOX) | (1u << PRIM_INSTANCE);
                }
                uint32_t n = 0, kinds = 0;
                for (uint32_t i = 0; i < 64; ++i) if (!L[i].done && walking[i] && L[i].tr.node < 0 && L[i].tr.node != TRAV_DONE) {
                    const uint32_t k = L[i].tr.node == CHILD_EMPTY ? 7u : (leaf_kind(L[i].tr.node) & 7u);
                    if (!(serve >> k & 1u)) continue;
                    if (k < 5) { kinds |= 1u << k; out[40 + k]++; }
                    trav_leaf_step(L[i].tr, hs.view, L[i].ps.ray, t_min, L[i].stack, cnt); ++n;
                }
                if (n) { out[3]++; out[4] += n; out[6 + (kinds & 15u)]++; out[48 + ((kinds >> 4) & 1u)]++; }
            }
        }
        return;
    }
    for (uint32_t w = 0; w < n_waves; ++w) {
        uint64_t next = uint64_t(w) * (rc.n_jobs / n_waves) / 64 * 64, end = next + jobs_per_wave;
        for (auto& l : L) { l.alive = false; l.done = false; l.s = l.s_end = 0; }
        for (;;) {
            bool any = false;
            for (auto& l : L) {
                while (!l.done && !l.alive && l.s >= l.s_end) {
                    if (next >= end || next >= rc.n_jobs) { l.done = true; break; }
                    const JobInfo ji = job_decode(rc, uint32_t(next++));
                    l.px = ji.px; l.row = ji.row; l.s = ji.s; l.s_end = ji.s_end;
                }
                if (l.done) continue;
                any = true;
                if (!l.alive) { path_begin(l.ps, camr, rc, l.px, l.row, l.s); l.alive = true; }
                trav_begin(l.tr, hs.view, l.ps.ray, l.stack);
                out[32]++;
            }
            if (!any) break;
            out[0]++;
            // ---- the walk, in lockstep
            auto at_node = [&](const Lane& l) { return !l.done && l.tr.node >= 0; };
            auto at_leaf = [&](const Lane& l) { return !l.done && l.tr.node < 0 && l.tr.node != TRAV_DONE; };
            auto node_step = [&]() {
                uint32_t n = 0;
                for (auto& l : L) if (at_node(l)) { trav_node_step(l.tr, hs.view, l.ps.ray, t_min, l.stack, cnt); ++n; }
                if (n) { out[1]++; out[2] += n; }
                return n;
            };
            auto leaf_step = [&]() {
                uint32_t n = 0, kinds = 0;
                for (auto& l : L) if (at_leaf(l)) {
                    if (l.tr.node != CHILD_EMPTY) kinds |= 1u << (leaf_kind(l.tr.node) & 15u);
                    trav_leaf_step(l.tr, hs.view, l.ps.ray, t_min, l.stack, cnt); ++n;
                }
                if (n) { out[3]++; out[4] += n; out[6 + (kinds & 15u)]++; }
                return n;
            };
            for (;;) {
                uint32_t nn = 0, nl = 0;
                for (auto& l : L) { nn += at_node(l); nl += at_leaf(l); }
                if (nn + nl == 0) break;
                if (policy == 0) {
                    for (int k = 0; k < a; ++k) node_step();
                    leaf_step();
                } else if (policy == 3) {
                    // the kernel's trip with a leaf STASHED inside the trip: a lane that arrives at a one-record leaf in a node step but the last puts it
                    // aside, takes its next pending subtree and goes on with node steps; the trip's leaf step tests the stashed record (and the leaf the
                    // lane stands at waits for the next trip).  `a` node steps per trip.
                    std::vector<int32_t> stash(64, 0);
                    for (int k = 0; k < a; ++k) {
                        node_step();
                        if (k + 1 < a) {
                            uint32_t li = 0;
                            for (auto& l : L) {
                                const uint32_t me = li++;
                                if (l.done || stash[me] != 0) continue;
                                const int32_t nd = l.tr.node;
                                if (nd < 0 && nd != TRAV_DONE && nd != CHILD_EMPTY && leaf_kind(nd) != PRIM_INSTANCE && leaf_count(nd) == 1) {
                                    stash[me] = nd;
                                    trav_pop(l.tr, l.ps.ray, l.stack);
                                }
                            }
                        }
                    }
                    uint32_t n = 0, li = 0;
                    for (auto& l : L) {
                        const uint32_t me = li++;
                        if (l.done) continue;
                        if (stash[me] != 0) { cnt.prim(); trav_test_record(l.tr, hs.view, leaf_kind(stash[me]), leaf_first(stash[me]), t_min, l.tr.ray, l.tr.cur_inst); ++n; }
                        else if (at_leaf(l)) { trav_leaf_step(l.tr, hs.view, l.ps.ray, t_min, l.stack, cnt); ++n; }
                    }
                    if (n) { out[3]++; out[4] += n; }
                } else if (policy == 1) {
                    if (nn == 0 || (nl != 0 && nl * 256u >= uint32_t(b) * (nn + nl))) leaf_step(); else node_step();
                } else {
                    while (nn >= uint32_t(a) || (nn && !nl)) { node_step(); nn = nl = 0; for (auto& l : L) { nn += at_node(l); nl += at_leaf(l); } if (!nn) break; }
                    for (;;) { nn = nl = 0; for (auto& l : L) { nn += at_node(l); nl += at_leaf(l); } if (nl == 0 || (nl < uint32_t(b) && nn >= uint32_t(a))) break; leaf_step(); }
                    nn = nl = 0; for (auto& l : L) { nn += at_node(l); nl += at_leaf(l); }
                    if (nn && nn < uint32_t(a) && nl && nl < uint32_t(b)) { node_step(); leaf_step(); } // neither threshold met: a plain trip
                }
            }
            // ---- shade
            for (auto& l : L) {
                if (l.done) continue;
                l.alive = path_shade(l.ps, hs.view, rc, background, t_min, l.tr.found, l.tr.closest, l.tr.best, cnt);
                if (!l.alive) { ++l.s; out[5]++; }
            }
        }
    }
}

// The fast cube test (rt_core.hpp box_t_fast: one exact quotient where the ray is clear of the box's edges) against the six-rectangle test it
// replaces (box_t), on n generated cases in the host build's arithmetic (IEEE, nothing contracted = the reference's): random boxes from 1e-3 to 1e4,
// origins far, near, ON a face plane (the ray that has just scattered off the cube) and inside; directions at the box, at its corners / edges /
// face points displaced by 0 .. 1e-3 of the box, random, axis-parallel; ranges open, or ending / starting at, one ulp off, or 1e-9 off the exact t of
// a face.  out = {mismatches, verdict 0, verdict 1, verdict 2, hits, index of the first mismatch, inverted / flat boxes among the cases}.
template <typename R, int FORM> static void box_fast_check_t(uint64_t n, uint64_t seed, uint64_t* out) {
    uint64_t st = (seed ^ 0xD1B54A32D192ED03ull) * 0xBF58476D1CE4E5B9ull; // (streams of different seeds must not be shifts of one another)
    st = (st ^ (st >> 29)) * 0x94D049BB133111EBull + seed;
    auto next = [&]() { uint64_t z = (st += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    auto uni = [&]() { return double(next() >> 11) * (1.0 / 9007199254740992.0); };
    auto sym = [&]() { return 2.0 * uni() - 1.0; };
    const double eps_set[8] = {0.0, 1e-16, 1e-14, 1e-12, 1e-9, 1e-6, 1e-4, 1e-3};
    out[0] = out[1] = out[2] = out[3] = out[4] = out[6] = 0; out[5] = ~0ull;
    for (uint64_t i = 0; i < n; ++i) {
        const double scale = std::pow(10.0, std::floor(uni() * 7.0) - 3.0);
        BoxRec<R> bx{};
        double c[3], h[3];
        for (int a = 0; a < 3; ++a) {
            c[a] = sym() * 10.0 * scale;
            h[a] = scale * std::pow(10.0, uni() * 4.0 - 3.0);
            bx.mn[a] = R(c[a] - h[a]); bx.mx[a] = R(c[a] + h[a]);
            if (!(bx.mn[a] < bx.mx[a])) bx.mx[a] = std::nextafter(bx.mn[a], R(1e30));
        }
        // one case in eight: a cube from corners that are not min / max on one or two axes (Cube::new takes any two points, hittable.rs:551-558:
        // its rectangles' half-open ranges are then empty on that axis) or that coincide on an axis (zero thickness) — out[6] counts them
        if ((next() & 7u) == 0u) {
            const uint32_t what = uint32_t(next() % 3), k1 = uint32_t(next() % 3), k2 = uint32_t(next() % 3);
            if (what == 0) std::swap(bx.mn[k1], bx.mx[k1]);
            else if (what == 1) { std::swap(bx.mn[k1], bx.mx[k1]); if (k2 != k1) std::swap(bx.mn[k2], bx.mx[k2]); }
            else bx.mx[k1] = bx.mn[k1];
            out[6] += 1;
        }
        double o[3], d[3];
        const uint32_t mode = uint32_t(next() % 10);
        for (int a = 0; a < 3; ++a) o[a] = c[a] + sym() * scale * std::pow(10.0, uni() * 3.0);
        if (mode == 6 || mode == 7) { // the origin ON a face plane (exactly, or a few ulps off), the other two coordinates inside or just outside the face
            const int k = int(next() % 3);
            for (int a = 0; a < 3; ++a) o[a] = double(bx.mn[a]) + uni() * (double(bx.mx[a]) - double(bx.mn[a])) * (uni() < 0.8 ? 1.0 : 1.3);
            o[k] = (next() & 1) ? double(bx.mx[k]) : double(bx.mn[k]);
            if (uni() < 0.3) o[k] += sym() * 1e-15 * std::fabs(o[k]);
        } else if (mode == 8) { // inside
            for (int a = 0; a < 3; ++a) o[a] = double(bx.mn[a]) + uni() * (double(bx.mx[a]) - double(bx.mn[a]));
        }
        // a target point: per axis at mn, at mx, between, or beside the box; displaced by eps x extent
        double tg[3];
        for (int a = 0; a < 3; ++a) {
            const uint32_t pick = uint32_t(next() % 5);
            const double lo = double(bx.mn[a]), hi = double(bx.mx[a]);
            tg[a] = pick == 0 ? lo : pick == 1 ? hi : pick == 4 ? lo + (uni() * 1.6 - 0.3) * (hi - lo) : lo + uni() * (hi - lo);
            tg[a] += sym() * eps_set[next() % 8] * (hi - lo);
        }
        const double len = std::pow(10.0, uni() * 3.0 - 2.0);
        for (int a = 0; a < 3; ++a) d[a] = (tg[a] - o[a]) * len;
        if (mode == 6 || mode == 7 || mode == 8) { if (uni() < 0.7) for (int a = 0; a < 3; ++a) d[a] = sym(); }
        if (mode == 9) { // axis-parallel and nearly axis-parallel rays
            for (int a = 0; a < 3; ++a) if (uni() < 0.5) d[a] = uni() < 0.5 ? 0.0 : d[a] * 1e-12;
        }
        if (d[0] == 0 && d[1] == 0 && d[2] == 0) d[2] = 1.0;
        Ray<R> ray;
        ray.o = V3<R>(R(o[0]), R(o[1]), R(o[2])); ray.d = V3<R>(R(d[0]), R(d[1]), R(d[2])); ray.time = R(0);
        R t_min = uni() < 0.8 ? R(0.001) : R(0), t_max = Lim<R>::max();
        const uint32_t rsel = uint32_t(next() % 8);
        if (rsel < 5) { // a range that ends (or starts) at, beside, or near the exact t of one face
            const int f = int(next() % 6), k = f >> 1;
            const R oo[3] = {ray.o.x, ray.o.y, ray.o.z}, dd[3] = {ray.d.x, ray.d.y, ray.d.z};
            R tf = ((f & 1) ? bx.mx[k] : bx.mn[k]) - oo[k];
            tf = tf / dd[k];
            const uint32_t v = uint32_t(next() % 6);
            R lim = v == 0 ? tf : v == 1 ? std::nextafter(tf, R(-1e30)) : v == 2 ? std::nextafter(tf, R(1e30)) : v == 3 ? tf * R(1 + 1e-6) : v == 4 ? tf * R(1 - 1e-6) : tf * R(uni() * 2.0);
            if (rsel < 4) t_max = lim; else t_min = lim;
        } else if (rsel == 5) t_max = R(std::pow(10.0, uni() * 6.0 - 3.0) * scale);
        const SlabRay<R> sr = slab_ray<(sizeof(R) == 8 ? -1 : int(SLAB_EXACT))>(ray.o, ray.d);
        R t1 = R(-1), t2 = R(-1);
        int f1 = -1, f2 = -1, axis = 0;
        bool use_mx = false;
        const bool h1 = box_t(bx, ray, t_min, t_max, t1, f1);
        const bool h2 = box_t_fast<FORM>(bx, ray, sr, t_min, t_max, t2, f2);
        out[1 + box_classify<FORM>(bx, ray, sr, t_min, t_max, axis, use_mx)] += 1;
        out[4] += h1 ? 1 : 0;
        const bool same = h1 == h2 && (!h1 || (std::memcmp(&t1, &t2, sizeof(R)) == 0 && f1 == f2));
        if (!same) { out[0] += 1; if (out[5] == ~0ull) out[5] = i; }
    }
}
#include "cache_model.hpp"

extern "C" {
// L2 model of the decoupled kernel on a tree in HBM (cache_model.hpp).  prm: the Params fields in order (cache_bytes as prm[6] | prm[7] << 32
// after the six leading words), node_perm / sphere_perm: where a record lies (null: where it lies today), nodes_out: optional dump
// [n][4] of the 4-wide records' child slots for whoever computes a layout.
void hostsim_cache_model(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, const uint32_t* prm, const uint32_t* node_perm,
                         const uint32_t* sphere_perm, uint64_t* out) {
    cache_model::Params P{};
    P.n_waves = prm[0]; P.ways = prm[1]; P.node_steps = prm[2]; P.retire = prm[3]; P.warm_samples_per_slot = prm[4]; P.measure_samples_per_slot = prm[5];
    P.cache_bytes = uint64_t(prm[6]) | (uint64_t(prm[7]) << 32);
    P.sphere_bytes = prm[8]; P.mat_bytes = prm[9]; P.mat_by_sphere = prm[10]; P.real_bytes = prm[11]; P.precull = prm[12]; P.precull_pct = prm[13];
    P.xcds = prm[14]; P.node_bytes = prm[15]; P.batch_offset = prm[16]; P.unified = prm[17]; P.leaf_threshold = prm[18]; P.cold_words = prm[19] ? prm[19] : 5u;
    if (p->precision == RTTNW_F32) cache_model::run<float>(s, cam, p, P, node_perm, sphere_perm, out);
    else cache_model::run<double>(s, cam, p, P, node_perm, sphere_perm, out);
}
// The 4-wide records of the committed scene, for layout experiments: child[n][4], and the surface area of every child box area[n][4].
int hostsim_nodes(rttnw_scene* s, int32_t* child, float* area, int32_t* top_root) {
    if (!s || !s->committed) return RTTNW_ERR_INVALID;
    const auto& n4 = s->flat.nodes4;
    for (size_t i = 0; i < n4.size(); ++i)
        for (int c = 0; c < 4; ++c) {
            child[i * 4 + c] = n4[i].child[c];
            const float dx = n4[i].hi[0][c] - n4[i].lo[0][c], dy = n4[i].hi[1][c] - n4[i].lo[1][c], dz = n4[i].hi[2][c] - n4[i].lo[2][c];
            area[i * 4 + c] = n4[i].child[c] == CHILD_EMPTY ? 0.f : 2.f * (dx * dy + dy * dz + dz * dx);
        }
    *top_root = s->flat.top_root;
    return RTTNW_OK;
}
void hostsim_walk_histogram(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, int node_steps, uint64_t* hist,
                            uint64_t* out2) {
    walk_histogram(s, cam, p, node_steps, hist, out2);
}
void hostsim_wave_model(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, int policy, int a, int b, uint32_t n_waves,
                        uint32_t jobs_per_wave, uint64_t* out) {
    if (p->precision == RTTNW_F32) wave_model<float>(s, cam, p, policy, a, b, n_waves, jobs_per_wave, out);
    else wave_model<double>(s, cam, p, policy, a, b, n_waves, jobs_per_wave, out);
}
int hostsim_render(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, double* out_linear,
                   rttnw_stats* stats, int n_threads) {
    if (!s || !s->committed || !cam || !p || !out_linear) return RTTNW_ERR_INVALID;
    if (!s->flat.moving.empty() && (cam->open_time < s->flat.time0 || cam->close_time > s->flat.time1)) { // as render_api.cpp validate()
        std::string err;
        FlatScene wider;
        if (int rc = lower_scene(s->graph, wider, err, nullptr, std::min(s->flat.time0, cam->open_time), std::max(s->flat.time1, cam->close_time))) return rc;
        s->flat = std::move(wider);
    }
    // HOSTSIM_QUANT=1: walk the quantised records of the f64 decoupled kernel (rt_core.hpp trav_node_step4q) instead of the f32 ones
    const char* q = getenv("HOSTSIM_QUANT");
    if (q && q[0] == '1')
        return p->precision == RTTNW_F32 ? render_t<float, HostStack4Q>(s, cam, p, out_linear, stats, n_threads)
                                         : render_t<double, HostStack4Q>(s, cam, p, out_linear, stats, n_threads);
    if (q && q[0] == '3')
        return p->precision == RTTNW_F32 ? render_t<float, HostStackAll>(s, cam, p, out_linear, stats, n_threads)
                                         : render_t<double, HostStackAll>(s, cam, p, out_linear, stats, n_threads);
    return p->precision == RTTNW_F32 ? render_t<float, HostStack>(s, cam, p, out_linear, stats, n_threads)
                                     : render_t<double, HostStack>(s, cam, p, out_linear, stats, n_threads);
}
// The f64 kernels' 4-wide box test (rt_core.hpp slab_ray + slab_hit4, the form the host build is compiled with) on n cases: case i = four
// boxes lo[i][a][c] / hi[i][a][c] (floats), ray o[i], d[i] (doubles), range [tmin[i], tmax[i]]; pass[i][c] = 1 where the test lets the walk in.
int hostsim_slab4_f64(uint32_t n, const float* lo, const float* hi, const double* o, const double* d, const double* tmin, const double* tmax, uint8_t* pass) {
    for (uint32_t i = 0; i < n; ++i) {
        const V3<double> oo(o[3 * i], o[3 * i + 1], o[3 * i + 2]), dd(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        const SlabRay<double> sr = slab_ray<-1>(oo, dd);
        Planes4 nd;
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 4; ++c) {
                const float l = lo[(size_t(i) * 3 + a) * 4 + c], h = hi[(size_t(i) * 3 + a) * 4 + c];
                const bool near_is_lo = near_piece(a, sr) == uint32_t(a);
                nd.nr[a][c] = near_is_lo ? l : h;
                nd.fr[a][c] = near_is_lo ? h : l;
            }
        for (int c = 0; c < 4; ++c) nd.child[c] = 0;
        float lo_t, hi_t, e[4];
        bool h[4];
        slab_range(tmin[i], tmax[i], lo_t, hi_t);
        slab_hit4<-1>(nd, oo, sr, lo_t, hi_t, e, h);
        for (int c = 0; c < 4; ++c) pass[size_t(i) * 4 + c] = h[c] ? 1 : 0;
    }
    return 0;
}
// Chunk schedule and launch split of a render (rt_types.hpp plan_chunks / launch_chunks + rt_core.hpp plan_jobs) as a rank that
// owns rank_tiles 8x8 tiles sees it: out = {spp_chunk, n_main, n_chunks of the whole render, chunks per launch, launches, n_jobs
// of the first launch}; device_budget: the device's chunk-sum budget in bytes (render_common.hpp DeviceState::chunk_budget), 0 = the
// fallback constant; returns 0, or -1 when the job count of a launch does not fit.
int hostsim_plan(uint32_t spp, uint32_t user_chunk, uint32_t rank_tiles, uint32_t bytes_per_sum, uint64_t device_budget, uint32_t* out) {
    RenderConsts rc{};
    rc.my_tiles = rank_tiles;
    rc.spp = spp;
    plan_chunks(rc, spp, user_chunk);
    const uint32_t total = rc.n_chunks, per_launch = launch_chunks(uint64_t(rank_tiles) * 64, bytes_per_sum, total, device_budget);
    out[0] = rc.spp_chunk; out[1] = rc.n_main; out[2] = total; out[3] = per_launch; out[4] = (total + per_launch - 1) / per_launch;
    rc.n_chunks = std::min(per_launch, total);
    const bool ok = plan_jobs(rc);
    out[5] = rc.n_jobs;
    return ok ? 0 : -1;
}
// random_in_unit_space of the product core for n keys (f64 and f32 instantiations): out64[3 n], out32[3 n].
int hostsim_ball(uint32_t n, const uint64_t* keys, uint32_t bounce, double* out64, float* out32) {
    for (uint32_t i = 0; i < n; ++i) {
        const V3<double> a = random_in_unit_space<double>(keys[i], bounce);
        const V3<float> b = random_in_unit_space<float>(keys[i], bounce);
        out64[3 * i] = a.x; out64[3 * i + 1] = a.y; out64[3 * i + 2] = a.z;
        out32[3 * i] = b.x; out32[3 * i + 1] = b.y; out32[3 * i + 2] = b.z;
    }
    return 0;
}
int hostsim_box_fast_check(uint64_t n, uint64_t seed, int f32, int form, uint64_t* out) { // form: rt_core.hpp box_classify's two ways of writing the verdicts
    if (f32) { if (form) box_fast_check_t<float, 1>(n, seed, out); else box_fast_check_t<float, 0>(n, seed, out); }
    else { if (form) box_fast_check_t<double, 1>(n, seed, out); else box_fast_check_t<double, 0>(n, seed, out); }
    return 0;
}
// Entries the traversal stacks have needed since the last call (the device sizes its LDS stacks by FlatScene::stack_depth).
int hostsim_max_stack() { return g_max_stack.exchange(0); }
int hostsim_scene_dims(rttnw_scene* s, uint32_t* out /* nodes, spheres, moving, rects, boxes, insts, media, stack_depth */) {
    if (!s || !s->committed) return RTTNW_ERR_INVALID;
    out[0] = uint32_t(s->flat.nodes4.size()); out[1] = uint32_t(s->flat.spheres.size()); out[2] = uint32_t(s->flat.moving.size());
    out[3] = uint32_t(s->flat.rects.size()); out[4] = uint32_t(s->flat.boxes.size()); out[5] = uint32_t(s->flat.insts.size());
    out[6] = uint32_t(s->flat.media.size()); out[7] = s->flat.stack_depth;
    return RTTNW_OK;
}
// What the lowering decided about the committed scene: bit 0 LEAN (no moving sphere, no medium, solid colours only), bit 1 sphere material slot i holds i
// (FlatScene::sphere_mat_is_index: the kernels — and this build's SceneView — take the index itself, SceneView::sphere_mat == nullptr)
int hostsim_scene_flags(rttnw_scene* s) {
    if (!s || !s->committed) return RTTNW_ERR_INVALID;
    return (s->flat.lean() ? 1 : 0) | (s->flat.sphere_mat_is_index ? 2 : 0);
}
int hostsim_probe_path(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, uint32_t px, uint32_t row,
                       uint32_t sample, double* out, uint32_t max_out) {
    return p->precision == RTTNW_F32 ? probe_path_t<float>(s, cam, p, px, row, sample, out, max_out)
                                     : probe_path_t<double>(s, cam, p, px, row, sample, out, max_out);
}
}
