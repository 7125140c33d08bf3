// TEST INFRASTRUCTURE (included by hostsim.cpp) — an L2 model of the DECOUPLED trace kernel (trace_kernels.hpp trace_kernel) on a tree that
// lives in HBM: what config 5 (spheres_1m) waits on is L2-miss lines (profiles/r05: 96.9 per sample, 1.64x the algorithmic bytes), so layouts of
// the records are priced here — in lines missed per sample by stream — before anything is built for the device.
//
// One XCD: `n_waves` waves (32 CUs x 13 waves on the f64 LEAN flavour) share ONE set-associative LRU cache of `cache_bytes` (4 MB) with 128-byte
// lines.  A wave is the kernel's: 128 path slots, 64 lanes that hold a ray being walked, a ray queue and a hit queue; a turn of a wave is one SHADE
// pass (up to 64 queued hits) or one walk TRIP (RT_WAVE_STEPS node steps + a leaf step) of its burst; waves take turns round-robin, which is how the
// real ones interleave in the L2 (every wave progresses at about the same rate).  The walk is the product's (rt_core.hpp steps over the quantised
// records); every memory access of the real kernel is mapped to a line of one of the streams below and sent through the cache:
//   nodes (64-byte records, through `node_perm`), sphere records (32 B in f64, through `sphere_perm`), sphere_mat (4 B), materials (40 B in f64),
//   the path-state pool (hot: throughput, key, bounce — every shade; cold: pixel, sample range, job, running sum — when a path ends), job sums.
// Writes allocate without a fetch and count as a write-back when the dirty line leaves the cache.
// out: [0] samples measured, [1] rays, [2] node visits, [3] sphere tests, [4] sphere tests the pre-cull skipped;
//      [8 + 4 s ...]: accesses, read misses, write-backs, (spare) of stream s; [64 + d] / [96 + d]: node accesses / misses at tree depth d (< 32);
//      [128 ..]: wave-level executions of the node step and the lanes they served, likewise the leaf step and the shade pass.
#pragma once

namespace cache_model {
using namespace rt;

enum : uint32_t { S_NODE = 0, S_SPHERE, S_SPHMAT, S_MAT, S_POOL_HOT, S_POOL_COLD, S_PARTIAL, S_COUNT };

struct L2 {
    uint32_t sets = 0, ways = 0;
    std::vector<uint64_t> tag;   // line + 1 (0 = empty)
    std::vector<uint32_t> stamp;
    std::vector<uint8_t> dirty;
    uint32_t clock = 0;
    uint64_t acc[S_COUNT] = {}, miss[S_COUNT] = {}, wb[S_COUNT] = {};
    void init(uint64_t bytes, uint32_t w) {
        ways = w;
        sets = uint32_t(bytes / 128u / w);
        tag.assign(size_t(sets) * w, 0); stamp.assign(size_t(sets) * w, 0); dirty.assign(size_t(sets) * w, 0);
    }
    // returns true on a hit
    bool touch(uint32_t stream, uint64_t line_in_stream, bool write) {
        const uint64_t line = (uint64_t(stream) << 40) | line_in_stream;
        // (set index: the low line bits hashed with the stream and the upper bits, as address interleaving would spread separate allocations)
        uint64_t h = line * 0x9E3779B97F4A7C15ull;
        const uint32_t set = uint32_t((h >> 20) % sets);
        uint64_t* t = &tag[size_t(set) * ways];
        uint32_t* st = &stamp[size_t(set) * ways];
        uint8_t* d = &dirty[size_t(set) * ways];
        ++acc[stream];
        ++clock;
        for (uint32_t w = 0; w < ways; ++w)
            if (t[w] == line + 1) { st[w] = clock; d[w] |= uint8_t(write); return true; }
        uint32_t victim = 0, age = 0;
        for (uint32_t w = 0; w < ways; ++w) { // an empty way, else the least recently used
            if (t[w] == 0) { victim = w; break; }
            const uint32_t a = clock - st[w];
            if (a >= age) { age = a; victim = w; }
        }
        if (t[victim] != 0 && d[victim]) ++wb[uint32_t((t[victim] - 1) >> 40)];
        if (!write) ++miss[stream];
        t[victim] = line + 1; st[victim] = clock; d[victim] = uint8_t(write);
        return false;
    }
    void reset_counts() { for (uint32_t s = 0; s < S_COUNT; ++s) acc[s] = miss[s] = wb[s] = 0; }
};

struct Params {
    uint32_t n_waves, ways, node_steps, retire, warm_samples_per_slot, measure_samples_per_slot;
    uint64_t cache_bytes;
    uint32_t sphere_bytes;   // bytes of a sphere record (32 in f64)
    uint32_t mat_bytes;      // bytes of a material record (40 in f64)
    uint32_t mat_by_sphere;  // 1: the material lies at the sphere's (permuted) index — no sphere_mat read
    uint32_t real_bytes;     // 8 / 4: the pool's reals
    uint32_t precull;        // 1: a one-sphere leaf's record is not fetched when the ray misses the sphere inflated by precull_pct %
    uint32_t precull_pct;
    uint32_t xcds;           // 0: batches strided over the whole render; k > 0: this cache's waves take every k-th batch starting at batch_offset
    uint32_t batch_offset;
    uint32_t cold_words;     // words of a slot's bookkeeping read when its path ends: 5 (pixel, sample, sample end, 64-bit sum index); 2 = the job as its index (measured: slower)
    uint32_t leaf_threshold; // 0: a leaf step ends every trip (the kernel's); T: also inside the trip, after any node step that leaves >= T lanes at a leaf
    uint32_t unified;        // 1: node and sphere records share ONE buffer — node_perm / sphere_perm are positions in units of a sphere record (sphere_bytes) of it
    uint32_t node_bytes;     // 64 (quantised records) / 128
};

template <typename R>
static void run(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, const Params& P, const uint32_t* node_perm,
                const uint32_t* sphere_perm, uint64_t* out) {
    HostScene<R> hs(s->flat);
    hs.make_quant4(s->flat);
    CameraRec<double> cam64;
    make_camera(cam->lookfrom, cam->lookat, cam->view_up, cam->vertical_fov, cam->aspect_ratio, cam->aperture, cam->focus_distance,
                cam->open_time, cam->close_time, cam64);
    CameraRec<R> camr = narrow_camera<R>(cam64);
    RenderConsts rc{};
    rc.width = p->width; rc.height = p->height; rc.spp = p->spp; rc.max_depth = p->max_depth; rc.quirks = p->quirks; rc.seed = p->seed;
    rc.sample_begin = p->sample_begin;
    plan_chunks(rc, p->spp, p->spp_chunk);
    rc.tiles_x = (rc.width + 7) / 8; rc.tiles_y = (rc.height + 7) / 8; rc.n_tiles = rc.tiles_x * rc.tiles_y;
    rc.tile_rank = 0; rc.tile_world = 1; rc.my_tiles = rc.n_tiles;
    rc.div_tiles_x = make_fastdiv(std::max<uint32_t>(1u, rc.tiles_x));
    plan_jobs(rc);
    const V3<R> background(R(p->background[0]), R(p->background[1]), R(p->background[2]));
    const R t_min = R(p->t_min);
    NoCounters cnt;

    // depth of every node record (root 0)
    const size_t n_nodes = s->flat.nodes4.size();
    std::vector<uint8_t> depth(n_nodes, 31);
    {
        std::vector<int32_t> q{hs.view.top_root};
        depth[size_t(hs.view.top_root)] = 0;
        for (size_t i = 0; i < q.size(); ++i)
            for (int c = 0; c < 4; ++c) {
                const int32_t ch = s->flat.nodes4[size_t(q[i])].child[c];
                if (ch >= 0) { depth[size_t(ch)] = uint8_t(std::min<int>(31, depth[size_t(q[i])] + 1)); q.push_back(ch); }
            }
    }

    struct Slot {
        PathState<R> ps;
        uint32_t px = 0, row = 0, s = 0, s_end = 0;
        bool fresh = true, found = false;
        R closest = R(0);
        HitRef best{};
    };
    struct Lane { bool has_ray = false; uint32_t slot = 0; Trav<R> tr; HostStack4Q stack; };
    struct Wave {
        std::vector<Lane> lane;
        std::vector<Slot> slot;
        std::vector<uint8_t> rayq, hitq;
        uint64_t batch_next = 0, batch_end = 0;
        bool in_burst = false, done = false;
    };
    std::vector<Wave> W(P.n_waves);
    for (auto& w : W) {
        w.lane.resize(64); w.slot.resize(128);
        for (uint32_t i = 0; i < 128; ++i) w.hitq.push_back(uint8_t(i));
    }
    L2 l2;
    l2.init(P.cache_bytes, P.ways);
    uint64_t job_counter = 0; // in batches of 256 handed to THIS cache's waves
    const uint64_t n_batches = std::max<uint64_t>(1, rc.n_jobs / 256u);
    const uint64_t n_slots_total = uint64_t(P.n_waves) * 128u;
    uint64_t samples_done = 0, rays = 0, node_visits = 0, sphere_tests = 0, skipped = 0;
    uint64_t node_execs = 0, node_lanes = 0, leaf_execs = 0, leaf_lanes = 0, shade_execs = 0, shade_lanes = 0; // wave-level executions of a step and the lanes they served
    uint64_t dacc[32] = {}, dmiss[32] = {};
    bool measuring = false;
    const uint64_t warm = uint64_t(P.warm_samples_per_slot) * n_slots_total, total = warm + uint64_t(P.measure_samples_per_slot) * n_slots_total;

    auto pool = [&](uint32_t stream, uint32_t array, uint32_t elt, uint64_t g, bool write) {
        l2.touch(stream, (uint64_t(array) << 32) | ((g * elt) >> 7), write);
    };
    auto node_access = [&](int32_t idx) {
        const uint64_t at = node_perm ? node_perm[idx] : uint32_t(idx);
        const bool hit = l2.touch(S_NODE, P.unified ? at * P.sphere_bytes >> 7 : at * P.node_bytes >> 7, false);
        ++node_visits;
        const uint32_t d = depth[size_t(idx)];
        ++dacc[d];
        if (!hit) ++dmiss[d];
    };
    auto sphere_line = [&](uint32_t idx) { return (uint64_t(sphere_perm ? sphere_perm[idx] : idx) * P.sphere_bytes) >> 7; };
    const uint32_t S_SPH = P.unified ? uint32_t(S_NODE) : uint32_t(S_SPHERE);

    uint32_t live = P.n_waves;
    while (live != 0 && samples_done < total) {
        for (uint32_t wi = 0; wi < P.n_waves; ++wi) {
            Wave& w = W[wi];
            if (w.done) continue;
            const uint64_t gbase = uint64_t(wi) * 128u;
            if (!w.in_burst) {
                bool any_ray = false;
                for (auto& l : w.lane) any_ray |= l.has_ray;
                if (w.hitq.size() >= 64 || (!any_ray && w.rayq.empty())) {
                    if (w.hitq.empty()) { w.done = true; --live; continue; }
                    // ---------------- SHADE
                    const uint32_t m = uint32_t(std::min<size_t>(64, w.hitq.size()));
                    ++shade_execs; shade_lanes += m;
                    for (uint32_t k = 0; k < m; ++k) {
                        const uint32_t hs_ = w.hitq.back();
                        w.hitq.pop_back();
                        Slot& sl = w.slot[hs_];
                        const uint64_t g = gbase + hs_;
                        bool emit = false, need_sample = false;
                        if (!sl.fresh) {
                            for (uint32_t a = 0; a < 3; ++a) pool(S_POOL_HOT, a, P.real_bytes, g, false);
                            for (uint32_t a = 3; a < 6; ++a) pool(S_POOL_HOT, a, 4, g, false);
                            if (sl.found && ref_kind(sl.best.prim) == PRIM_SPHERE) {
                                const uint32_t idx = ref_index(sl.best.prim);
                                l2.touch(S_SPH, sphere_line(idx), false); // make_record reads the centre again
                                uint64_t mat_at;
                                if (P.mat_by_sphere) mat_at = uint64_t(sphere_perm ? sphere_perm[idx] : idx);
                                else {
                                    if (hs.view.sphere_mat) l2.touch(S_SPHMAT, uint64_t(idx) * 4u >> 7, false);
                                    mat_at = hs.view.sphere_mat ? uint64_t(uint32_t(hs.view.sphere_mat[idx]) & uint32_t(MAT_INDEX_MASK)) : uint64_t(idx);
                                }
                                const uint64_t b0 = mat_at * P.mat_bytes, b1 = b0 + P.mat_bytes - 1;
                                l2.touch(S_MAT, b0 >> 7, false);
                                if ((b1 >> 7) != (b0 >> 7)) l2.touch(S_MAT, b1 >> 7, false);
                            }
                            if (path_shade(sl.ps, hs.view, rc, background, t_min, sl.found, sl.closest, sl.best, cnt)) {
                                emit = true;
                            } else {
                                for (uint32_t a = 0; a < P.cold_words; ++a) pool(S_POOL_COLD, a, 4, g, false);       // pixel, sample, sample end, sum index
                                for (uint32_t a = 5; a < 8; ++a) pool(S_POOL_COLD, a, P.real_bytes, g, false);          // the job's running sum
                                ++sl.s;
                                ++samples_done;
                                need_sample = true;
                            }
                        } else {
                            need_sample = true;
                            sl.fresh = false;
                        }
                        bool slot_done = false;
                        while (need_sample && !slot_done && sl.s >= sl.s_end) {
                            if (sl.s_end != 0) l2.touch(S_PARTIAL, (uint64_t(sl.px) + uint64_t(sl.row) * rc.width) * 3u * P.real_bytes >> 7, true);
                            if (w.batch_next >= w.batch_end) {
                                // (batches spread over the whole render by a stride coprime to their number: a short model run must see the frame's mix of
                                // sky and geometry, not the first tiles of the first chunk group)
                                w.batch_next = (P.xcds ? (P.batch_offset + job_counter * P.xcds) % n_batches : (job_counter * 1000003ull) % n_batches) * 256u;
                                w.batch_end = w.batch_next + 256u;
                                ++job_counter;
                            }
                            const uint64_t job = w.batch_next++;
                            if (job >= rc.n_jobs) { slot_done = true; break; }
                            const JobInfo ji = job_decode(rc, uint32_t(job));
                            sl.px = ji.px; sl.row = ji.row; sl.s = ji.s; sl.s_end = ji.real ? ji.s_end : ji.s;
                        }
                        if (need_sample && !slot_done) {
                            path_begin(sl.ps, camr, rc, sl.px, sl.row, sl.s);
                            for (uint32_t a = 3; a < 5; ++a) pool(S_POOL_HOT, a, 4, g, true);            // key
                            for (uint32_t a = 0; a < P.cold_words; ++a) pool(S_POOL_COLD, a, 4, g, true);
                            for (uint32_t a = 5; a < 8; ++a) pool(S_POOL_COLD, a, P.real_bytes, g, true);
                            emit = true;
                        }
                        if (emit) {
                            for (uint32_t a = 0; a < 3; ++a) pool(S_POOL_HOT, a, P.real_bytes, g, true);
                            pool(S_POOL_HOT, 5, 4, g, true);
                            w.rayq.push_back(uint8_t(hs_));
                        }
                    }
                    if (!measuring && samples_done >= warm) {
                        measuring = true;
                        l2.reset_counts();
                        rays = node_visits = sphere_tests = skipped = 0;
                        node_execs = node_lanes = leaf_execs = leaf_lanes = shade_execs = shade_lanes = 0;
                        for (int d = 0; d < 32; ++d) dacc[d] = dmiss[d] = 0;
                        out[0] = samples_done;
                    }
                    continue;
                }
                // ---------------- hand queued rays to the idle lanes
                for (auto& l : w.lane) {
                    if (l.has_ray || w.rayq.empty()) continue;
                    l.slot = w.rayq.back();
                    w.rayq.pop_back();
                    trav_begin(l.tr, hs.view, w.slot[l.slot].ps.ray, l.stack);
                    l.has_ray = true;
                    ++rays;
                }
                w.in_burst = true;
            }
            // ---------------- one trip of the burst
            uint32_t walking = 0, finished = 0;
            auto leaf_step = [&]() {
                uint32_t served = 0;
                for (auto& l : w.lane) {
                    if (!l.has_ray || l.tr.node >= 0 || l.tr.node == TRAV_DONE) continue;
                    const Ray<R>& ray = w.slot[l.slot].ps.ray;
                    if (l.tr.node != CHILD_EMPTY && leaf_kind(l.tr.node) == PRIM_SPHERE) {
                        const uint32_t idx = leaf_first(l.tr.node) + l.tr.leaf_k;
                        bool fetch = true;
                        if (P.precull && leaf_count(l.tr.node) == 1) {
                            const SphereRec<R> sp = hs.view.spheres[idx];
                            R tt;
                            fetch = sphere_t(V3<R>(sp.cx, sp.cy, sp.cz), sp.r * R(1.0 + 0.01 * P.precull_pct), ray, t_min, l.tr.closest, tt);
                        }
                        ++sphere_tests;
                        if (fetch) l2.touch(S_SPH, sphere_line(idx), false); else ++skipped;
                    }
                    trav_leaf_step(l.tr, hs.view, ray, t_min, l.stack, cnt);
                    ++served;
                }
                if (served) { ++leaf_execs; leaf_lanes += served; }
            };
            for (uint32_t k = 0; k < P.node_steps; ++k) {
                uint32_t served = 0, at_leaf = 0;
                for (auto& l : w.lane)
                    if (l.has_ray && l.tr.node >= 0) {
                        node_access(l.tr.node);
                        trav_node_step(l.tr, hs.view, w.slot[l.slot].ps.ray, t_min, l.stack, cnt);
                        ++served;
                    }
                if (served) { ++node_execs; node_lanes += served; }
                // (a dynamic leaf step: taken inside the trip as soon as leaf_threshold lanes wait at a leaf — 0: only at the trip's end, the kernel's)
                if (P.leaf_threshold && k + 1 < P.node_steps) {
                    for (auto& l : w.lane) at_leaf += l.has_ray && l.tr.node < 0 && l.tr.node != TRAV_DONE;
                    if (at_leaf >= P.leaf_threshold) leaf_step();
                }
            }
            leaf_step();
            for (auto& l : w.lane) {
                if (!l.has_ray) continue;
                if (l.tr.node == TRAV_DONE) ++finished; else ++walking;
            }
            const uint32_t retire_batch = w.rayq.empty() ? 64u : P.retire;
            if (walking == 0 || finished >= retire_batch) {
                for (auto& l : w.lane)
                    if (l.has_ray && l.tr.node == TRAV_DONE) {
                        Slot& sl = w.slot[l.slot];
                        sl.found = l.tr.found; sl.closest = l.tr.closest; sl.best = l.tr.best;
                        if (!sl.found) sl.best.prim = make_ref(PRIM_NONE, 0);
                        w.hitq.push_back(uint8_t(l.slot));
                        l.has_ray = false;
                    }
                w.in_burst = false;
            }
        }
    }
    out[0] = samples_done - out[0];
    out[1] = rays; out[2] = node_visits; out[3] = sphere_tests; out[4] = skipped;
    out[128] = node_execs; out[129] = node_lanes; out[130] = leaf_execs; out[131] = leaf_lanes; out[132] = shade_execs; out[133] = shade_lanes;
    for (uint32_t st = 0; st < S_COUNT; ++st) { out[8 + 4 * st] = l2.acc[st]; out[9 + 4 * st] = l2.miss[st]; out[10 + 4 * st] = l2.wb[st]; }
    for (int d = 0; d < 32; ++d) { out[64 + d] = dacc[d]; out[96 + d] = dmiss[d]; }
}
} // namespace cache_model
