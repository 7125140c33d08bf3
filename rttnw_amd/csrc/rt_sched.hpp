// rt_sched.hpp — the lane state machine and the stage-selection policy of the trace kernel's
// intra-wave scheduler (render.hip), shared with the test-only host simulator (tests/hostsim) that
// replays a wave's 64 lanes to evaluate a policy offline.
//
// A lane owns one path and is, at any moment, waiting for exactly one kind of step.  Each iteration
// the wave counts its lanes per state (__ballot + popcount) and runs ONE stage for the lanes waiting
// on it.  Inner-node visits are ~80 % of all steps, so the policy keeps lanes accumulated in ST_NODE:
// every other stage is drained as soon as `threshold[stage]` lanes wait for it, ST_NODE runs
// otherwise, and when no lane is at a node the fullest queue runs.
#pragma once
#include "rt_core.hpp"

namespace rt {

enum : uint32_t {
    ST_NODE = 0,   // traversal: an inner BVH node to visit
    ST_SPHERE = 1, // traversal: a sphere record to test
    ST_BOX = 2,    // traversal: a box (Cube) record to test
    ST_MISC = 3,   // traversal: rectangle / moving sphere / instance entry / empty slot
    ST_POST = 4,   // traversal finished: media, hit record, emitted + scatter
    ST_NEW = 5,    // needs a camera ray (and possibly a new job)
    ST_COUNT = 6,
    ST_DONE = 6    // no jobs left
};

struct SchedPolicy {
    uint32_t threshold[ST_COUNT]; // lanes waiting before a non-node stage is drained ([ST_NODE] unused)
};
RT_HD SchedPolicy default_policy() {
    SchedPolicy p;
    p.threshold[ST_NODE] = 1;
    p.threshold[ST_SPHERE] = 16;
    p.threshold[ST_BOX] = 8;
    p.threshold[ST_MISC] = 8;
    p.threshold[ST_POST] = 24;
    p.threshold[ST_NEW] = 8;
    return p;
}

// Rough VALU instructions of one execution of each stage (from the gfx950 ISA of trace_kernel<float>):
// the weights of the threshold rule below, not a timing model.
RT_HD float stage_cost(uint32_t k) {
    return k == ST_NODE ? 100.f : k == ST_SPHERE ? 80.f : k == ST_BOX ? 90.f : k == ST_MISC ? 90.f : k == ST_POST ? 400.f : 300.f;
}

// Thresholds from the lane-steps served per stage so far.  A queue k that is drained at T_k lanes holds T_k/2
// idle lanes on average, which the node stage then lacks: occ_N = 64 - sum T_k/2.  Minimising the wave's
// instruction count  sum_k steps_k*cost_k / occ_k  over T gives  T_k = occ_N * sqrt(2 w_k / w_N),  w = steps*cost.
RT_HD void adapt_policy(SchedPolicy& pol, const uint32_t served[ST_COUNT]) {
    const float w_node = float(served[ST_NODE]) * stage_cost(ST_NODE);
    if (!(w_node > 0.f)) return;
    float r[ST_COUNT], sum = 0.f;
#pragma unroll
    for (uint32_t k = 1; k < ST_COUNT; ++k) {
        r[k] = sqrtf(2.f * float(served[k]) * stage_cost(k) / w_node);
        sum += r[k];
    }
    const float occ_node = 64.f / (1.f + 0.5f * sum);
#pragma unroll
    for (uint32_t k = 1; k < ST_COUNT; ++k) {
        float t = occ_node * r[k] + 0.5f;
        t = t < 2.f ? 2.f : (t > 48.f ? 48.f : t);
        pol.threshold[k] = uint32_t(t);
    }
}

// Which stage runs this iteration, given the number of lanes waiting per state.  ST_DONE: all done.
RT_HD uint32_t sched_pick(const uint32_t n[ST_COUNT], const SchedPolicy& pol) {
    uint32_t pick = ST_DONE, best = 0;
    // 1. drain a full-enough queue (the fullest one wins)
#pragma unroll
    for (uint32_t k = 1; k < ST_COUNT; ++k) {
        if (n[k] >= pol.threshold[k] && n[k] > best) { best = n[k]; pick = k; }
    }
    if (pick != ST_DONE) return pick;
    // 2. otherwise keep traversing
    if (n[ST_NODE] > 0) return ST_NODE;
    // 3. nobody is at a node: run the fullest queue
#pragma unroll
    for (uint32_t k = 1; k < ST_COUNT; ++k) {
        if (n[k] > best) { best = n[k]; pick = k; }
    }
    return pick;
}

RT_HD uint32_t stage_of(int32_t node) {
    if (node >= 0) return ST_NODE;
    if (node == TRAV_DONE) return ST_POST;
    if (node == CHILD_EMPTY) return ST_MISC;
    const uint32_t k = leaf_kind(node);
    return k == PRIM_SPHERE ? ST_SPHERE : (k == PRIM_BOX ? ST_BOX : ST_MISC);
}

// Everything a lane carries between steps.
template <typename R> struct Lane {
    uint32_t st;
    bool has_job;
    unsigned long long job;
    uint32_t px, row, s, s_end;
    V3<R> acc;       // sequential sum of the job's samples (main.rs:211-216)
    PathState<R> ps;
    Trav<R> tr;

    RT_HD void init() {
        st = ST_NEW; has_job = false; job = 0; px = row = s = s_end = 0; acc = V3<R>();
    }
    RT_HD bool needs_job() const { return st == ST_NEW && s >= s_end; }

    // Take job `j` (or go idle when the queue is exhausted).  job = (chunk, tile, pixel-in-tile).
    RT_HD void take_job(unsigned long long j, unsigned long long n_jobs, const RenderConsts& rc) {
        job = j;
        if (j >= n_jobs) { st = ST_DONE; return; }
        const unsigned long long jobs_per_chunk = (unsigned long long)rc.my_tiles * 64ull;
        const uint32_t chunk = uint32_t(j / jobs_per_chunk);
        const uint32_t rem = uint32_t(j % jobs_per_chunk);
        const uint32_t local_tile = rem >> 6, l = rem & 63u;
        uint32_t tx, ty;
        tile_unpermute(rc.tile_rank + local_tile * rc.tile_world, rc.tiles_x, tx, ty);
        px = tx * 8u + (l & 7u);
        row = ty * 8u + (l >> 3);
        s = chunk * rc.spp_chunk;
        s_end = s + rc.spp_chunk < rc.spp ? s + rc.spp_chunk : rc.spp;
        if (px >= rc.width || row >= rc.height) s = s_end; // outside the image: an empty job
        acc = V3<R>();
        has_job = true;
    }

    // ST_NEW with samples left: main.rs:212-215
    template <typename Cnt> RT_HD void step_new(const SceneView<R>& sc, const CameraRec<R>& cam, const RenderConsts& rc, Cnt& cnt) {
        path_begin(ps, cam, rc, px, row, s);
        cnt.ray();
        trav_begin(tr, sc, ps.ray);
        st = stage_of(tr.node);
    }
    template <typename Stack, typename Cnt> RT_HD void step_node(const SceneView<R>& sc, R t_min, Stack& stack, Cnt& cnt) {
        trav_node_step(tr, sc, ps.ray, t_min, stack, cnt);
        st = stage_of(tr.node);
    }
    template <typename Stack, typename Cnt> RT_HD void step_leaf(const SceneView<R>& sc, R t_min, Stack& stack, Cnt& cnt) {
        trav_leaf_step(tr, sc, ps.ray, t_min, stack, cnt);
        st = stage_of(tr.node);
    }
    template <typename Cnt> RT_HD void step_post(const SceneView<R>& sc, const RenderConsts& rc, V3<R> background, R t_min, Cnt& cnt) {
        const bool alive = path_shade(ps, sc, rc, background, t_min, tr.found, tr.closest, tr.best, cnt);
        if (alive) { // next world.hit of the same path
            cnt.ray();
            trav_begin(tr, sc, ps.ray);
            st = stage_of(tr.node);
        } else { // main.rs:216: acc + color(...)
            acc = acc + ps.radiance;
            ++s;
            st = ST_NEW;
        }
    }
};

} // namespace rt
