// trace_tally.hpp — DEBUG / MEASUREMENT ONLY: one round of the lane-owns-path kernel (trace_kernels.hpp trace_kernel_plain) with the
// wave clock read between its phases and the lockstep walk loop tallied.  Compiled into the COUNT = true instantiations only
// (rttnw_params.collect_counters != 0: the untimed counting pass of bench.py, RTTNW_DEBUG_SCHED); the product kernels never contain it.
//   prof[0..3]  wave clock: job hand-out / path begin / BVH walk / shade;  [15] media + hit record part of shade
//   prof[4]     lockstep iterations of the walk loop ([7] with a node lane, [8] with a leaf lane), [5] / [6] lane steps served (node / leaf)
//   prof[9]     bounce rounds, [10] lanes alive in them, [11] rounds that began paths, [12] paths begun
//   prof[13,14] clock of the node steps / the leaf steps
//   counters->dbg[16..], [80..], [144..] (collect_counters 2, 3): leaf clock by the set of record kinds served; histograms of trips per walk
#pragma once

namespace rt {
inline namespace RT_ARITH_NS {

template <typename R, typename Stack, typename Cnt>
__device__ __forceinline__ void plain_round_tallied(bool done, bool& alive, uint32_t px, uint32_t row, uint32_t& s, uint32_t s_end, V3<R>& acc, PathState<R>& ps,
                                                    const CameraRec<R>& cam, const RenderConsts& rc, const SceneView<R>& sc, V3<R> background, R t_min, Stack& stack,
                                                    Cnt& cnt, unsigned long long* prof, DeviceCounters* __restrict__ counters, uint32_t lane, long long tk0) {
    long long tk1 = 0, tk2 = 0, tk3 = 0;
    tk1 = clock64();
    const bool begin = !done && !alive && s < s_end;
    const unsigned long long bm = __ballot(begin);
    if (begin) {
        path_begin(ps, cam, rc, px, row, s);
        alive = true;
    }
    tk2 = clock64();
    const unsigned long long am = __ballot(alive);
    bool found = false;
    R closest = R(0);
    HitRef best;
    best.prim = 0; best.inst = -1; best.aux = 0;
    uint32_t my_trips = 0;
    if (alive) {
        cnt.ray();
        Trav<R> tr;
        trav_begin(tr, sc, ps.ray, stack);
        while (tr.node != TRAV_DONE) {
            ++my_trips;
            // the loop body of closest_solid() (two node steps, then a leaf step for the lanes at a leaf by then), tallied
            const unsigned long long act = __ballot(true);
            const bool is_node = tr.node >= 0;
            const unsigned long long nm = __ballot(is_node);
            const long long q0 = clock64();
            if (is_node) { prof[5] += 1; trav_node_step(tr, sc, ps.ray, t_min, stack, cnt); }
            if (tr.node >= 0) { prof[5] += 1; trav_node_step(tr, sc, ps.ray, t_min, stack, cnt); }
            const long long q1 = clock64();
            const bool is_leaf = tr.node < 0 && tr.node != TRAV_DONE;
            const unsigned long long lm = __ballot(is_leaf);
            uint32_t kmask = 0; // kinds among the leaf lanes: bit k = record kind k, bit 5 = empty slot
            {
                const uint32_t kd = tr.node == CHILD_EMPTY ? 5u : leaf_kind(tr.node);
    #pragma unroll
                for (uint32_t k = 0; k < 6; ++k) kmask |= __ballot(is_leaf && kd == k) != 0ull ? (1u << k) : 0u;
            }
            const long long q1b = clock64();
            if (is_leaf) { prof[6] += 1; trav_leaf_step(tr, sc, ps.ray, t_min, stack, cnt); }
            const long long q2 = clock64();
            if (lane == uint32_t(__ffsll((long long)act) - 1)) {
                prof[4] += 1;
                prof[7] += nm != 0ull;
                prof[8] += lm != 0ull;
                prof[13] += (unsigned long long)(q1 - q0);
                prof[14] += (unsigned long long)(q2 - q1b);
                if (kmask && rc.profile == 2u) { // leaf time by the set of record kinds the iteration served: dbg[16+set], count dbg[80+set]
                    atomicAdd(&counters->dbg[16 + kmask], (unsigned long long)(q2 - q1b));
                    atomicAdd(&counters->dbg[80 + kmask], 1ull);
                }
            }
        }
        found = tr.found; closest = tr.closest; best = tr.best;
        if (rc.profile == 3u) {
            atomicAdd(&counters->dbg[16 + min(my_trips, 63u)], 1ull); // histogram of trips per walk
            // ... and trips by what the walk found: 0 miss, 1 + kind (sphere, moving, rect, box), 6 anything inside an instance
            const uint32_t cls = !tr.found ? 0u : (tr.best.inst >= 0 ? 6u : 1u + ref_kind(tr.best.prim));
            atomicAdd(&counters->dbg[144 + cls], (unsigned long long)my_trips);
            atomicAdd(&counters->dbg[152 + cls], 1ull);
        }
    }
    if (rc.profile == 3u) { // ... and of the trips of the wave's longest walk, per round
        uint32_t mx = my_trips;
    #pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = max(mx, uint32_t(__shfl_xor(int(mx), off, 64)));
        if (lane == 0) atomicAdd(&counters->dbg[80 + min(mx, 63u)], 1ull);
    }
    tk3 = clock64();
    long long tk3b = tk3;
    if (alive) { // path_shade(), with the wave clock read between its two halves
        HitRecord<R> rec;
        const bool hit = world_hit_finish(sc, ps.ray, t_min, ps.key, ps.bounce, rc.quirks, found, closest, best, rec, cnt);
        tk3b = clock64();
        if (!hit) {
            ps.radiance = ps.throughput * background;
            if (closest != closest) ps.radiance = V3<R>(closest, closest, closest); // (as path_shade: a ray that was not walked)
            keep_rounded(ps.radiance);
            alive = false;
        } else {
            V3<R> att, emitted;
            const bool cont = shade(sc, rec, ps.key, ps.bounce, ps.ray, att, emitted, cnt);
            ps.radiance = ps.throughput * emitted;
            keep_rounded(ps.radiance); // (as path_shade: the path's value is a rounded product in every kernel form)
            if (cont) { ps.throughput = ps.throughput * att; ps.bounce += 1; }
            alive = cont && ps.bounce < rc.max_depth;
        }
        if (!alive) {
            acc = acc + ps.radiance;
            ++s;
        }
    }
    const long long tk4 = clock64();
    if (lane == 0) {
        prof[0] += (unsigned long long)(tk1 - tk0);
        prof[1] += (unsigned long long)(tk2 - tk1);
        prof[2] += (unsigned long long)(tk3 - tk2);
        prof[3] += (unsigned long long)(tk4 - tk3);
        prof[15] += (unsigned long long)(tk3b - tk3);
        prof[9] += 1;
        prof[10] += (unsigned long long)__popcll(am);
        prof[11] += bm != 0ull;
        prof[12] += (unsigned long long)__popcll(bm);
    }
}

// The same for the kernel's CURRENT form (RT_ASYNC_SHADE, round 5): one PHASE = start the walks of the lanes that have none, trips until at most
// RT_ASYNC_SLACK walks are unfinished, shade the finished ones.  Same prof[] slots; [9] counts phases, [10] the lanes a phase SHADED, [4..8] the trips
// and the lane steps they served, profile 3's histograms: trips per walk as before, and ([80..]) the trips of a PHASE.
template <int NSTEPS, typename R, typename Stack, typename Cnt>
__device__ __forceinline__ void plain_phase_tallied(bool done, bool& alive, bool& walking, Trav<R>& tr, uint32_t& my_trips, uint32_t px, uint32_t row, uint32_t& s, uint32_t s_end,
                                                    V3<R>& acc, PathState<R>& ps, const CameraRec<R>& cam, const RenderConsts& rc, const SceneView<R>& sc, V3<R> background,
                                                    R t_min, Stack& stack, Cnt& cnt, unsigned long long* prof, DeviceCounters* __restrict__ counters, uint32_t lane,
                                                    long long tk0) {
    const long long tk1 = clock64();
    const bool fresh_path = !done && !walking && !alive && s < s_end;
    const unsigned long long bm = __ballot(fresh_path);
    if (fresh_path) {
        path_begin(ps, cam, rc, px, row, s);
        alive = true;
    }
    const long long tk2 = clock64();
    const bool fresh_walk = !done && !walking && alive;
    if (fresh_walk) { cnt.ray(); trav_init(tr, sc); walking = true; my_trips = 0; }
    if constexpr (Cnt::NO_INST) trav_set_ray(tr, ps.ray, stack);
    else if (fresh_walk) trav_set_ray(tr, ps.ray, stack);
    if (fresh_walk) trav_reject_unwalkable(tr, ps.ray);
    uint32_t phase_trips = 0;
    for (;;) {
        const bool unfinished = walking && tr.node != TRAV_DONE;
        const unsigned long long act = __ballot(unfinished);
        if (act != 0ull) {
            ++phase_trips;
            unsigned long long n_node = 0, n_leaf = 0;
            bool any_node = false;
            const long long q0 = clock64();
            if (unfinished) {
                ++my_trips;
#pragma unroll
                for (int k = 0; k < NSTEPS; ++k)
                    if (tr.node >= 0) { ++n_node; trav_node_step(tr, sc, ps.ray, t_min, stack, cnt); }
            }
            any_node = __ballot(n_node != 0ull) != 0ull;
            const long long q1 = clock64();
            const bool is_leaf = unfinished && tr.node < 0 && tr.node != TRAV_DONE;
            const unsigned long long lm = __ballot(is_leaf);
            uint32_t kmask = 0;
            if (rc.profile == 2u) {
                const uint32_t kd = tr.node == CHILD_EMPTY ? 5u : leaf_kind(tr.node);
#pragma unroll
                for (uint32_t k = 0; k < 6; ++k) kmask |= __ballot(is_leaf && kd == k) != 0ull ? (1u << k) : 0u;
            }
            const long long q1b = clock64();
            if (is_leaf) { ++n_leaf; trav_leaf_step(tr, sc, ps.ray, t_min, stack, cnt); }
            const long long q2 = clock64();
            prof[5] += n_node;
            prof[6] += n_leaf;
            if (lane == uint32_t(__ffsll((long long)act) - 1)) {
                prof[4] += 1;
                prof[7] += any_node;
                prof[8] += lm != 0ull;
                prof[13] += (unsigned long long)(q1 - q0);
                prof[14] += (unsigned long long)(q2 - q1b);
                if (kmask && rc.profile == 2u) {
                    atomicAdd(&counters->dbg[16 + kmask], (unsigned long long)(q2 - q1b));
                    atomicAdd(&counters->dbg[80 + kmask], 1ull);
                }
            }
        }
        const unsigned long long um = __ballot(walking && tr.node != TRAV_DONE);
        if (um == 0ull) break;
        if (uint32_t(__popcll(um)) <= uint32_t(RT_ASYNC_SLACK) && __ballot(walking && tr.node == TRAV_DONE) != 0ull) break;
    }
    const long long tk3 = clock64();
    long long tk3b = tk3;
    const bool shade_now = walking && tr.node == TRAV_DONE;
    const unsigned long long sm = __ballot(shade_now);
    if (shade_now) {
        walking = false;
        if (rc.profile == 3u) {
            atomicAdd(&counters->dbg[16 + min(my_trips, 63u)], 1ull); // histogram of trips per walk
            const uint32_t cls = !tr.found ? 0u : (tr.best.inst >= 0 ? 6u : 1u + ref_kind(tr.best.prim));
            atomicAdd(&counters->dbg[144 + cls], (unsigned long long)my_trips);
            atomicAdd(&counters->dbg[152 + cls], 1ull);
        }
        HitRecord<R> rec;
        const bool hit = world_hit_finish(sc, ps.ray, t_min, ps.key, ps.bounce, rc.quirks, tr.found, tr.closest, tr.best, rec, cnt);
        tk3b = clock64();
        if (!hit) {
            ps.radiance = ps.throughput * background;
            if (tr.closest != tr.closest) ps.radiance = V3<R>(tr.closest, tr.closest, tr.closest); // (as path_shade: a ray that was not walked)
            keep_rounded(ps.radiance);
            alive = false;
        } else {
            V3<R> att, emitted;
            const bool cont = shade(sc, rec, ps.key, ps.bounce, ps.ray, att, emitted, cnt);
            ps.radiance = ps.throughput * emitted;
            keep_rounded(ps.radiance);
            if (cont) { ps.throughput = ps.throughput * att; ps.bounce += 1; }
            alive = cont && ps.bounce < rc.max_depth;
        }
        if (!alive) {
            acc = acc + ps.radiance;
            ++s;
        }
    }
    if (rc.profile == 3u && lane == 0) atomicAdd(&counters->dbg[80 + min(phase_trips, 63u)], 1ull); // trips of a phase
    const long long tk4 = clock64();
    if (lane == 0) {
        prof[0] += (unsigned long long)(tk1 - tk0);
        prof[1] += (unsigned long long)(tk2 - tk1);
        prof[2] += (unsigned long long)(tk3 - tk2);
        prof[3] += (unsigned long long)(tk4 - tk3);
        prof[15] += (unsigned long long)(tk3b - tk3);
        prof[9] += 1;
        prof[10] += (unsigned long long)__popcll(sm);
        prof[11] += bm != 0ull;
        prof[12] += (unsigned long long)__popcll(bm);
    }
}

} // namespace RT_ARITH_NS
} // namespace rt
