// rt_sched.hpp — stage selection for the TRAVERSE half of the trace kernel (render.hip).
//
// Every lane that holds a ray is waiting for exactly one kind of traversal step: visit an inner node, or test a
// record of one primitive kind.  Running "whatever each lane needs" every iteration executes the code of EVERY
// kind present in the wave (with 64 busy lanes that is nearly always all of them: ~590 VALU instructions per
// iteration on final_scene).  Instead the wave counts its lanes per stage (__ballot + popcount) and runs ONE
// stage per iteration.  Inner-node visits are ~85 % of the steps, so lanes are kept accumulated at nodes: a
// primitive stage is drained once `threshold[stage]` lanes wait for it, the node stage runs otherwise, and when
// no lane is at a node the fullest queue runs.
#pragma once
#include "rt_core.hpp"

namespace rt {

enum : uint32_t {
    ST_NODE = 0,   // an inner BVH node to visit
    ST_SPHERE = 1, // a sphere record to test
    ST_BOX = 2,    // a box (Cube) record to test
    ST_MISC = 3,   // rectangle / moving sphere / instance entry / empty slot
    ST_COUNT = 4,
    ST_NONE = 4    // lane without a ray
};

RT_HD uint32_t stage_of(int32_t node) {
    if (node >= 0) return ST_NODE;
    if (node == CHILD_EMPTY) return ST_MISC;
    const uint32_t k = leaf_kind(node);
    return k == PRIM_SPHERE ? ST_SPHERE : (k == PRIM_BOX ? ST_BOX : ST_MISC);
}

struct SchedPolicy {
    uint32_t threshold[ST_COUNT]; // lanes waiting before a primitive stage is drained ([ST_NODE] unused)
};
RT_HD SchedPolicy default_policy() {
    SchedPolicy p;
    p.threshold[ST_NODE] = 1;
    p.threshold[ST_SPHERE] = 16;
    p.threshold[ST_BOX] = 8;
    p.threshold[ST_MISC] = 8;
    return p;
}

// Rough VALU instructions of one execution of each stage (gfx950 ISA of trace_kernel<float>): weights of the
// threshold rule below, not a timing model.
RT_HD float stage_cost(uint32_t k) { return k == ST_NODE ? 100.f : k == ST_SPHERE ? 70.f : 100.f; }

// Thresholds from the lane-steps served per stage so far.  A queue k drained at T_k lanes holds T_k/2 idle lanes on
// average, which the node stage then lacks: occ_N = 64 - sum T_k/2.  Minimising the wave's instruction count
// sum_k steps_k*cost_k / occ_k over T gives  T_k = occ_N * sqrt(2 w_k / w_N),  w = steps * cost.
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

// Which stage runs this iteration, given the number of lanes waiting per stage (at least one must be non-zero).
RT_HD uint32_t sched_pick(const uint32_t n[ST_COUNT], const SchedPolicy& pol) {
    uint32_t pick = ST_NONE, best = 0;
#pragma unroll
    for (uint32_t k = 1; k < ST_COUNT; ++k) // 1. drain a full-enough queue (the fullest one wins)
        if (n[k] >= pol.threshold[k] && n[k] > best) { best = n[k]; pick = k; }
    if (pick != ST_NONE) return pick;
    if (n[ST_NODE] > 0) return ST_NODE;     // 2. otherwise keep descending
#pragma unroll
    for (uint32_t k = 1; k < ST_COUNT; ++k) // 3. nobody is at a node: run the fullest queue
        if (n[k] > best) { best = n[k]; pick = k; }
    return pick;
}

} // namespace rt
