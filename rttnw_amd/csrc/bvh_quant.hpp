// bvh_quant.hpp — f32 4-wide records -> the quantised records the f64 decoupled kernel walks (rt_types.hpp Bvh4QNode), one record
// per thread: host and device code (the HIP kernel around it lives in bvh_build.hip; tests/hostsim runs it on the host).  Record i
// keeps 4-wide record i's children in their slots, so indices, roots and the traversal-stack bound keep their meaning.  Boxes only
// cull: which records are walked never changes a result (hittable.rs:356-368 returns the closest hit whatever the boxes).
#pragma once
#include "rt_types.hpp"

#include <math.h>

namespace rt {

// 2^(e - 127) as a float, e in [1, 254]
RT_HD float quant_step(uint32_t e) {
    const uint32_t bits = e << 23;
    float f;
    __builtin_memcpy(&f, &bits, 4);
    return f;
}

// The 4-wide record i with its boxes quantised (rt_types.hpp Bvh4QNode): same children in the same slots.
RT_HD void quant4_make(const Bvh4Node* nodes4, int32_t i, Bvh4QNode& out) {
    const Bvh4Node& nd = nodes4[i];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    int n = 0;
    for (int c = 0; c < 4; ++c) {
        if (nd.child[c] == CHILD_EMPTY) continue;
        ++n;
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], nd.lo[a][c]); hi[a] = fmaxf(hi[a], nd.hi[a][c]); }
    }
    out.n_children = uint8_t(n);
    for (int k = 0; k < 8; ++k) out.pad[k] = 0;
    for (int a = 0; a < 3; ++a) {
        if (!(lo[a] <= hi[a])) { lo[a] = 0.f; hi[a] = 0.f; }
        out.org[a] = lo[a];
        const double extent = double(hi[a]) - double(lo[a]);
        int e2 = 0;
        (void)frexp(extent / 255.0, &e2); // extent / 255 = m 2^e2, m in [0.5, 1): the step 2^e2 >= extent / 255 (extent in double: exact for two floats)
        int e = e2 + 127;
        if (!(extent > 0.0) || e < 1) e = 1;
        if (e > 254) e = 254;
        while (e < 254 && ceil(extent / double(quant_step(uint32_t(e)))) > 255.0) ++e;
        out.ex[a] = uint8_t(e);
        const double step = double(quant_step(uint32_t(e)));
        for (int c = 0; c < 4; ++c) {
            if (nd.child[c] == CHILD_EMPTY) { out.q[a][0][c] = 255; out.q[a][1][c] = 0; continue; }
            double ql = floor((double(nd.lo[a][c]) - double(lo[a])) / step), qh = ceil((double(nd.hi[a][c]) - double(lo[a])) / step);
            if (!(ql >= 0.0)) ql = 0.0;
            if (ql > 255.0) ql = 255.0;
            if (!(qh <= 255.0)) qh = 255.0;
            if (qh < 0.0) qh = 0.0;
            out.q[a][0][c] = uint8_t(ql);
            out.q[a][1][c] = uint8_t(qh);
        }
    }
    for (int c = 0; c < 4; ++c) out.child[c] = nd.child[c];
}

} // namespace rt
