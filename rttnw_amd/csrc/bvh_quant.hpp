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

// ---- half-precision node-local records (rt_types.hpp Bvh4HNode) ----
// binary16 bits of the largest half <= x (x >= 0) / the smallest half >= x, by bit manipulation of the float (no hardware conversion:
// the rounding DIRECTION is the point).  Halves below the smallest normal (6.1e-5) are avoided: a positive hi rounds up to it at least.
RT_HD uint16_t half_down(float x) { // x >= 0 (or NaN -> 0)
    if (!(x > 0.f)) return 0;
    if (x >= 65504.f) return x < INFINITY ? 0x7BFFu : 0x7C00u; // the largest finite half (or +inf for +inf)
    if (x < 6.103515625e-5f) return 0;                           // below the smallest normal: down to zero
    uint32_t bits;
    __builtin_memcpy(&bits, &x, 4);
    const uint32_t e = (bits >> 23) - 112u, m = (bits >> 13) & 1023u; // truncation of the mantissa rounds a positive value down
    return uint16_t((e << 10) | m);
}
RT_HD uint16_t half_up(float x) { // x >= 0
    if (!(x > 0.f)) return 0;
    if (x > 65504.f) return 0x7C00u;                              // +inf
    if (x <= 6.103515625e-5f) return 0x0400u;                     // the smallest normal
    uint16_t h = half_down(x);
    if (half_bits_to_float(h) < x) ++h;                           // (mantissa overflow carries into the exponent: the next half; 0x7BFF + 1 = inf)
    return h;
}
RT_HD void half4_make(const Bvh4Node* nodes4, int32_t i, Bvh4HNode& out) {
    const Bvh4Node& nd = nodes4[i];
    float org[3] = {INFINITY, INFINITY, INFINITY};
    for (int c = 0; c < 4; ++c) {
        if (nd.child[c] == CHILD_EMPTY) continue;
        for (int a = 0; a < 3; ++a) org[a] = fminf(org[a], nd.lo[a][c]);
    }
    out.pad0 = 0;
    for (int k = 0; k < 12; ++k) out.pad[k] = 0;
    for (int a = 0; a < 3; ++a) {
        if (!(org[a] < INFINITY)) org[a] = 0.f; // no children
        out.org[a] = org[a];
        for (int c = 0; c < 4; ++c) {
            if (nd.child[c] == CHILD_EMPTY) { out.h[a][0][c] = 0x7C00u; out.h[a][1][c] = 0; continue; }
            // the offsets in double (exact for two floats), then to float outward, then to half outward
            const double dl = double(nd.lo[a][c]) - double(org[a]), dh = double(nd.hi[a][c]) - double(org[a]);
            float fl = float(dl), fh = float(dh);
            if (double(fl) > dl) fl = nextafterf(fl, -INFINITY);
            if (double(fh) < dh) fh = nextafterf(fh, INFINITY);
            out.h[a][0][c] = half_down(fl);
            out.h[a][1][c] = half_up(fh);
        }
    }
    for (int c = 0; c < 4; ++c) out.child[c] = nd.child[c];
}

} // namespace rt
