// rt_core.hpp — the per-lane tracing core: one path per lane, written as branch-flattened
// __host__ __device__ inline functions over the flat scene of rt_types.hpp.
//
// What the reference does by recursion over trait objects (color() main.rs:26-45 -> List::hit ->
// BvhTree::hit -> Sphere/Rectangle/Cube/Translate/YRotate/ConstantMedium::hit -> Material::scatter
// -> Texture::value), a lane does here as: stack-driven flat-BVH walk that only tracks
// (t, primitive, instance) -> one hit-record reconstruction for the winner -> media -> shade.
// Arithmetic follows the reference's formulas (cited per function) so that the f64 instantiation
// agrees with the oracle to rounding; R = float is the throughput instantiation.
#pragma once
#include "rt_types.hpp"
#include <math.h>
#include <stddef.h>

#ifndef RT_NODE_STEPS
#define RT_NODE_STEPS 2 // node steps per trip round the walk loop (closest_solid)
#endif

// Two builds of the arithmetic (the f64 translation units):
//   contracted   (default; render_f32.hip, render_f64.hip)  -ffp-contract=fast, shared-reciprocal f64 quotients (Recip / rt_div2 below):
//                agrees with the f64 reference arithmetic to rounding;
//   ieee_strict  (RT_STRICT_F64; render_f64_strict.hip, precision RTTNW_F64_STRICT)  -ffp-contract=off and every f64 quotient the
//                IEEE division the reference performs: the same operations in the same order as vec3.rs / hittable.rs / material.rs,
//                so a path takes the SAME decisions as the CPU reference bit for bit (what differs is the grouping of a pixel's sum and
//                the last place of transcendental functions).
// Everything that depends on the arithmetic lives in an inline namespace named after the build, so that the two f64 builds link into
// one library without their template instantiations colliding.
#if defined(RT_STRICT_F64)
#define RT_ARITH_NS ieee_strict
#ifndef RT_SHARED_RECIPROCALS // (overridden in experiments only: what the IEEE quotients cost)
#define RT_SHARED_RECIPROCALS 0
#endif
#else
#define RT_ARITH_NS contracted
#define RT_SHARED_RECIPROCALS 1
#endif

namespace rt {
#ifndef RT_NODE_EMPTY_CHECK
#define RT_NODE_EMPTY_CHECK 0 // 1: the node steps over f32 and quantised records test child != CHILD_EMPTY beside the (inverted) box of an unused slot.  Without:
                              // final_scene f64 1394 -> 1434 Msamples/s, f32 1857 -> 1889, cornell_box f64 1728 -> 1751 (four compares and the scalar ANDs between the
                              // hit tests and the selects); the kernels that read f32 records from memory +-0.3 %
#endif
#ifndef RT_PIN_CHILD_PIECE
#define RT_PIN_CHILD_PIECE 1 // the step over quantised records (read from memory) keeps the read of the children's piece beside the other pieces' reads
#endif
#ifndef RT_F64_SLAB_FOLDED
#define RT_F64_SLAB_FOLDED 2 // the f64 kernels' box test: 0 every box's entry / exit widened by 3.6e-7 |t| + slack (15 operations per box), 1 the widening in per-walk
                             // constants for both ends (11, five registers more: spills), 2 the slack in the constants and one multiplication per entry (12, two
                             // registers more), 3 one fma per end (13).  Measured (final_scene / cornell_box f64, Msamples/s): 1364 / 1692, 1370 / 1589, 1393 / 1727, 1.5-2 % behind 2
#endif
inline namespace RT_ARITH_NS {

// ---------------------------------------------------------------- math wrappers
RT_HD float rt_sqrt(float x) { return sqrtf(x); }
RT_HD double rt_sqrt(double x) { return sqrt(x); }
RT_HD float rt_sin(float x) { return sinf(x); }
RT_HD double rt_sin(double x) { return sin(x); }
RT_HD float rt_log(float x) { return logf(x); }
RT_HD double rt_log(double x) { return log(x); }
RT_HD float rt_acos(float x) { return acosf(x); }
RT_HD double rt_acos(double x) { return acos(x); }
RT_HD float rt_atan2(float y, float x) { return atan2f(y, x); }
RT_HD double rt_atan2(double y, double x) { return atan2(y, x); }
RT_HD float rt_floor(float x) { return floorf(x); }
RT_HD double rt_floor(double x) { return floor(x); }
RT_HD float rt_fabs(float x) { return fabsf(x); }
RT_HD double rt_fabs(double x) { return fabs(x); }
// IEEE minNum/maxNum: return the non-NaN operand, like Rust's f64::min/max (bound.rs:25-26)
RT_HD float rt_min(float a, float b) { return fminf(a, b); }
RT_HD double rt_min(double a, double b) { return fmin(a, b); }
RT_HD float rt_max(float a, float b) { return fmaxf(a, b); }
RT_HD double rt_max(double a, double b) { return fmax(a, b); }
// 1/x and a/b.  f64: IEEE division, as the reference.  f32 on the device: v_rcp_f32 (1 ulp) and one multiply — the
// compiler's own f32 `/` wraps the same v_rcp_f32 in 5 more instructions of range scaling for denormal divisors,
// which a ray tracer does not need (a denormal direction component is an axis-parallel ray either way).
RT_HD float rt_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
RT_HD double rt_rcp(double x) { return 1.0 / x; }
RT_HD float rt_div(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return a * __builtin_amdgcn_rcpf(b);
#else
    return a / b;
#endif
}
// (f64 single divisions stay IEEE: a v_rcp_f64 + Newton + Markstein form of ONE quotient was slower — final_scene f64 990
// against 1000 Msamples/s, spheres_1m f64 239 against 254; the forms below pay where a divisor is shared)
RT_HD double rt_div(double a, double b) { return a / b; }
// Two quotients with ONE divisor: n1 / d and n2 / d.  f32 and the host: two divisions.  f64 on the device: the compiler's
// IEEE division is an 11-instruction sequence around a quarter-rate v_rcp_f64, per quotient; here the reciprocal (v_rcp_f64
// + two Newton steps) is shared and each quotient is q = n r corrected once by its residual (q + (n - d q) r, Markstein's
// form: the correctly rounded quotient unless r is 1 ulp off the rounded reciprocal AND the quotient sits within that of a
// rounding boundary; d = 0 gives NaN where IEEE gives inf — no hit either way in the one caller, box_t).
RT_HD void rt_div2(float n1, float n2, float d, float& q1, float& q2) { q1 = rt_div(n1, d); q2 = rt_div(n2, d); }
RT_HD void rt_div2(double n1, double n2, double d, double& q1, double& q2) {
#if defined(__HIP_DEVICE_COMPILE__) && RT_SHARED_RECIPROCALS
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    const double a1 = n1 * r, a2 = n2 * r;
    q1 = __builtin_fma(__builtin_fma(-d, a1, n1), r, a1);
    q2 = __builtin_fma(__builtin_fma(-d, a2, n2), r, a2);
#else
    q1 = n1 / d; q2 = n2 / d;
#endif
}
// A divisor used more than once: its reciprocal kept beside it.  f32: v_rcp_f32, a quotient is one multiply (what rt_div
// does).  f64 on the device: the refined reciprocal of rt_div2 and Markstein's corrected quotient, 3 FMA-rate
// instructions per quotient instead of the 11-instruction IEEE sequence with its quarter-rate v_rcp_f64; the host build
// divides.  A zero divisor gives NaN quotients where IEEE gives infinities: callers reject with tests NaN fails.
template <typename R> struct Recip;
template <> struct Recip<float> { float r; };
template <> struct Recip<double> { double d, r; };
RT_HD Recip<float> recip_of(float d) { return {rt_rcp(d)}; }
RT_HD Recip<double> recip_of(double d) {
#if defined(__HIP_DEVICE_COMPILE__) && RT_SHARED_RECIPROCALS
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return {d, r};
#else
    return {d, 0.0};
#endif
}
RT_HD float div_by(float n, const Recip<float>& k) { return n * k.r; }
RT_HD double div_by(double n, const Recip<double>& k) {
#if defined(__HIP_DEVICE_COMPILE__) && RT_SHARED_RECIPROCALS
    const double q = n * k.r;
    return __builtin_fma(__builtin_fma(-k.d, q, n), k.r, q);
#else
    return n / k.d;
#endif
}

template <typename R> struct Lim;
template <> struct Lim<float> {
    static RT_HD float inf() { return __builtin_huge_valf(); }
    static RT_HD float max() { return 3.402823466e+38f; }
};
template <> struct Lim<double> {
    static RT_HD double inf() { return __builtin_huge_val(); }
    static RT_HD double max() { return 1.7976931348623157e+308; }
};

// ---------------------------------------------------------------- Vec3f (vec3.rs:27-262)
template <typename R> struct V3 {
    R x, y, z;
    RT_HD V3() : x(0), y(0), z(0) {}
    RT_HD V3(R a, R b, R c) : x(a), y(b), z(c) {}
    RT_HD explicit V3(const R* p) : x(p[0]), y(p[1]), z(p[2]) {}
    RT_HD R get(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
template <typename R> RT_HD V3<R> operator+(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename R> RT_HD V3<R> operator-(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename R> RT_HD V3<R> operator*(V3<R> a, V3<R> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <typename R> RT_HD V3<R> operator*(V3<R> a, R k) { return {a.x * k, a.y * k, a.z * k}; }
template <typename R> RT_HD V3<R> operator*(R k, V3<R> a) { return {a.x * k, a.y * k, a.z * k}; }
template <typename R> RT_HD V3<R> operator/(V3<R> a, R k) {
    const Recip<R> rk = recip_of(k); // (f32: three multiplies by v_rcp_f32(k), as before)
    return {div_by(a.x, rk), div_by(a.y, rk), div_by(a.z, rk)};
}
template <typename R> RT_HD V3<R> operator-(V3<R> a) { return {-a.x, -a.y, -a.z}; }
template <typename R> RT_HD R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // vec3.rs:77
template <typename R> RT_HD R squared_length(V3<R> a) { return a.x * a.x + a.y * a.y + a.z * a.z; } // vec3.rs:93
template <typename R> RT_HD R magnitude(V3<R> a) { return rt_sqrt(a.x * a.x + a.y * a.y + a.z * a.z); } // vec3.rs:90
template <typename R> RT_HD V3<R> unit(V3<R> a) {
    R k;
    if constexpr (sizeof(R) == 4) k = R(1) / magnitude(a); else k = rt_rcp(magnitude(a));
    return a * k;
}          // vec3.rs:97
template <typename R> RT_HD V3<R> reflect(V3<R> v, V3<R> n) { return v - R(2) * dot(v, n) * n; }      // vec3.rs:112
template <typename R> RT_HD V3<R> refract(V3<R> v, V3<R> n, R eta) {                                  // vec3.rs:116-121
    R cos_theta = rt_min(dot(-v, n), R(1));
    V3<R> perp = eta * (v + cos_theta * n);
    V3<R> par = -rt_sqrt(rt_fabs(R(1) - squared_length(perp))) * n;
    return perp + par;
}

template <typename R> struct Ray { // ray.rs:9-27
    V3<R> o, d;
    R time;
    RT_HD V3<R> at(R t) const { return o + t * d; }
};

// ---------------------------------------------------------------- keyed RNG (DESIGN.md "RNG")
// U = mixd(key + (ctr+1)*GAMMA) with key = f(seed, pixel, sample): every draw is addressed by
// (pixel, sample, bounce, slot), so results do not depend on traversal order, lane assignment,
// work stealing or GPU count.  f64 uniform = top 53 bits, f32 uniform = top 24 bits of the SAME word.
constexpr uint64_t RNG_GAMMA = 0x9E3779B97F4A7C15ull;
RT_HD uint64_t mix64(uint64_t z) { // SplitMix64's finaliser: the sample keys (two per sample) and the scene streams
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// The mixer of a DRAW (round 4; ~40 of them per sample): three folds of the high word into the low one around two multiplications by
// 32-BIT constants.  On gfx950 a 64 x 64-bit product is three quarter-rate instructions and a shift by 27-31 three more operations;
// a product with a 32-bit constant is two, a fold by exactly 32 one xor: 4 quarter-rate + 5 full-rate instructions a draw against
// mix64's 6 + 13 (a quarter of the f64 kernels' vector instructions were integer, the keyed hash the largest part).  The key is a
// mix64 output and the counter enters through the Weyl step, so the weaker avalanche of a two-round 32-bit mixer is spent on
// inputs that are already mixed; tests/test_rng_quality.py holds the draws of a path to uniformity, pairwise independence over the
// counters a path uses, serial independence over adjacent samples and the unit-ball acceptance rate.
RT_HD uint64_t mixd(uint64_t z) {
    z ^= z >> 32; z *= 0x9E3779B1ull;
    z ^= z >> 32; z *= 0x85EBCA6Bull;
    return z ^ (z >> 32);
}
RT_HD uint64_t sample_key(uint64_t seed, uint64_t pixel, uint64_t sample) {
    uint64_t k0 = mix64(seed + RNG_GAMMA);
    uint64_t k1 = mix64(k0 + pixel * 0xD1B54A32D192ED03ull);
    return mix64(k1 + sample * 0x8CB92BA72F3D8DD7ull);
}
RT_HD uint64_t rng_word(uint64_t key, uint32_t ctr) { return mixd(key + (uint64_t(ctr) + 1) * RNG_GAMMA); }
template <typename R> RT_HD R uniform01(uint64_t key, uint32_t ctr);
template <> RT_HD double uniform01<double>(uint64_t key, uint32_t ctr) {
    return double(rng_word(key, ctr) >> 11) * (1.0 / 9007199254740992.0);
}
template <> RT_HD float uniform01<float>(uint64_t key, uint32_t ctr) {
    const uint32_t top = uint32_t(rng_word(key, ctr) >> 40);
#if defined(__HIP_DEVICE_COMPILE__)
    // hipcc widens `float(uint32_t(w >> 40))` to a u64 -> f32 conversion (6 more instructions per draw): pin v_cvt_f32_u32
    float f;
    asm("v_cvt_f32_u32 %0, %1" : "=v"(f) : "v"(top));
    return f * (1.0f / 16777216.0f);
#else
    return float(top) * (1.0f / 16777216.0f);
#endif
}
// Uniform for a draw that goes through log() (the media's free-flight distance, hittable.rs:765): f32 keeps 24
// SIGNIFICANT bits of the same word instead of its top 24 bits, so that -ln(U)/density keeps f32's relative
// precision for small U (top-24-bit U near 0.01 would move a fog event by ~0.06 scene units).  May round to 1.0
// (ln = 0).  f64 is unchanged.
template <typename R> RT_HD R uniform01_log(uint64_t key, uint32_t ctr);
template <> RT_HD double uniform01_log<double>(uint64_t key, uint32_t ctr) { return uniform01<double>(key, ctr); }
template <> RT_HD float uniform01_log<float>(uint64_t key, uint32_t ctr) {
    return float(rng_word(key, ctr) >> 11) * (1.0f / 9007199254740992.0f);
}
// counter = block * 1024 + slot; block 0 = camera, block b+1 = b-th world.hit() of the path
enum : uint32_t { SLOT_JITTER_U = 0, SLOT_JITTER_V = 1, SLOT_TIME = 2, SLOT_LENS = 8,
                  SLOT_MEDIUM = 0, SLOT_DIELECTRIC = 16, SLOT_SCATTER = 32 };
RT_HD uint32_t rng_ctr(uint32_t block, uint32_t slot) { return block * 1024u + slot; }

// Vec3f::random_in_unit_space — vec3.rs:149-160: rejection sampling of the unit BALL:
//     loop { v = 2 (U, U, U) - 1; if |v|^2 < 1 return v }
// A wave runs this loop to the iteration count of its unluckiest lane — about six trips for a mean of 1.9 — and what a trip
// costs is its 64-bit hashes (quarter-rate integer multiplies): with one hash per uniform the loop was
// 12 % of the f64 kernel.  So the three uniforms of a candidate are drawn HIERARCHICALLY: one word H
// gives the 21 leading bits of each, TWO more words (round 4: their 96 leading bits; round 3 spent three words) their remaining 32,
//     u_c = ( field_c(H) * 2^32 + low_c ) * 2^-53,   field_0 = H >> 43, field_1 = (H >> 22) & 0x1FFFFF, field_2 = (H >> 1) & 0x1FFFFF,
//     low_0 = L >> 32, low_1 = L & 0xFFFFFFFF, low_2 = M >> 32,
// slots 32 + 4 i (H), 32 + 4 i + 1 (L), 32 + 4 i + 2 (M) of iteration i: still 53 independent uniform bits per coordinate.  H alone
// pins the candidate to a cube of side 2^-20 around (x_c): |v|^2 lies within 1.7e-6 of |x_c|^2, so unless |x_c|^2 is within
// 4e-6 of 1 the reference's test `|v|^2 < 1` is decided without the other three words — a rejected trip costs one hash
// instead of three — and the accepted candidate's two low words are fetched once, after the loop.  The sliver in between
// (4e-6 of the candidates) evaluates the full candidate.  Same algorithm, same predicate on the same candidate: the oracle
// simply evaluates every candidate in full.
RT_HD uint32_t ball_field(uint64_t h, int c) { return c == 0 ? uint32_t(h >> 43) : (c == 1 ? uint32_t(h >> 22) & 0x1FFFFFu : uint32_t(h >> 1) & 0x1FFFFFu); }
template <typename R> RT_HD R ball_uniform(uint32_t field, uint32_t low);
template <> RT_HD double ball_uniform<double>(uint32_t field, uint32_t low) {
    return double((uint64_t(field) << 32) | uint64_t(low)) * (1.0 / 9007199254740992.0);
}
template <> RT_HD float ball_uniform<float>(uint32_t field, uint32_t low) { // the top 24 of the same 53 bits
    const uint32_t top = (field << 3) | (low >> 29);
#if defined(__HIP_DEVICE_COMPILE__)
    float f;
    asm("v_cvt_f32_u32 %0, %1" : "=v"(f) : "v"(top));
    return f * (1.0f / 16777216.0f);
#else
    return float(top) * (1.0f / 16777216.0f);
#endif
}
template <typename R> RT_HD V3<R> ball_candidate(uint64_t key, uint32_t c, uint64_t h) {
    const uint64_t l = rng_word(key, c + 1), m = rng_word(key, c + 2);
    const V3<R> r(ball_uniform<R>(ball_field(h, 0), uint32_t(l >> 32)), ball_uniform<R>(ball_field(h, 1), uint32_t(l)),
                  ball_uniform<R>(ball_field(h, 2), uint32_t(m >> 32)));
    return R(2) * r - V3<R>(R(1), R(1), R(1));
}
template <typename R> RT_HD V3<R> random_in_unit_space(uint64_t key, uint32_t bounce) {
    uint32_t c = rng_ctr(bounce + 1, SLOT_SCATTER);
    uint64_t h;
    for (;;) {
        h = rng_word(key, c);
        // the cube's centre: x_c = 2 (field + 1/2) 2^-21 - 1
        const float x = float(ball_field(h, 0)) * 9.5367431640625e-7f + (4.76837158203125e-7f - 1.f),
                    y = float(ball_field(h, 1)) * 9.5367431640625e-7f + (4.76837158203125e-7f - 1.f),
                    z = float(ball_field(h, 2)) * 9.5367431640625e-7f + (4.76837158203125e-7f - 1.f);
        const float s = x * x + y * y + z * z;
        bool inside = s < 1.f - 4e-6f;
        if (!inside && !(s > 1.f + 4e-6f)) inside = squared_length(ball_candidate<R>(key, c, h)) < R(1);
        if (inside) break;
        c += 4;
    }
    return ball_candidate<R>(key, c, h);
}

// ---------------------------------------------------------------- counters (instrumentation)
// Counter policies.  Their template flag GENERAL also selects, at compile time, whether the code for the rare graph shapes
// (more than FAST_INSTANCE_OPS wrappers around an object, List / BvhTree medium boundaries, media inside transformed groups;
// FlatScene::needs_general) is compiled into a kernel at all: it costs the common kernels registers even when it never runs.
// Since round 4 the flag has a third value: SHAPES_NONE = a scene whose WALK NEVER CHANGES FRAMES: no tree holds an instance with a tree of its own
// (spheres_1m, random_scene: no wrapper at all; final_scene: its cluster's spheres are world-space copies).  The walk's per-lane copy of the ray in the current frame, the sentinel test of
// every pop and the descent into an instance's tree are compiled out — 14 registers in the f64 decoupled kernel, which is held to
// 168 for its third wave: scratch 120 -> 52 B per lane, spheres_1m f64 310 -> 325 Msamples/s, RTTNW_F64_STRICT 300 -> 320.
// SHAPES_SINGLE is the same where instance leaves do exist but all are single wrapped records (cornell_box: +2.9 % in f64, +4 % in the strict build): it
// keeps the branch that tests such a record in place, which SHAPES_NONE scenes are better off without (final_scene f32: 1 %).
// Round 5: SHAPES_NONE_NT / SHAPES_SINGLE_NT are the same two for LEAN scenes — no MovingSphere (nothing reads Ray::time), no ConstantMedium, every
// material a solid colour (cornell_box, spheres_1m, random_scene without its moving spheres ...): the moving-sphere code, the media loop with its log, the
// texture chain with its three sines / Perlin turbulence / image lookup and the spheres' (u, v) with their acos / atan2 are COMPILED OUT, and with them
// the registers they hold in the f64 kernels, which run at their register cap (every spilled register is a scratch access per shade phase: cornell_box f64
// +3 % for two registers, +6 % for five, profiles/r05/README.md) — and the ray time's one of the thirteen reals of the decoupled kernel's path slots.
enum : int { SHAPES_FAST = 0, SHAPES_GENERAL = 1, SHAPES_NONE = 2, SHAPES_SINGLE = 3, SHAPES_NONE_NT = 4, SHAPES_SINGLE_NT = 5 };
template <int G> struct NoCountersT {
    static constexpr bool GENERAL = G == SHAPES_GENERAL;
    static constexpr bool NO_INST = G == SHAPES_NONE || G == SHAPES_SINGLE || G == SHAPES_NONE_NT || G == SHAPES_SINGLE_NT; // the walk never changes frames
    static constexpr bool NO_INST_LEAF = G == SHAPES_NONE || G == SHAPES_NONE_NT;                                            // ... and meets no instance leaf at all
    static constexpr bool NO_TIME = G == SHAPES_NONE_NT || G == SHAPES_SINGLE_NT;                                            // nothing reads Ray::time ...
    static constexpr bool LEAN = NO_TIME;                                                                                    // ... and the scene has no medium and no texture but solid colours
    RT_HD void ray() {}
    RT_HD void node() {}
    RT_HD void prim() {}
    RT_HD void texel() {}
};
template <int G> struct LaneCountersT {
    static constexpr bool GENERAL = G == SHAPES_GENERAL;
    static constexpr bool NO_INST = G == SHAPES_NONE || G == SHAPES_SINGLE || G == SHAPES_NONE_NT || G == SHAPES_SINGLE_NT;
    static constexpr bool NO_INST_LEAF = G == SHAPES_NONE || G == SHAPES_NONE_NT;
    static constexpr bool NO_TIME = G == SHAPES_NONE_NT || G == SHAPES_SINGLE_NT;
    static constexpr bool LEAN = NO_TIME;
    uint32_t rays = 0, nodes = 0, prims = 0, texels = 0;
    RT_HD void ray() { ++rays; }
    RT_HD void node() { ++nodes; }
    RT_HD void prim() { ++prims; }
    RT_HD void texel() { ++texels; }
};
using NoCounters = NoCountersT<SHAPES_GENERAL>;     // host build, probes: every shape
using LaneCounters = LaneCountersT<SHAPES_GENERAL>;

// ---------------------------------------------------------------- camera (camera.rs:63-84)
template <typename R>
RT_HD Ray<R> camera_ray(const CameraRec<R>& cam, R s, R t, uint64_t key, bool time_is_read = true) {
    // random_in_unit_disk runs even when lens_radius == 0 (Q15); with a zero radius its value
    // cannot matter, so the loop is skipped then — the draws are keyed, nothing shifts.
    R px = 0, py = 0;
    if (cam.lens_radius != R(0)) {
        uint32_t c = rng_ctr(0, SLOT_LENS);
        for (;;) {
            px = R(2) * uniform01<R>(key, c) - R(1);
            py = R(2) * uniform01<R>(key, c + 1) - R(1);
            if (px * px + py * py + R(0) < R(1)) break;
            c += 2;
        }
    }
    R rdx = cam.lens_radius * px, rdy = cam.lens_radius * py;
    V3<R> offset = V3<R>(cam.u) * rdx + V3<R>(cam.v) * rdy;
    Ray<R> r;
    r.o = V3<R>(cam.origin) + offset;
    r.d = V3<R>(cam.lower_left_corner) + s * V3<R>(cam.horizontal) + t * V3<R>(cam.vertical) - V3<R>(cam.origin) - offset;
    // camera.rs:82.  The draw is keyed: a scene in which nothing reads Ray::time (no MovingSphere) may leave it out
    r.time = time_is_read ? cam.open_time + (cam.close_time - cam.open_time) * uniform01<R>(key, rng_ctr(0, SLOT_TIME)) : cam.open_time;
    return r;
}

// ---------------------------------------------------------------- AABB slab test (bound.rs:13-32)
// The three per-axis early-outs of the reference collapse into one comparison: tmin only grows and
// tmax only shrinks, so `tmax < tmin` at the end <=> it held at some axis.
// Forms of the f32 kernels' slab test (the f64 kernels always use the conservative 7-value form below): the Stack type of a
// kernel names its form (Stack::SLAB_F32).  The lane-owns-path kernel with its nodes in LDS is bound by vector-instruction
// issue and takes the FOLDED conservative fma form, t = plane * inv - o * inv with the error bound folded into the per-walk
// constants: 11 operations per box instead of 18 (final_scene f32 1645 -> 1755 Msamples/s, cornell_box 2068 -> 2103).  The
// kernels that read nodes from global memory (big trees) keep the exact two-operation form (plane - o) * inv: the fma form's
// error is ABSOLUTE, ~eps |o inv|, which for a far-away origin and a small direction component — spheres_1m's primary rays —
// is larger than a leaf, so its widening opens boxes the exact form culls (spheres_1m f32 396 -> 346 with the folded form).
enum : int { SLAB_EXACT = 0, SLAB_FMA_FOLDED = 2 };
template <typename R> struct SlabRay { // what a ray contributes to every slab test of its walk
    V3<R> inv; // 1 / d                                                               (SLAB_EXACT)
    // near planes: t = plane * inv_n - oinv_n comes out ALREADY moved down by the error bound, far planes up   (SLAB_FMA_FOLDED)
    float inv_n[3], oinv_n[3], inv_f[3], oinv_f[3];
};
template <typename Stack, typename R> constexpr int slab_form() { return sizeof(R) == 8 ? -1 : Stack::SLAB_F32; }
// f64: the boxes are f32 and only cull, so the f64 kernels test them in f32 too — CONSERVATIVELY, which the f32
// kernels need not be: origin and 1/d are rounded to f32 once per walk (1/d as v_rcp_f32 of the rounded d: 1 ulp — three
// f64 divisions per walk start, instance entry and instance exit were 5 % of the f64 kernel's instructions), a plane
// distance is ONE fused multiply-add, t32 = fl(b inv32 - oi32) with oi32 = fl(o32 inv32) kept per walk (round 3: 6 instead
// of 12 operations per box, a sixth of a node step), and every plane distance is widened by a bound on what the roundings
// can have done to it.  With o32 = o(1+e0), |e0| <= 2^-24, inv32 = inv(1+e1), |e1| <= 2^-24 + 2^-23, oi32 = o32 inv32 (1+e2),
// |e2| <= 2^-24, and the fma's single rounding e3:
//   t32 = [ (b - o) inv (1+e1) - o inv (1+e1)((1+e0)(1+e2) - 1) ] (1+e3)
//   |t32 - t| <= 2.4e-7 |t| + 1.2e-7 |o inv|          (t = (b - o) inv exactly, b a float)
// so near planes move down and far planes up by 3.6e-7 |t32| + slack, slack = 2.4e-7 |oi32| (twice the bound, as before;
// the two-operation form fl(fl(b - o32) inv32) it replaces had 6e-8 |o inv| there and used 1.2e-7), the range's ends are
// rounded outward, and NaN / inf (axis-parallel rays) never cull.  A box the exact test would pass always passes: images
// and hits are those of f64 slab tests, a node step costs about a third (f64 runs at half rate and selects move register
// pairs).
template <> struct SlabRay<double> {
    float oinv[3], inv[3]; // o * (1 / d) and 1 / d: a plane distance is ONE fma, plane * inv - oinv (below)
    float slack; // the largest of the three axes' slacks: one widening of the box's entry / exit serves all planes (below)
#if RT_F64_SLAB_FOLDED
    float inv_n[3], oinv_n[3], inv_f[3], oinv_f[3]; // the same with the widening folded into the constants (slab_hit4 below)
#endif
};
template <int FORM, typename R> RT_HD SlabRay<R> slab_ray(V3<R> o, V3<R> d) {
    SlabRay<R> sr;
    if constexpr (sizeof(R) == 8) {
        const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
        sr.slack = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float o32 = float(oo[a]);
            float inv = rt_rcp(float(dd[a]));
            // an axis the ray is exactly parallel to (or whose direction component underflows in f32): 1/d = inf would make
            // the ONE slack below infinite and the whole walk lose its culling.  NaN instead: every plane distance of this
            // axis is then NaN, which the maxNum / minNum of slab4_planes drop — the axis never culls (conservative: the
            // exact test could at most cull more) and stays out of the slack.
            if (!(rt_fabs(inv) < __builtin_huge_valf())) inv = __builtin_nanf("");
            sr.inv[a] = inv;
            sr.oinv[a] = o32 * inv;
            sr.slack = rt_max(sr.slack, rt_fabs(sr.oinv[a]) * 2.4e-7f); // maxNum: a NaN axis drops out
#if RT_F64_SLAB_FOLDED
            // The widening folded into the per-walk constants, as in the f32 kernels' SLAB_FMA_FOLDED form: near planes take
            // inv (1 - 4.8e-7) and oi (1 - 4.8e-7) + 3.0e-7 |oi|, far planes the opposite.  With r1, r2, r3 the roundings of inv_n, of
            // oinv_n's fma and of the plane's fma (each <= 2^-24):
            //   t_n32 = (1 - 4.8e-7)(1 + e1)(1 + r3) [ t (1 + r1) + o inv (r1 - e0 - e2 - r2) ] - slack (1 + r2)(1 + r3)
            // — for t >= 0 the first factor times (1 + r1) is <= 1 (4.8e-7 >= |e1| + |r1| + |r3| = 3.0e-7) and the second term is at most
            // 2.4e-7 |o inv| < slack: t_n32 <= t.  A negative near distance cannot set the entry (t_min >= 0 or another axis does), a
            // negative far distance is a box behind the origin; a NaN axis drops out of max3 / min3.
            const float slack_a = rt_fabs(sr.oinv[a]) * 3.0e-7f;
            sr.inv_n[a] = inv * (1.f - 4.8e-7f);
            sr.inv_f[a] = inv * (1.f + 4.8e-7f);
            sr.oinv_n[a] = __builtin_fmaf(sr.oinv[a], 1.f - 4.8e-7f, slack_a);
            sr.oinv_f[a] = __builtin_fmaf(sr.oinv[a], 1.f + 4.8e-7f, -slack_a);
#if RT_F64_SLAB_FOLDED == 2
            sr.oinv_n[a] = sr.oinv[a] + slack_a;
            sr.oinv_f[a] = sr.oinv[a] - slack_a;
#endif
#endif
        }
    } else {
        sr.inv = V3<R>(rt_rcp(d.x), rt_rcp(d.y), rt_rcp(d.z));
        if constexpr (FORM == SLAB_FMA_FOLDED) {
            // t32 = fl(b inv - oi), oi = fl(o inv), inv = rcp(d) (1 ulp): |t32 - t| <= 1.8e-7 |t| + 6e-8 |o inv|; the widening by
            // 4.2e-7 |t| + 3.0e-7 |o_a inv_a| is folded into the constants: scaling by (1 -+ 4.2e-7) widens a positive distance —
            // a negative near distance cannot set the entry (another axis's positive one or t_min does), a negative far
            // distance means a box behind the origin — and an axis-parallel ray's NaNs drop out of max3 / min3.
            const float oo[3] = {o.x, o.y, o.z}, ii[3] = {sr.inv.x, sr.inv.y, sr.inv.z};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float inv = ii[a];
                if (!(rt_fabs(inv) < __builtin_huge_valf())) inv = __builtin_nanf("");
                const float oi = oo[a] * inv, slack = rt_fabs(oi) * 3.0e-7f;
                sr.inv_n[a] = inv * (1.f - 4.2e-7f);
                sr.inv_f[a] = inv * (1.f + 4.2e-7f);
                sr.oinv_n[a] = __builtin_fmaf(oi, 1.f - 4.2e-7f, slack);
                sr.oinv_f[a] = __builtin_fmaf(oi, 1.f + 4.2e-7f, -slack);
            }
        }
    }
    return sr;
}
// 1 / d of axis a as the walk's slab tests hold it
template <typename R> RT_HD float slab_inv_of(const SlabRay<R>& sr, int a) {
    if constexpr (sizeof(R) == 8) return sr.inv[a];
    else return a == 0 ? sr.inv.x : (a == 1 ? sr.inv.y : sr.inv.z);
}
// A 4-wide record as a lane holds it for a node step: the NEAR and the FAR plane of every axis for the four children
// (which of lo / hi is the near one depends on the sign of the ray direction only, so the lane fetches them by address —
// NodePlanes::near_q, set once per walk — instead of selecting per child), and the four child slots.
struct Planes4 {
    float nr[3][4], fr[3][4];
    int32_t child[4];
};
// piece of the record (16-byte units: 0-2 lo.xyz, 3-5 hi.xyz, 6 children) holding the near planes of axis a
template <typename R> RT_HD uint32_t near_piece(int a, const SlabRay<R>& sr) {
    if constexpr (sizeof(R) == 8) return uint32_t(a) + (sr.inv[a] < 0.f ? 3u : 0u);
    else return uint32_t(a) + ((a == 0 ? sr.inv.x : (a == 1 ? sr.inv.y : sr.inv.z)) < R(0) ? 3u : 0u);
}
// The four children's entry / exit distances against their boxes alone (no [t_min, closest] yet): tn[c] = the largest
// near-plane distance, tf[c] = the smallest far-plane distance.  (As 24 PACKED f32 operations — v_pk_add_f32 / v_pk_mul_f32
// with the ray's component splat by op_sel, which the by-axis layout makes natural — it measured SLOWER: final_scene f32
// 1389 against 1452, spheres_1m 340 against 397: on wave64 a packed operation issues no faster than its two halves.)
RT_HD void slab4_planes_fma(const Planes4& nd, const float oinv[3], const float inv[3], float tn[4], float tf[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float nx = __builtin_fmaf(nd.nr[0][c], inv[0], -oinv[0]), fx = __builtin_fmaf(nd.fr[0][c], inv[0], -oinv[0]);
        const float ny = __builtin_fmaf(nd.nr[1][c], inv[1], -oinv[1]), fy = __builtin_fmaf(nd.fr[1][c], inv[1], -oinv[1]);
        const float nz = __builtin_fmaf(nd.nr[2][c], inv[2], -oinv[2]), fz = __builtin_fmaf(nd.fr[2][c], inv[2], -oinv[2]);
        tn[c] = rt_max(nz, rt_max(ny, nx));
        tf[c] = rt_min(fz, rt_min(fy, fx));
    }
}
RT_HD void slab4_planes(const Planes4& nd, const float o[3], const float inv[3], float tn[4], float tf[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float nx = (nd.nr[0][c] - o[0]) * inv[0], fx = (nd.fr[0][c] - o[0]) * inv[0];
        const float ny = (nd.nr[1][c] - o[1]) * inv[1], fy = (nd.fr[1][c] - o[1]) * inv[1];
        const float nz = (nd.nr[2][c] - o[2]) * inv[2], fz = (nd.fr[2][c] - o[2]) * inv[2];
        tn[c] = rt_max(nz, rt_max(ny, nx));
        tf[c] = rt_min(fz, rt_min(fy, fx));
    }
}
// Hit test of the four children against [lo_t, hi_t]; e[c] = the entry distance (>= lo_t: the ordering key).
// f64 kernels: the conservative form.  With k the plane that sets tn, the true entry is >= t_k >= n_k - (3.6e-7 |n_k| +
// slack_k), so tn - 3.6e-7 |tn| - slack (slack = the largest axis slack) is a lower bound of the true entry, and likewise
// an upper bound of the true exit: ONE widening per box instead of one per plane (round 1), a box the exact f64 test
// would pass still always passes.  NaN / inf (axis-parallel rays) never cull.
template <int FORM> RT_HD void slab_hit4(const Planes4& nd, V3<double>, const SlabRay<double>& sr, float lo_t, float hi_t, float e[4], bool h[4]) {
#if RT_F64_SLAB_FOLDED == 3
    {
        float tn[4], tf[4];
        slab4_planes_fma(nd, sr.oinv, sr.inv, tn, tf);
#pragma unroll
        for (int c = 0; c < 4; ++c) { // 13 operations per box: one fma per end widens it (valid for the non-negative distances that matter)
            const float n = __builtin_fmaf(tn[c], 1.f - 3.6e-7f, -sr.slack);
            const float f = __builtin_fmaf(tf[c], 1.f + 3.6e-7f, sr.slack);
            const float lo = rt_max(n, lo_t), hi = rt_min(f, hi_t);
            e[c] = lo;
            h[c] = !(hi < lo);
        }
        return;
    }
#elif RT_F64_SLAB_FOLDED == 2
#pragma unroll
    for (int c = 0; c < 4; ++c) { // 12 operations per box: the slack in the constants, the relative widening on the entry alone
        const float nx = __builtin_fmaf(nd.nr[0][c], sr.inv[0], -sr.oinv_n[0]), fx = __builtin_fmaf(nd.fr[0][c], sr.inv[0], -sr.oinv_f[0]);
        const float ny = __builtin_fmaf(nd.nr[1][c], sr.inv[1], -sr.oinv_n[1]), fy = __builtin_fmaf(nd.fr[1][c], sr.inv[1], -sr.oinv_f[1]);
        const float nz = __builtin_fmaf(nd.nr[2][c], sr.inv[2], -sr.oinv_n[2]), fz = __builtin_fmaf(nd.fr[2][c], sr.inv[2], -sr.oinv_f[2]);
        const float lo = rt_max(rt_max(nz, rt_max(ny, nx)) * (1.f - 9.6e-7f), lo_t), hi = rt_min(rt_min(fz, rt_min(fy, fx)), hi_t);
        e[c] = lo;
        h[c] = !(hi < lo);
    }
    return;
#elif RT_F64_SLAB_FOLDED
#pragma unroll
    for (int c = 0; c < 4; ++c) { // 11 operations per box instead of 15: the widening is in the constants (slab_ray)
        const float nx = __builtin_fmaf(nd.nr[0][c], sr.inv_n[0], -sr.oinv_n[0]), fx = __builtin_fmaf(nd.fr[0][c], sr.inv_f[0], -sr.oinv_f[0]);
        const float ny = __builtin_fmaf(nd.nr[1][c], sr.inv_n[1], -sr.oinv_n[1]), fy = __builtin_fmaf(nd.fr[1][c], sr.inv_f[1], -sr.oinv_f[1]);
        const float nz = __builtin_fmaf(nd.nr[2][c], sr.inv_n[2], -sr.oinv_n[2]), fz = __builtin_fmaf(nd.fr[2][c], sr.inv_f[2], -sr.oinv_f[2]);
        const float lo = rt_max(rt_max(nz, rt_max(ny, nx)), lo_t), hi = rt_min(rt_min(fz, rt_min(fy, fx)), hi_t); // maxNum / minNum: a NaN drops out
        e[c] = lo;
        h[c] = !(hi < lo);
    }
    return;
#endif
    float tn[4], tf[4];
    slab4_planes_fma(nd, sr.oinv, sr.inv, tn, tf);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float n = __builtin_fmaf(-rt_fabs(tn[c]), 3.6e-7f, tn[c]) - sr.slack;
        const float f = __builtin_fmaf(rt_fabs(tf[c]), 3.6e-7f, tf[c]) + sr.slack;
        const float lo = rt_max(n, lo_t), hi = rt_min(f, hi_t); // maxNum / minNum: a NaN drops out
        e[c] = lo;
        h[c] = !(hi < lo);
    }
}
template <int FORM> RT_HD void slab_hit4(const Planes4& nd, V3<float> o, const SlabRay<float>& sr, float tmin, float tmax, float e[4], bool h[4]) {
    if constexpr (FORM == SLAB_FMA_FOLDED) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float nx = __builtin_fmaf(nd.nr[0][c], sr.inv_n[0], -sr.oinv_n[0]), fx = __builtin_fmaf(nd.fr[0][c], sr.inv_f[0], -sr.oinv_f[0]);
            const float ny = __builtin_fmaf(nd.nr[1][c], sr.inv_n[1], -sr.oinv_n[1]), fy = __builtin_fmaf(nd.fr[1][c], sr.inv_f[1], -sr.oinv_f[1]);
            const float nz = __builtin_fmaf(nd.nr[2][c], sr.inv_n[2], -sr.oinv_n[2]), fz = __builtin_fmaf(nd.fr[2][c], sr.inv_f[2], -sr.oinv_f[2]);
            const float lo = rt_max(rt_max(nz, rt_max(ny, nx)), tmin), hi = rt_min(rt_min(fz, rt_min(fy, fx)), tmax); // maxNum / minNum: a NaN drops out
            e[c] = lo;
            h[c] = !(hi < lo);
        }
        return;
    }
    const float oo[3] = {o.x, o.y, o.z}, inv[3] = {sr.inv.x, sr.inv.y, sr.inv.z};
    float tn[4], tf[4];
    slab4_planes(nd, oo, inv, tn, tf);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float lo = rt_max(tn[c], tmin), hi = rt_min(tf[c], tmax);
        e[c] = lo;
        // f32: absorb the rounding of (bound - o) * inv and of the 1-2 ulp reciprocal (boxes are already padded)
        h[c] = !((hi > 0.f ? hi * 1.0000005f : hi) < lo);
    }
}
// the walk's [t_min, closest] as the floats the slab tests compare against (f64: rounded outward once per node)
RT_HD void slab_range(double tmin, double tmax, float& lo_t, float& hi_t) {
    lo_t = float(tmin); hi_t = float(tmax);
    lo_t = __builtin_fmaf(-rt_fabs(lo_t), 3.6e-7f, lo_t); // outward: float() rounds to nearest
    hi_t = __builtin_fmaf(rt_fabs(hi_t), 3.6e-7f, hi_t);
}
RT_HD void slab_range(float tmin, float tmax, float& lo_t, float& hi_t) { lo_t = tmin; hi_t = tmax; }
RT_HD uint32_t float_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
#endif
}

// ---------------------------------------------------------------- primitive tests: t only
// Sphere::hit — hittable.rs:87-109 (bounds inclusive: only `<` / `>` reject, Q10)
// f64 evaluates the reference's expression literally.  In f32 the textbook discriminant
// hb^2 - a*(|oc|^2 - r^2) cancels catastrophically for a small sphere far from the ray origin (the
// final_scene cluster: r = 10 seen from 1100 units, |oc|^2 ~ 1.2e6 has an ulp of 0.125), so the f32
// instantiation uses the algebraically identical, cancellation-free form  a*(r^2 - |oc - (hb/a) d|^2).
template <typename R> RT_HD R sphere_discriminant(V3<R> oc, V3<R> d, R a, R half_b, R radius) {
    if constexpr (sizeof(R) == 4) {
        V3<R> l = oc - rt_div(half_b, a) * d; // from the centre to the closest point of the ray's line
        return a * (radius * radius - dot(l, l));
    } else {
        R c = dot(oc, oc) - radius * radius;
        return half_b * half_b - a * c;
    }
}
template <typename R>
RT_HD bool sphere_t(V3<R> center, R radius, const Ray<R>& ray, R t_min, R t_max, R& t_out) {
    V3<R> oc = ray.o - center;
    // (the two roots share |d|^2's reciprocal.  Keeping it with the walk's ray instead — it is the same for every sphere the
    // ray meets — cost the f64 kernels more in registers than the v_rcp_f64 it saved: final_scene f64 970 against 1002)
    const R a = dot(ray.d, ray.d);
    const Recip<R> inv_a = recip_of(a);
    R half_b = dot(oc, ray.d);
    R disc = sphere_discriminant(oc, ray.d, a, half_b, radius);
    if (disc < R(0)) return false;
    R sqrtd = rt_sqrt(disc);
    // (range tests written so that NaN — a zero direction — is out of range, like the infinities IEEE division gives there)
    R root = div_by(-half_b - sqrtd, inv_a);
    if (!(root >= t_min && t_max >= root)) {
        root = div_by(-half_b + sqrtd, inv_a);
        if (!(root >= t_min && t_max >= root)) return false;
    }
    t_out = root;
    return true;
}
// MovingSphere::center — hittable.rs:187-191
template <typename R> RT_HD V3<R> moving_center(const MovingSphereRec<R>& m, R time) {
    return V3<R>(m.c0) + rt_div(time - m.t0, m.t1 - m.t0) * (V3<R>(m.c1) - V3<R>(m.c0));
}
// Rectangle::hit — hittable.rs:503-513 (half-open extents, Q9).  plane: 0 XY, 1 XZ, 2 YZ.
template <typename R>
RT_HD bool rect_t(int plane, R a0, R a1, R b0, R b1, R k, const Ray<R>& ray, R t_min, R t_max, R& t_out, R& a_out, R& b_out) {
    // (axis0, axis1, k-axis): XY -> (x,y,z); XZ -> (x,z,y); YZ -> (y,z,x) — hittable.rs:450-488
    // The components are copied into scalars first: `c ? ray.o.z : ray.o.y` on member lvalues is an lvalue, i.e. a
    // pointer select into the ray, which keeps the whole lane state from being promoted to registers.
    const R ox = ray.o.x, oy = ray.o.y, oz = ray.o.z, dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
    const R ok = plane == 0 ? oz : (plane == 1 ? oy : ox);
    const R dk = plane == 0 ? dz : (plane == 1 ? dy : dx);
    const R oa = plane == 2 ? oy : ox, da = plane == 2 ? dy : dx;
    const R ob = plane == 0 ? oy : oz, db = plane == 0 ? dy : dz;
    // straight-line: `t < t_min || t > t_max` and the half-open extents as one predicate (a NaN t passes the range
    // test and fails the extents, as in the reference)
    const R t = rt_div(k - ok, dk);
    const R a = oa + t * da;
    const R b = ob + t * db;
    const bool hit = !(t < t_min) & !(t > t_max) & (a0 <= a) & (a < a1) & (b0 <= b) & (b < b1);
    if (hit) { t_out = t; a_out = a; b_out = b; }
    return hit;
}
// Cube::hit = List::hit over its six rectangles in the order xy@min.z, xy@max.z, xz@min.y, xz@max.y,
// yz@min.x, yz@max.x with a shrinking `closest` (hittable.rs:560-569,153-163).  Returns the winning
// face index in `face` (later face wins exact ties, like the List).
template <typename R>
RT_HD bool box_t(const BoxRec<R>& bx, const Ray<R>& ray, R t_min, R t_max, R& t_out, int& face) {
    bool any = false;
    R closest = t_max;
    int fc = 0;
    const R o[3] = {ray.o.x, ray.o.y, ray.o.z}, d[3] = {ray.d.x, ray.d.y, ray.d.z};
    R tf[6]; // (k - origin) / direction of the six faces (Rectangle::hit, hittable.rs:503-513): the two faces of an axis share the divisor
#pragma unroll
    for (int plane = 0; plane < 3; ++plane) {
        const int ik = plane == 0 ? 2 : (plane == 1 ? 1 : 0);
        rt_div2(bx.mn[ik] - o[ik], bx.mx[ik] - o[ik], d[ik], tf[2 * plane], tf[2 * plane + 1]);
    }
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        const int plane = f >> 1;                            // 0 XY, 1 XZ, 2 YZ
        const int ia = plane == 2 ? 1 : 0, ib = plane == 0 ? 1 : 2;
        const R t = tf[f];
        const R a = o[ia] + t * d[ia], b = o[ib] + t * d[ib];
        const bool hit = !(t < t_min) & !(t > closest) & (bx.mn[ia] <= a) & (a < bx.mx[ia]) & (bx.mn[ib] <= b) & (b < bx.mx[ib]);
        closest = hit ? t : closest;
        fc = hit ? f : fc;
        any |= hit;
    }
    t_out = closest;
    if (any) face = fc;
    return any;
}

// Cube::hit with ONE exact quotient instead of six, where that is provably the same (round 5).  The six rectangle tests of box_t answer "which face
// does the ray meet first inside [t_min, closest]"; for a ray that is clear of the box's edges the only face that can is known from the walk's own f32
// plane distances (SlabRay: the constants of the node steps), and only that face's t has to be the reference's: t = (k - o) / d, the same expression
// on the same operands, with the two range tests run on it exactly.  With N_k / F_k the near / far plane distances of axis k (approximate: |error| <=
// 3e-7 (|t| + |o / d|), the bound derived at SlabRay<double>), tn = max N, tf = min F, N2 / F2 the second largest / smallest, and a gap
// g = BOX_FAST_MARGIN x (sum |o_k / d_k| + |tn| + |tf|):
//   * no face can hit when tn - tf >= g (the line passes the box: every face's point lies outside another axis' slab by g), when t_min - tf >= g
//     (the box ends before t_min) or when tn - closest >= g (it begins beyond the incumbent);
//   * ENTRY — tn - t_min >= g, tn - N2 >= g, tf - tn >= g: the entry face's point is inside both other slabs by g (its extents test is TRUE), every
//     other front face's point is outside the entry axis' slab by g (FALSE), every back face lies beyond by g (a larger t: cannot win, cannot tie):
//     the entry face if its exact t passes the two range tests, nothing otherwise;
//   * EXIT — t_min - tn >= g (every front face's t is below t_min: the origin is inside, or ON the face it has just scattered off), F2 - tf >= g,
//     tf - tn >= g: the exit face's point is inside the other slabs by g (TRUE), the other back faces' points outside the exit axis' slab by g
//     (FALSE): the exit face if its exact t is in range — the ray leaving the cube it scattered off has t ~ 1e-14 < t_min there: nothing.
// A gap of g in t is |d_a| g >= MARGIN (|o_a| + |t d_a|) in the coordinate the extents test compares — 10^11 times the rounding error of evaluating
// o_a + t d_a in f64 (2^-52 of the same sum), 60 times the f32 error of the plane distances themselves.  Everything else — a ray within g of an
// edge, an entry within g of t_min, any NaN / infinity (axis-parallel rays: S is not finite, no comparison holds) — takes box_t.  `face` as box_t.
// tests/test_core_parity_cpu.py::test_fast_cube_test_is_the_six_rectangle_test holds the two to the same (hit, t, face) bit for bit.
#ifndef RT_BOX_FAST
#define RT_BOX_FAST 1
#endif
#ifndef RT_BOX_FAST_MARGIN
#define RT_BOX_FAST_MARGIN 2e-5f
#endif
RT_HD float rt_med3(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(a, b, c);
#else
    return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
#endif
}
template <typename R> RT_HD float slab_inv32(const SlabRay<R>& sr, int a) {
    if constexpr (sizeof(R) == 8) return sr.inv[a];
    else return a == 0 ? sr.inv.x : (a == 1 ? sr.inv.y : sr.inv.z);
}
// returns 0: certainly no face hits; 1: the only face that can is (axis, use_mx), and its extents test is true; 2: not certain (box_t decides)
// FORM: the same verdicts written two ways — 0: comparison by comparison, 1: one comparison of a min3 / max3 per verdict (fewer instructions).
// The kernels take whichever their build runs faster with (final_scene f64 1664 -> 1677 with form 1, RTTNW_F64_STRICT 1399 -> 1386: that build
// keeps form 0); tests/hostsim holds BOTH to box_t.
#if defined(RT_STRICT_F64)
constexpr int BOX_CLASSIFY_FORM = 0;
#else
constexpr int BOX_CLASSIFY_FORM = 1;
#endif
template <int FORM = BOX_CLASSIFY_FORM, typename R>
RT_HD int box_classify(const BoxRec<R>& bx, const Ray<R>& ray, const SlabRay<R>& sr, R t_min, R t_max, int& axis, bool& use_mx) {
    const R o[3] = {ray.o.x, ray.o.y, ray.o.z};
    float nr[3], fr[3], S = 0.f;
    bool ng[3]; // (no array is indexed by a run-time axis below: that would put it — and the walk's state around it — in scratch)
    bool inverted = false; // a cube built from corners with mn > mx on an axis (rttnw_cube stores what it is given, like Cube::new hittable.rs:551-558):
                           // the reference's six half-open rectangles still hit the two faces across that axis, while near / far below — chosen by the
                           // direction's sign alone — would swap and the verdicts with them: such a record is box_t's (round-5 advisor)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float inv = slab_inv32(sr, a);
        float oi;
        if constexpr (sizeof(R) == 8) oi = sr.oinv[a]; else oi = float(o[a]) * inv;
        const float mn32 = float(bx.mn[a]), mx32 = float(bx.mx[a]);
        inverted |= mn32 > mx32;
        const float t0 = __builtin_fmaf(mn32, inv, -oi), t1 = __builtin_fmaf(mx32, inv, -oi);
        const bool neg = inv < 0.f;
        ng[a] = neg;
        nr[a] = neg ? t1 : t0;
        fr[a] = neg ? t0 : t1;
        S += rt_fabs(oi);
    }
    const float tn = rt_max(rt_max(nr[0], nr[1]), nr[2]), tf = rt_min(rt_min(fr[0], fr[1]), fr[2]);
    const float n2 = rt_med3(nr[0], nr[1], nr[2]), f2 = rt_med3(fr[0], fr[1], fr[2]);
    const float g = (S + rt_fabs(tn) + rt_fabs(tf)) * RT_BOX_FAST_MARGIN;
    const float lo = float(t_min), hi = float(t_max);
    if (inverted) return 2;
    if constexpr (FORM == 0) {
        if (tn - tf >= g || lo - tf >= g || tn - hi >= g) return 0;
        if (!(tf - tn >= g)) return 2;
        if (tn - lo >= g) { // the entry face, or nothing
            if (!(tn - n2 >= g)) return 2;
            axis = nr[0] >= nr[1] ? (nr[0] >= nr[2] ? 0 : 2) : (nr[1] >= nr[2] ? 1 : 2);
            use_mx = axis == 0 ? ng[0] : (axis == 1 ? ng[1] : ng[2]);
            return 1;
        }
        if (lo - tn >= g && f2 - tf >= g) { // the exit face, or nothing
            axis = fr[0] <= fr[1] ? (fr[0] <= fr[2] ? 0 : 2) : (fr[1] <= fr[2] ? 1 : 2);
            use_mx = !(axis == 0 ? ng[0] : (axis == 1 ? ng[1] : ng[2]));
            return 1;
        }
        return 2;
    }
    // (each verdict is ONE comparison of a min3 / max3 against the gap — a NaN anywhere makes the comparison false; the minNum / maxNum of the
    // instructions would drop a NaN operand, but g itself is NaN then: S sums the |o / d| every distance is made from)
    const float inside = tf - tn, before = lo - tn;
    if (rt_max(rt_max(-inside, lo - tf), tn - hi) >= g) return 0;
    const bool entry = rt_min(rt_min(-before, tn - n2), inside) >= g;   // the entry face, or nothing
    const bool exit_ = rt_min(rt_min(before, f2 - tf), inside) >= g;    // the exit face, or nothing (never both: -before and before)
    if (!(entry | exit_)) return 2;
    // the face's axis: the one whose distance IS the extreme (unique by the gap)
    const float c0 = entry ? nr[0] : fr[0], c1 = entry ? nr[1] : fr[1], ce = entry ? tn : tf;
    axis = c0 == ce ? 0 : (c1 == ce ? 1 : 2);
    use_mx = (axis == 0 ? ng[0] : (axis == 1 ? ng[1] : ng[2])) == entry;
    return 1;
}
// (`bx`: the record WHERE IT LIES — LDS or global memory: the classification reads its six bounds once and keeps them as floats, the exact quotient
// reads the one bound it needs by index, the rare undecided ray leaves the whole record to box_t: no copy of it is held in registers)
template <int FORM = BOX_CLASSIFY_FORM, typename R>
RT_HD bool box_t_fast(const BoxRec<R>& bx, const Ray<R>& ray, const SlabRay<R>& sr, R t_min, R t_max, R& t_out, int& face) {
    int axis = 0;
    bool use_mx = false;
    const int verdict = box_classify<FORM>(bx, ray, sr, t_min, t_max, axis, use_mx);
    if (verdict == 2) { const BoxRec<R> whole = bx; return box_t(whole, ray, t_min, t_max, t_out, face); }
    if (verdict == 0) return false;
    // (components copied into scalars first: a conditional on member lvalues is a pointer select into the ray — see rect_t)
    const R ox = ray.o.x, oy = ray.o.y, oz = ray.o.z, dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
    const R ok = axis == 0 ? ox : (axis == 1 ? oy : oz);
    const R dk = axis == 0 ? dx : (axis == 1 ? dy : dz);
    static_assert(offsetof(BoxRec<R>, mx) == 3 * sizeof(R), "mn[3] then mx[3]");
    const R k = (&bx.mn[0])[axis + (use_mx ? 3 : 0)]; // (a load by index from the record in memory, not a register select)
    R t, unused;
    rt_div2(k - ok, k - ok, dk, t, unused); // (the quotient box_t makes for this face: Rectangle::hit's t, hittable.rs:504)
    const bool hit = !(t < t_min) & !(t > t_max);
    if (hit) {
        t_out = t;
        face = 2 * (2 - axis) + (use_mx ? 1 : 0); // box_t's numbering: xy@min.z, xy@max.z, xz@min.y, xz@max.y, yz@min.x, yz@max.x
    }
    return hit;
}

template <typename R> RT_HD const InstanceHead<R>& head_of(const InstanceRec<R>& in) { return reinterpret_cast<const InstanceHead<R>&>(in); }

// Ray -> object space through the ops of an instance (Translate::hit :600-604, YRotate::hit :687-697): a chain of up to
// FAST_INSTANCE_OPS wrappers (copied by value, unrolled) ...
template <typename R> RT_HD Ray<R> to_object_fast(const InstanceHead<R>& in, Ray<R> ray, int n) {
#pragma unroll
    for (int i = 0; i < FAST_INSTANCE_OPS; ++i) {
        if (i < n) {
            if (in.ops[i].type == OP_TRANSLATE) {
                ray.o = ray.o - V3<R>(in.ops[i].v);
            } else {
                R s = in.ops[i].v[0], c = in.ops[i].v[1];
                R ox = c * ray.o.x - s * ray.o.z, oz = s * ray.o.x + c * ray.o.z;
                R dx = c * ray.d.x - s * ray.d.z, dz = s * ray.d.x + c * ray.d.z;
                ray.o.x = ox; ray.o.z = oz; ray.d.x = dx; ray.d.z = dz;
            }
        }
    }
    return ray;
}
// ... and any longer one, op by op through the record in memory.
template <typename R> RT_HD void to_object_general(const InstanceRec<R>* in, Ray<R>* ray, int n) {
    Ray<R> r = *ray;
    for (int i = 0; i < n; ++i) {
        const int32_t type = in->ops[i].type;
        const R v0 = in->ops[i].v[0], v1 = in->ops[i].v[1], v2 = in->ops[i].v[2];
        if (type == OP_TRANSLATE) {
            r.o = r.o - V3<R>(v0, v1, v2);
        } else {
            const R s = v0, c = v1;
            const R ox = c * r.o.x - s * r.o.z, oz = s * r.o.x + c * r.o.z;
            const R dx = c * r.d.x - s * r.d.z, dz = s * r.d.x + c * r.d.z;
            r.o.x = ox; r.o.z = oz; r.d.x = dx; r.d.z = dz;
        }
    }
    *ray = r;
}
template <bool G, typename R> RT_HD Ray<R> to_object_n(const InstanceRec<R>& in, Ray<R> ray, int n) {
    if (!G || n <= FAST_INSTANCE_OPS) { // !G: the host selected this kernel because no chain of the scene is longer
        const InstanceHead<R> h = head_of(in);
        return to_object_fast(h, ray, n);
    }
    if constexpr (G) to_object_general(&in, &ray, n);
    return ray;
}
template <bool G, typename R> RT_HD Ray<R> to_object(const InstanceRec<R>& in, Ray<R> ray) { return to_object_n<G>(in, ray, head_of(in).n_ops); }

template <typename R> RT_HD void face_normal(V3<R> dir, V3<R> outward, V3<R>& normal, bool& front);
// Unwind a hit record (p, normal, front_face) through ops[n-1] .. ops[0], innermost first, with the reference's face_normal
// call at every level — level i sees the ray direction after ops[0..i] (Translate's `moved_ray` :607, YRotate's rotated
// ray :706) — and, under quirk Q1, its overwritten-x back-rotation (hittable.rs:700-706, :607-611).
template <typename R> RT_HD void unwind_op(int32_t type, R v0, R v1, R v2, V3<R> dir, uint32_t quirks, V3<R>& p, V3<R>& normal, bool& front_face) {
    if (type == OP_ROTATE_Y) { // hittable.rs:700-706
        const R s = v0, c = v1;
        const R px = c * p.x + s * p.z;
        const R nx = c * normal.x + s * normal.z;
        // Q1: the reference's z line reads the x it has just overwritten
        const R pxz = (quirks & 1u) ? px : p.x;
        const R nxz = (quirks & 1u) ? nx : normal.x;
        const R pz = -s * pxz + c * p.z;
        const R nz = -s * nxz + c * normal.z;
        p.x = px; p.z = pz;
        normal.x = nx; normal.z = nz;
    } else { // hittable.rs:607-611
        p = p + V3<R>(v0, v1, v2);
    }
    V3<R> nn; bool ff;
    face_normal(dir, normal, nn, ff);
    normal = nn; front_face = ff;
}
template <typename R> RT_HD void unwind_fast(const InstanceHead<R>& in, int n, V3<R> world_dir, uint32_t quirks, V3<R>& p, V3<R>& normal, bool& front_face) {
    V3<R> dirs[FAST_INSTANCE_OPS];
    {
        V3<R> d = world_dir;
#pragma unroll
        for (int i = 0; i < FAST_INSTANCE_OPS; ++i) {
            if (i < n && in.ops[i].type == OP_ROTATE_Y) {
                R s = in.ops[i].v[0], c = in.ops[i].v[1];
                R dx = c * d.x - s * d.z, dz = s * d.x + c * d.z;
                d.x = dx; d.z = dz;
            }
            dirs[i] = d;
        }
    }
#pragma unroll
    for (int i = FAST_INSTANCE_OPS - 1; i >= 0; --i)
        if (i < n) unwind_op(in.ops[i].type, in.ops[i].v[0], in.ops[i].v[1], in.ops[i].v[2], dirs[i], quirks, p, normal, front_face);
}
template <typename R> RT_HD void unwind_general(const InstanceRec<R>* in, int n, V3<R> world_dir, uint32_t quirks, V3<R>* p_io, V3<R>* n_io, bool* ff_io) {
    V3<R> p = *p_io, normal = *n_io;
    bool ff = *ff_io;
    for (int i = n - 1; i >= 0; --i) {
        V3<R> d = world_dir; // the direction after ops[0..i]
        for (int k = 0; k <= i; ++k)
            if (in->ops[k].type == OP_ROTATE_Y) {
                const R s = in->ops[k].v[0], c = in->ops[k].v[1];
                const R dx = c * d.x - s * d.z, dz = s * d.x + c * d.z;
                d.x = dx; d.z = dz;
            }
        unwind_op(in->ops[i].type, in->ops[i].v[0], in->ops[i].v[1], in->ops[i].v[2], d, quirks, p, normal, ff);
    }
    *p_io = p; *n_io = normal; *ff_io = ff;
}
template <bool G, typename R>
RT_HD void unwind_record(const InstanceRec<R>& in, int n, V3<R> world_dir, uint32_t quirks, V3<R>& p, V3<R>& normal, bool& front_face) {
    if (!G || n <= FAST_INSTANCE_OPS) {
        const InstanceHead<R> h = head_of(in);
        unwind_fast(h, n, world_dir, quirks, p, normal, front_face);
    } else {
        if constexpr (G) unwind_general(&in, n, world_dir, quirks, &p, &normal, &front_face);
    }
}
// The ray DIRECTION after ops[0..upto] (a medium inside a transformed group measures its free flight along that ray).
template <bool G, typename R> RT_HD V3<R> dir_after(const InstanceRec<R>& in, V3<R> d, int upto) {
    Ray<R> r;
    r.o = V3<R>(); r.d = d; r.time = R(0);
    return to_object_n<G>(in, r, upto + 1).d;
}

// Rectangle hit coordinates at parameter t, no range checks (used to rebuild the winner's record).
template <typename R> RT_HD void rect_ab(int plane, const Ray<R>& ray, R t, R& a, R& b) {
    const R ox = ray.o.x, oy = ray.o.y, oz = ray.o.z, dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
    const R oa = plane == 2 ? oy : ox, da = plane == 2 ? dy : dx;
    const R ob = plane == 0 ? oy : oz, db = plane == 0 ? dy : dz;
    a = oa + t * da;
    b = ob + t * db;
}

// One primitive, t only (+ `aux`: the winning face of a box).  `ray` is in the primitive's space.
// (HAVE_SR: `sr` holds the walk's slab constants and `ray` is the ray they were made from — the fast cube test reads its plane distances off them)
template <bool NO_TIME, bool HAVE_SR, typename R>
RT_HD bool prim_t(const SceneView<R>& sc, uint32_t kind, uint32_t idx, const Ray<R>& ray, R t_min, R t_max, R& t, int& aux, const SlabRay<R>& sr) {
    if (kind == PRIM_SPHERE) {
        SphereRec<R> s = sc.spheres[idx];
        return sphere_t(V3<R>(s.cx, s.cy, s.cz), s.r, ray, t_min, t_max, t);
    } else if (kind == PRIM_BOX) {
#if RT_BOX_FAST
        if constexpr (HAVE_SR) return box_t_fast(sc.boxes[idx], ray, sr, t_min, t_max, t, aux);
#endif
        const BoxRec<R> bx = sc.boxes[idx];
        return box_t(bx, ray, t_min, t_max, t, aux);
    } else if (kind == PRIM_RECT) {
        const RectRec<R> r = sc.rects[idx];
        R a, b;
        return rect_t(r.plane, r.a0, r.a1, r.b0, r.b1, r.k, ray, t_min, t_max, t, a, b);
    } else if (!NO_TIME && kind == PRIM_MOVING_SPHERE) { // (NO_TIME: the scene has none)
        const MovingSphereRec<R> m = sc.moving[idx];
        return sphere_t(moving_center(m, ray.time), m.r, ray, t_min, t_max, t);
    }
    return false;
}
template <bool NO_TIME = false, typename R>
RT_HD bool prim_t(const SceneView<R>& sc, uint32_t kind, uint32_t idx, const Ray<R>& ray, R t_min, R t_max, R& t, int& aux) {
    return prim_t<NO_TIME, false>(sc, kind, idx, ray, t_min, t_max, t, aux, SlabRay<R>());
}

// Sequence number of a primitive record (tie-break only; see rt_types.hpp).
template <typename R> RT_HD int32_t prim_seq(const SceneView<R>& sc, uint32_t kind, uint32_t idx) {
    if (kind == PRIM_SPHERE) return sc.sphere_seq[idx];
    if (kind == PRIM_BOX) return sc.boxes[idx].seq;
    if (kind == PRIM_RECT) return sc.rects[idx].seq;
    return sc.moving[idx].seq;
}

struct HitRef {
    int32_t prim; // make_ref(kind, index); kind PRIM_NONE = miss
    int32_t inst; // enclosing instance or -1
    int32_t aux;  // box: winning face 0..5
};

// ---------------------------------------------------------------- closest solid hit
// Flat-BVH walk with a per-lane stack (LDS on the device).  Semantics of the reference's
// List::hit / BvhTree::hit: closest t in [t_min, t_max], a candidate with t == closest replaces
// the incumbent (hittable.rs:157-159,366-367).  Only (t, primitive, instance) are tracked here;
// the hit record is rebuilt once for the winner (Q11: same result, far fewer acos/atan2).
//
// The walk is written as RESUMABLE STEPS over an explicit per-lane state, so that the trace kernel's
// intra-wave scheduler can run "one inner-node step" or "one primitive test" for whichever lanes
// are waiting on it (trace_kernels.hpp), while closest_solid() below simply drives the same steps to
// completion (probe kernel, host build).  The per-lane sequence of steps — hence every result and
// counter — is the same whoever drives it.
constexpr int32_t TRAV_DONE = INT32_MIN + 2; // Trav::node once the stack has run empty

template <typename R> struct Trav {
    Ray<R> ray;       // the ray in the current space (world, or an instance's object space)
    SlabRay<R> sr;    // 1 / ray.d (and the f32 FMA-form products)
    uint32_t near_off[3]; // where the near planes of axis a lie in a node record, in the stack's addressing (Stack::plane_off)
    R closest;
    HitRef best;
    int32_t node;     // >= 0: inner node to visit; < 0: leaf bits (or CHILD_EMPTY); TRAV_DONE: finished
    int32_t sp;       // stack entries in use
    int32_t cur_inst; // instance being walked, or -1
    uint32_t leaf_k;  // next record of the current leaf
    bool found;
};


template <typename R, typename Stack> RT_HD void trav_set_ray(Trav<R>& tr, const Ray<R>& ray, const Stack& stack) {
    tr.ray = ray;
    tr.sr = slab_ray<slab_form<Stack, R>()>(ray.o, ray.d);
#pragma unroll
    for (int a = 0; a < 3; ++a) tr.near_off[a] = stack.plane_off(near_piece(a, tr.sr));
}
// A ray NOTHING can cull: on every axis its plane distances are NaN (a NaN / infinite origin or direction, a zero direction, |o / d| beyond the
// floats) — maxNum / minNum drop them all, so even the inverted box of an UNUSED node slot "passes" (the node steps do not test the slot itself,
// RT_NODE_EMPTY_CHECK).  Such a walk would push three entries per node, past the bound the lowering sized the stacks by (FlatScene::stack_depth
// counts real children).  One axis with a finite, non-zero 1 / d and a finite o / d is enough for every unused slot to miss (entry +inf, exit -inf).
// A ray without one is not walked at all: no hit, closest = NaN — which path_shade turns into a NaN path value, the NaN pixel the reference
// renders from such a ray (main.rs:219-225 then writes 0).  Checked once per world.hit(), not per node.
template <typename R> RT_HD bool slab_ray_can_be_culled(const SlabRay<R>& sr, V3<R> o) {
    const R oo[3] = {o.x, o.y, o.z};
    bool any = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float inv = slab_inv_of(sr, a);
        float oi;
        if constexpr (sizeof(R) == 8) oi = sr.oinv[a]; else oi = float(oo[a]) * inv;
        any = any || (rt_fabs(oi) < __builtin_huge_valf() && rt_fabs(inv) > 0.f && rt_fabs(inv) < __builtin_huge_valf());
    }
    return any;
}
// trav_begin() in two halves — the walk's cursor, and what the ray contributes to its slab tests — for the kernel that SUSPENDS walks across its
// shade phases (trace_kernels.hpp trace_kernel_plain, RT_ASYNC_SHADE): only the cursor lives through a shade phase, the slab constants are made
// again from the path's ray when the walk goes on.
template <typename R> RT_HD void trav_init(Trav<R>& tr, const SceneView<R>& sc) {
    tr.closest = Lim<R>::max(); // world.hit(ray, t_min, f64::MAX) — main.rs:33
    tr.best.prim = make_ref(PRIM_NONE, 0);
    tr.best.inst = -1;
    tr.best.aux = 0;
    tr.node = sc.top_root;
    tr.sp = 0;
    tr.cur_inst = -1;
    tr.leaf_k = 0;
    tr.found = false;
}
template <typename R> RT_HD void trav_reject_unwalkable(Trav<R>& tr, const Ray<R>& wray) {
    if (!slab_ray_can_be_culled(tr.sr, wray.o)) { tr.node = TRAV_DONE; tr.closest = Lim<R>::inf() - Lim<R>::inf(); }
}
template <typename R, typename Stack> RT_HD void trav_begin(Trav<R>& tr, const SceneView<R>& sc, const Ray<R>& wray, const Stack& stack) {
    trav_set_ray(tr, wray, stack);
    trav_init(tr, sc);
    trav_reject_unwalkable(tr, wray);
}

// Take the next pending subtree off the stack (leaving an instance when its sentinel comes up).
template <bool NO_INST = false, typename R, typename Stack> RT_HD void trav_pop(Trav<R>& tr, const Ray<R>& wray, Stack& stack) {
    tr.leaf_k = 0;
    if (tr.sp == 0) { tr.node = TRAV_DONE; return; }
    int32_t node = stack.get(--tr.sp);
    if (!NO_INST && node == STACK_SENTINEL) { // (a scene without instances never pushes one)
        trav_set_ray(tr, wray, stack);
        tr.cur_inst = -1;
        if (tr.sp == 0) { tr.node = TRAV_DONE; return; }
        node = stack.get(--tr.sp);
    }
    tr.node = node;
}

// One inner node (tr.node >= 0): test the (up to) four child boxes, descend into the nearest hit child, push the others
// farthest first.  The hit children are ordered by a 5-comparator sorting network on (key, child) pairs: key = the bits
// of the (positive) entry distance — they order like the values — 0xFFFFFFFF for a miss; equal keys keep slot order
// (the order of visits never changes a result).
RT_HD void pair_swap(uint32_t& ka, int32_t& ca, uint32_t& kb, int32_t& cb) { // (key, child) pairs: smaller key first
    const bool sw = kb < ka;
    const uint32_t k0 = sw ? kb : ka, k1 = sw ? ka : kb;
    const int32_t c0 = sw ? cb : ca, c1 = sw ? ca : cb;
    ka = k0; kb = k1; ca = c0; cb = c1;
}
// ---- helpers of the step over QUANTISED records (rt_types.hpp Bvh4QNode; trav_node_step4q below) ----
RT_HD float bits_float(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
#endif
}
// The second half of a 4-wide node step: order the (key, child) pairs (key = the bits of the entry distance, MISS_KEY for a miss),
// descend into the nearest hit child, push the others farthest first.
constexpr uint32_t MISS_KEY = 0xFFFFFFFFu;
template <bool NO_INST = false, typename R, typename Stack>
RT_HD void trav_descend_sorted4(Trav<R>& tr, const Ray<R>& wray, Stack& stack, uint32_t* k, int32_t* ch) {
    constexpr uint32_t MISS = MISS_KEY;
    pair_swap(k[0], ch[0], k[1], ch[1]); pair_swap(k[2], ch[2], k[3], ch[3]); pair_swap(k[0], ch[0], k[2], ch[2]);
    pair_swap(k[1], ch[1], k[3], ch[3]); pair_swap(k[1], ch[1], k[2], ch[2]);
    if (k[0] == MISS) { trav_pop<NO_INST>(tr, wray, stack); return; }
    // the other hit children become pending, farthest first
    const int32_t n_push = int32_t(k[1] != MISS) + int32_t(k[2] != MISS) + int32_t(k[3] != MISS);
    const int32_t sp = tr.sp;
    if (stack.room_for_three(sp)) {
        // the usual case, branch-free: three stores inside the LDS part of the stack, a store for a child that is not
        // pending goes to the lane's spare slot
        stack.set_fast(k[3] != MISS ? sp : Stack::SPARE, ch[3]);
        stack.set_fast(k[2] != MISS ? sp + n_push - 2 : Stack::SPARE, ch[2]);
        stack.set_fast(k[1] != MISS ? sp + n_push - 1 : Stack::SPARE, ch[1]);
    } else {
        if (k[3] != MISS) stack.set(sp, ch[3]);
        if (k[2] != MISS) stack.set(sp + n_push - 2, ch[2]);
        if (k[1] != MISS) stack.set(sp + n_push - 1, ch[1]);
    }
    tr.sp = sp + n_push;
    tr.node = ch[0];
}
template <typename R, typename Stack, typename Cnt>
RT_HD void trav_node_step4(Trav<R>& tr, const SceneView<R>& sc, const Ray<R>& wray, R t_min, Stack& stack, Cnt& cnt) {
    Planes4 nd;
    stack.fetch(sc, tr.node, tr.near_off, nd); // from global memory, or from LDS when the kernel keeps the tree there
    cnt.node();
    float lo_t, hi_t;
    slab_range(t_min, tr.closest, lo_t, hi_t);
    constexpr uint32_t MISS = 0xFFFFFFFFu;
    uint32_t k[4];
    int32_t ch[4];
    float e[4];
    bool h[4];
    slab_hit4<slab_form<Stack, R>()>(nd, (Cnt::NO_INST ? wray : tr.ray).o, tr.sr, lo_t, hi_t, e, h);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        ch[c] = nd.child[c];
        // (entry distances are positive: their bit patterns order like the values.  An unused slot needs no test of its own: its box is
        // inverted, lo = +inf / hi = -inf (rt_types.hpp Bvh4Node), so its entry is +inf and its exit -inf whatever the direction's signs;
        // were every one of its plane distances NaN — |o / d| overflowing — it would be descended into and popped at once, trav_leaf_step)
#if RT_NODE_EMPTY_CHECK
        k[c] = (h[c] && ch[c] != CHILD_EMPTY) ? float_bits(e[c]) : MISS;
#else
        k[c] = h[c] ? float_bits(e[c]) : MISS;
#endif
    }
    trav_descend_sorted4<Cnt::NO_INST>(tr, wray, stack, k, ch);
}
// ---- the same step over a 4-wide QUANTISED record (rt_types.hpp Bvh4QNode: four 16-byte pieces instead of the f32 record's seven).
// Plane distances: with D = org - o, A = D inv, S = step inv (step a power of two: exact), a child's plane at org + q step lies at
//   t = q S + A,                          one v_cvt_f32_ubyte + one fma per plane.
// Conservative as the f64 kernels' f32-record test is (a box the exact test passes always passes): D is rounded once (|e| <= 2^-24;
// the f64 kernels subtract in double first, so no |o| 2^-24 term appears), inv is the walk's 1-ulp reciprocal of the rounded
// direction (|e| <= 1.8e-7), A's product and the fma round once each:
//   |t32 - t| <= 3.6e-7 |t| + 4.8e-7 |q S|  <=  3.6e-7 |t| + 1.23e-4 max_a |S_a|      (q <= 255),
// so the entry moves down and the exit up by 4e-7 |t| + K, K = 1.5e-4 max_a |S_a| per record — a fma each: t (1 -+ 4e-7) -+ K is
// that widening for t >= 0, and a negative entry cannot set the walk's entry (t_min >= 0 does: render_api.cpp validate), a negative
// exit is a box behind the origin.  An axis the ray is parallel to gets NaN distances, which maxNum / minNum drop: it never culls.
// The children are ordered exactly, by entry distance, as in trav_node_step4.
template <typename R, typename Stack, typename Cnt>
RT_HD void trav_node_step4q(Trav<R>& tr, const SceneView<R>& sc, const Ray<R>& wray, R t_min, Stack& stack, Cnt& cnt) {
    uint32_t w[16]; // pieces 0-3 of the record
    stack.fetch4q(sc, tr.node, w);
#if RT_PIN_CHILD_PIECE && defined(__HIP_DEVICE_COMPILE__)
    // the children's piece is READ with the others: nothing below needs it before the hit tests are done, and left alone the compiler sinks its
    // read below them — a second dependent round trip to memory per visit (spheres_1m f64 280 -> 248 Msamples/s)
    asm volatile("" : "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15]));
#endif
    cnt.node();
    float lo_t, hi_t;
    slab_range(t_min, tr.closest, lo_t, hi_t);
    const Ray<R>& cray = Cnt::NO_INST ? wray : tr.ray; // the ray in the current frame
    const R oo[3] = {cray.o.x, cray.o.y, cray.o.z};
    float A[3], S[3];
    uint32_t near_w[3], far_w[3]; // the four children's near / far plane bytes of axis a
    float smax = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float inv = slab_inv_of(tr.sr, a);
        // (an axis the ray is parallel to must not cull: NaN, not infinity.  The f64 walk's constants are made that way once per walk — slab_ray —;
        // the f32 walk's are the bare reciprocals)
        if constexpr (sizeof(R) == 4)
            if (!(rt_fabs(inv) < __builtin_huge_valf())) inv = __builtin_nanf("");
        const bool neg = inv < 0.f;
        const float D = float(R(bits_float(w[a])) - oo[a]);
        A[a] = D * inv;
        S[a] = bits_float(((w[3] >> (8 * a)) & 0xFFu) << 23) * inv;
        smax = rt_max(smax, rt_fabs(S[a]));
        const uint32_t l = w[4 + 2 * a], h = w[5 + 2 * a];
        near_w[a] = neg ? h : l;
        far_w[a] = neg ? l : h;
    }
    const float K = smax * 1.5e-4f;
    uint32_t k[4];
    int32_t ch[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        ch[c] = int32_t(w[12 + c]);
        const int sh = 8 * c;
        const float nx = __builtin_fmaf(float((near_w[0] >> sh) & 0xFFu), S[0], A[0]), fx = __builtin_fmaf(float((far_w[0] >> sh) & 0xFFu), S[0], A[0]);
        const float ny = __builtin_fmaf(float((near_w[1] >> sh) & 0xFFu), S[1], A[1]), fy = __builtin_fmaf(float((far_w[1] >> sh) & 0xFFu), S[1], A[1]);
        const float nz = __builtin_fmaf(float((near_w[2] >> sh) & 0xFFu), S[2], A[2]), fz = __builtin_fmaf(float((far_w[2] >> sh) & 0xFFu), S[2], A[2]);
        const float tn = rt_max(nz, rt_max(ny, nx)), tf = rt_min(fz, rt_min(fy, fx));
        const float n = __builtin_fmaf(tn, 1.f - 4e-7f, -K), f = __builtin_fmaf(tf, 1.f + 4e-7f, K);
        const float lo = rt_max(n, lo_t), hi = rt_min(f, hi_t); // maxNum / minNum: a NaN drops out
        // (an unused slot: q_lo = 255, q_hi = 0 — its entry lies 255 max |S| beyond its exit, the widening moves them by 3e-4 max |S| + 8e-7 |t|)
#if RT_NODE_EMPTY_CHECK || !RT_PIN_CHILD_PIECE
        k[c] = (!(hi < lo) && ch[c] != CHILD_EMPTY) ? float_bits(lo) : MISS_KEY;
#else
        k[c] = !(hi < lo) ? float_bits(lo) : MISS_KEY;
#endif
    }
    trav_descend_sorted4<Cnt::NO_INST>(tr, wray, stack, k, ch);
}

template <typename R, typename Stack, typename Cnt>
RT_HD void trav_node_step(Trav<R>& tr, const SceneView<R>& sc, const Ray<R>& wray, R t_min, Stack& stack, Cnt& cnt) {
    if constexpr (Stack::WIDE == NODES_Q8X4) trav_node_step4q(tr, sc, wray, t_min, stack, cnt); // quantised records (the decoupled kernels)
    else trav_node_step4(tr, sc, wray, t_min, stack, cnt);
}

// One step at a leaf (tr.node < 0, not TRAV_DONE): enter an instance, or test ONE primitive record.  WHOLE_LEAF tests
// all (<= 4) records of the leaf in one step instead — the same tests in the same order; measured 10-14 % SLOWER in
// the lockstep kernel (lanes with a one-record leaf wait instead of going on with node steps), so nothing uses it.
// (`ray`: the ray in the frame of the record — tr.ray, the world ray itself where the walk never changes frames, or a single wrapped record's
// object-space ray; `inst`: the instance that frame belongs to, or -1)
// Leaves of kind PRIM_SPHERE_WC (the lowering RTTNW_F64_STRICT renders; the host build of the core knows them too): the world-space copy of a sphere
// that lives under Translate / YRotate wrappers was found through its world-space BOX, which only culls; the test itself is the reference's —
// the ray taken through the group's wrappers (Translate::hit hittable.rs:599-606, YRotate::hit :686-699: the walk's own to_object code) and
// Sphere::hit (:87-109) on the object-space record, so t is the reference's bit for bit.  The copy's material slot leads to both (MAT_HOME_*).
#if defined(RT_STRICT_F64) || !defined(__HIP_DEVICE_COMPILE__)
#define RT_HAS_SPHERE_WC 1
#else
#define RT_HAS_SPHERE_WC 0
#endif
template <bool G, typename R> RT_HD bool sphere_wc_t(const SceneView<R>& sc, uint32_t idx, const Ray<R>& wray, R t_min, R t_max, R& t) {
    const int32_t home = sc.sphere_mat[idx];
    const InstanceRec<R>& in = sc.insts[(home >> MAT_HOME_INST_SHIFT) & MAT_HOME_INST_MAX];
    const SphereRec<R> s = sc.spheres[uint32_t(home) & MAT_HOME_SPHERE_MASK];
    const Ray<R> obj = to_object<G>(in, wray);
    return sphere_t(V3<R>(s.cx, s.cy, s.cz), s.r, obj, t_min, t_max, t);
}
// A record passed its test at t: it becomes the incumbent — unless it TIES the incumbent exactly and comes earlier in list order (hittable.rs:157-159)
template <typename R> RT_HD void trav_accept(Trav<R>& tr, const SceneView<R>& sc, uint32_t kind, uint32_t idx, R t, int aux, int32_t inst) {
    const bool loses_tie = tr.found && t == tr.closest &&
                           prim_seq(sc, kind, idx) < prim_seq(sc, ref_kind(tr.best.prim), ref_index(tr.best.prim));
    if (!loses_tie) {
        tr.closest = t;
        tr.best.prim = make_ref(kind, idx);
        tr.best.inst = inst;
        tr.best.aux = aux;
        tr.found = true;
    }
}
// FRAME_RAY: `ray` is the ray of the walk's current frame — the one tr.sr was made from
template <bool G = false, bool NO_TIME = false, bool FRAME_RAY = false, typename R> RT_HD void trav_test_record(Trav<R>& tr, const SceneView<R>& sc, uint32_t kind, uint32_t idx, R t_min, const Ray<R>& ray, int32_t inst) {
    R t;
    int aux = 0;
    bool hit;
#if RT_HAS_SPHERE_WC
    if (kind == PRIM_SPHERE_WC) { // (only ever in the top tree: `ray` is the world ray)
        hit = sphere_wc_t<G>(sc, idx, ray, t_min, tr.closest, t);
        kind = PRIM_SPHERE; // the hit reference names the copy's record: make_record finds the object-space one from it
    } else
#endif
    hit = prim_t<NO_TIME, FRAME_RAY>(sc, kind, idx, ray, t_min, tr.closest, t, aux, tr.sr);
    if (hit) trav_accept(tr, sc, kind, idx, t, aux, inst);
}
template <bool WHOLE_LEAF = false, typename R, typename Stack, typename Cnt>
RT_HD void trav_leaf_step(Trav<R>& tr, const SceneView<R>& sc, const Ray<R>& wray, R t_min, Stack& stack, Cnt& cnt) {
    constexpr bool NI = Cnt::NO_INST;
    // (the fast cube test where the walk's plane distances are to hand and cubes are what a leaf step meets: not in the decoupled kernel — big trees of
    // spheres, whose LEAN flavour is held to 128 registers: spheres_1m strict 484 -> 479 with the code aboard)
    constexpr bool FAST_CUBES = Stack::WIDE != NODES_Q8X4;
    if (tr.node == CHILD_EMPTY) { trav_pop<NI>(tr, wray, stack); return; }
    const uint32_t kind = leaf_kind(tr.node), count = leaf_count(tr.node), first = leaf_first(tr.node);
    if (!Cnt::NO_INST_LEAF && kind == PRIM_INSTANCE) { // Translate / YRotate wrappers: continue in object space (hittable.rs:599-606,686-699)
        cnt.prim();
        const InstanceRec<R>& in_rec = sc.insts[first];
        const InstanceHead<R> head = head_of(in_rec); // one by-value copy serves the transform below
        const int32_t single_leaf = head.single_leaf, inst_root = head.root;
        auto object_ray = [&]() -> Ray<R> {
            if (!Cnt::GENERAL || head.n_ops <= FAST_INSTANCE_OPS) return to_object_fast(head, wray, head.n_ops);
            return to_object<Cnt::GENERAL>(in_rec, wray);
        };
        if (single_leaf != 0) { // one wrapped record: test it here in object space — no sentinel, no one-node tree to walk, the walk stays in its frame
            const Ray<R> obj = object_ray();
            cnt.prim();
            trav_test_record<Cnt::GENERAL, Cnt::NO_TIME>(tr, sc, leaf_kind(single_leaf), leaf_first(single_leaf), t_min, obj, int32_t(first));
            trav_pop<NI>(tr, wray, stack);
            return;
        }
        if constexpr (!NI) { // (a scene whose walk never changes frames has no instance with a tree: FlatScene::walk_changes_frames)
            stack.set(tr.sp++, STACK_SENTINEL);
            trav_set_ray(tr, object_ray(), stack);
            tr.cur_inst = int32_t(first);
            tr.node = inst_root;
        }
        return;
    }
    if constexpr (WHOLE_LEAF) {
        for (uint32_t k = 0; k < count; ++k) {
            cnt.prim();
            trav_test_record<Cnt::GENERAL, Cnt::NO_TIME, FAST_CUBES>(tr, sc, kind, first + k, t_min, NI ? wray : tr.ray, NI ? -1 : tr.cur_inst);
        }
        trav_pop<NI>(tr, wray, stack);
    } else {
        cnt.prim();
        trav_test_record<Cnt::GENERAL, Cnt::NO_TIME, FAST_CUBES>(tr, sc, kind, first + tr.leaf_k, t_min, NI ? wray : tr.ray, NI ? -1 : tr.cur_inst);
        if (++tr.leaf_k >= count) trav_pop<NI>(tr, wray, stack);
    }
}

template <int NODE_STEPS = RT_NODE_STEPS, typename R, typename Stack, typename Cnt>
RT_HD bool closest_solid(const SceneView<R>& sc, const Ray<R>& wray, R t_min, R& closest, HitRef& best, Stack& stack, Cnt& cnt) {
    Trav<R> tr;
    trav_begin(tr, sc, wray, stack);
    while (tr.node != TRAV_DONE) {
        // one trip: two node steps, then a leaf step for the lanes at a leaf by then.  A lane whose node step arrives at
        // a leaf tests it in the same trip, together with the lanes that were already waiting at theirs; the second node
        // step halves the trips of node-heavy walks and with them the executions of the (wide, poorly filled) leaf
        // code.  Measured, node steps per trip 1 / 2 / 3 / 4: final_scene 1288 / 1368 / 1357 / 1356 Msamples/s,
        // cornell_box 1707 / 1718 / 1665 / 1691.
        // (re-measured with the spheres of transformed groups in the top tree, node steps per trip 1 / 2 / 3: final_scene f32
        // 1605 / 1632 / 1562, f64 - / 1058 / 1037; cornell_box f32 1893 / 2040 / 2066, f64 - / 1359 / 1370)
        // (round 4, per tree size: the lane-owns-path kernel takes THREE for a top tree of <= 16 nodes — cornell_box, whose walks are mostly
        // entered instances: f64 1749 -> 1804, f32 2305 -> 2347; final_scene with three: 1429 -> 1401 — render_tiles.hpp picks the instantiation)
        constexpr int node_steps = NODE_STEPS;
#pragma unroll
        for (int k = 0; k < node_steps; ++k)
            if (tr.node >= 0) trav_node_step(tr, sc, wray, t_min, stack, cnt);
        if (tr.node < 0 && tr.node != TRAV_DONE) trav_leaf_step(tr, sc, wray, t_min, stack, cnt);
    }
    closest = tr.closest;
    best = tr.best;
    return tr.found;
}

// ---------------------------------------------------------------- hit record of the winner
template <typename R> struct HitRecord { // hittable.rs:15-27
    R t;
    V3<R> p, normal;
    R u, v;
    int32_t mat;
    bool front_face;
};
template <typename R> RT_HD void face_normal(V3<R> dir, V3<R> outward, V3<R>& normal, bool& front) { // hittable.rs:30-44
    front = dot(dir, outward) < R(0);
    normal = front ? outward : -outward;
}
template <typename R> RT_HD void sphere_uv(V3<R> p, R& u, R& v) { // hittable.rs:77-83
    const R pi = R(3.14159265358979323846264338327950288);
    R theta = rt_acos(-p.y);
    R phi = rt_atan2(-p.z, p.x) + pi;
    u = div_by(phi, recip_of(R(2) * pi)); // (constant divisors: their reciprocals fold)
    v = div_by(theta, recip_of(pi));
}

// The reference computes (u, v) for every hit (Sphere::uv's acos + atan2, the rectangles' two divisions; Q11), but only
// an image texture — possibly under a checker — ever reads them: they are computed only then.  Same results.
RT_HD bool uv_is_read(int32_t mat_ref) { return (mat_ref & MAT_UV_FLAG) != 0; } // decided by the lowering, see MAT_UV_FLAG

template <bool G, bool NO_TIME = false, typename R> // (NO_TIME = a LEAN scene, rt_core.hpp SHAPES_*_NT: no texture reads (u, v) either)
RT_HD void make_record(const SceneView<R>& sc, const Ray<R>& wray, HitRef ref, R t, uint32_t quirks, HitRecord<R>& rec) {
    const uint32_t kind = ref_kind(ref.prim);
    uint32_t idx = ref_index(ref.prim);
    // (sc.sphere_mat == nullptr: sphere i's material slot holds i — FlatScene::sphere_mat_is_index, big clouds only: no slot is read, and none is a copy's)
    if (kind == PRIM_SPHERE && ref.inst < 0 && sc.sphere_mat) { // the world-space copy of a transformed group's sphere: its record is made in
        const int32_t home = sc.sphere_mat[idx]; // object space through the group's chain, like the reference's (scene_lower.cpp)
        if (home & MAT_HOME_FLAG) { idx = uint32_t(home) & MAT_HOME_SPHERE_MASK; ref.inst = (home >> MAT_HOME_INST_SHIFT) & MAT_HOME_INST_MAX; }
    }
    Ray<R> ray = wray;
    if (ref.inst >= 0) ray = to_object<G>(sc.insts[ref.inst], wray);
    rec.t = t;
    V3<R> outward;
    if (kind == PRIM_SPHERE) { // hittable.rs:109-113
        SphereRec<R> s = sc.spheres[idx];
        V3<R> c(s.cx, s.cy, s.cz);
        rec.p = ray.at(t);
        outward = (rec.p - c) / s.r;
        const int32_t mref = sc.sphere_mat ? sc.sphere_mat[idx] : int32_t(idx);
        rec.mat = mref & MAT_INDEX_MASK;
        rec.u = R(0); rec.v = R(0);
        // (f64: deferring these ~250 instructions to the image texture's texel choice, made in f32 wherever f32 is certain of the
        // texel and in f64 otherwise, was built, bit-identical and 1 % SLOWER — profiles/r03/README.md)
        if (!NO_TIME && uv_is_read(mref)) sphere_uv(outward, rec.u, rec.v);
    } else if (!NO_TIME && kind == PRIM_MOVING_SPHERE) { // hittable.rs:217-221
        const MovingSphereRec<R> m = sc.moving[idx];
        rec.p = ray.at(t);
        outward = (rec.p - moving_center(m, ray.time)) / m.r;
        rec.u = R(0); rec.v = R(0);
        rec.mat = m.mat & MAT_INDEX_MASK;
    } else if (kind == PRIM_RECT) { // hittable.rs:515-519
        const RectRec<R> r = sc.rects[idx];
        rec.u = R(0); rec.v = R(0);
        if (!NO_TIME && uv_is_read(r.mat)) {
            R a, b;
            rect_ab(r.plane, ray, t, a, b);
            rec.u = rt_div(a - r.a0, r.a1 - r.a0);
            rec.v = rt_div(b - r.b0, r.b1 - r.b0);
        }
        outward = V3<R>(r.plane == 2 ? R(1) : R(0), r.plane == 1 ? R(1) : R(0), r.plane == 0 ? R(1) : R(0));
        rec.p = ray.at(t);
        rec.mat = r.mat & MAT_INDEX_MASK;
    } else { // PRIM_BOX: the winning face's rectangle record
        const BoxRec<R> bx = sc.boxes[idx];
        const int plane = ref.aux >> 1;
        rec.u = R(0); rec.v = R(0);
        if (!NO_TIME && uv_is_read(bx.mat)) {
            R a, b;
            rect_ab(plane, ray, t, a, b);
            R a0 = plane == 2 ? bx.mn[1] : bx.mn[0], a1 = plane == 2 ? bx.mx[1] : bx.mx[0];
            R b0 = plane == 0 ? bx.mn[1] : bx.mn[2], b1 = plane == 0 ? bx.mx[1] : bx.mx[2];
            rec.u = rt_div(a - a0, a1 - a0);
            rec.v = rt_div(b - b0, b1 - b0);
        }
        outward = V3<R>(plane == 2 ? R(1) : R(0), plane == 1 ? R(1) : R(0), plane == 0 ? R(1) : R(0));
        rec.p = ray.at(t);
        rec.mat = bx.mat & MAT_INDEX_MASK;
    }
    face_normal(ray.d, outward, rec.normal, rec.front_face);

    if (ref.inst >= 0) {
        const InstanceRec<R>& in = sc.insts[ref.inst];
        unwind_record<G>(in, head_of(in).n_ops, wray.d, quirks, rec.p, rec.normal, rec.front_face);
    }
}

// ---------------------------------------------------------------- world.hit (main.rs:33)
// Solids through the BVH, then the constant media in creation order with the shrinking `closest`
// they would see in the reference's world List (hittable.rs:740-796, Q13).
// world_hit_finish(): everything after the BVH walk — media, then the winner's hit record.
// hittable.rs:751: `record1.t + 0.0001`.  The absolute epsilon must stay above the spacing of R at t1, or the
// second query finds the SAME root again (f32: ulp(5000) = 4.9e-4 swallows it and the medium is skipped); in
// f64 the guard never binds below |t| ~ 1e11, so the reference value is used unchanged.
template <typename R> RT_HD R medium_sep(R t1) {
    return rt_max(R(0.0001), rt_fabs(t1) * (sizeof(R) == 4 ? R(4.8e-7) : R(8.9e-16)));
}

template <typename R, typename Cnt>
RT_HD bool world_hit_finish(const SceneView<R>& sc, const Ray<R>& ray, R t_min, uint64_t key, uint32_t bounce, uint32_t quirks,
                            bool found, R closest, HitRef best, HitRecord<R>& rec, Cnt& cnt) {
    int32_t medium = -1;
    const int32_t n_media = Cnt::LEAN ? 0 : sc.n_media; // (a LEAN scene has none: the loop and its log are compiled out)
    const R world_length = n_media > 0 ? magnitude(ray.d) : R(0); // |direction|: once for all media (hittable.rs:760)
    for (int32_t m = 0; m < n_media; ++m) {
        const MediumRec<R> md = sc.media[m];
        Ray<R> bray = ray;
        if (md.inst >= 0) bray = to_object<Cnt::GENERAL>(sc.insts[md.inst], ray);
        const int32_t ref0 = md.ref0;
        R t1, t2;
        cnt.prim(); // first boundary query; the second is counted once the first has hit, as the reference would call it
        if ((!Cnt::GENERAL || md.b_count == 1) && ref_kind(ref0) == PRIM_SPHERE) {
            // boundary.hit(ray, -inf, +inf) then boundary.hit(ray, t1 + 0.0001, +inf) (hittable.rs:745-751) are two
            // evaluations of ONE quadratic: same discriminant, near root first, far root if the near one is out of
            // range.  Evaluated once here, with Sphere::hit's range tests kept literally (NaN behaviour included).
            const SphereRec<R> sp = sc.spheres[ref_index(ref0)];
            const V3<R> oc = bray.o - V3<R>(sp.cx, sp.cy, sp.cz);
            const R a = dot(bray.d, bray.d), half_b = dot(oc, bray.d);
            const R disc = sphere_discriminant(oc, bray.d, a, half_b, sp.r);
            if (disc < R(0)) continue;
            const R sqrtd = rt_sqrt(disc);
            R near_root, far_root;
            rt_div2(-half_b - sqrtd, -half_b + sqrtd, a, near_root, far_root);
            t1 = near_root;
            if (t1 < -Lim<R>::inf() || Lim<R>::inf() < t1) {
                t1 = far_root;
                if (t1 < -Lim<R>::inf() || Lim<R>::inf() < t1) continue;
            }
            cnt.prim();
            const R lo2 = t1 + medium_sep(t1);
            t2 = near_root;
            if (t2 < lo2 || Lim<R>::inf() < t2) {
                t2 = far_root;
                if (t2 < lo2 || Lim<R>::inf() < t2) continue;
            }
        } else if constexpr (!Cnt::GENERAL) { // one cube (smoke_cornell_box): the two queries of hittable.rs:745-751
            int aux;
            if (!prim_t(sc, ref_kind(ref0), ref_index(ref0), bray, -Lim<R>::inf(), Lim<R>::inf(), t1, aux)) continue;
            cnt.prim();
            if (!prim_t(sc, ref_kind(ref0), ref_index(ref0), bray, t1 + medium_sep(t1), Lim<R>::inf(), t2, aux)) continue;
        } else {
            // any boundary: a cube, or a List / BvhTree of spheres and cubes (ConstantMedium takes any Hittable and treats it
            // as convex, hittable.rs:731,739): List::hit over the members with a shrinking t_max (hittable.rs:153-163), twice
            bool any = false;
            R cur = Lim<R>::inf();
            for (int32_t k = 0; k < md.b_count; ++k) {
                const int32_t ref = sc.medium_refs[md.b_first + k];
                R t; int aux;
                if (k) cnt.prim();
                if (prim_t(sc, ref_kind(ref), ref_index(ref), bray, -Lim<R>::inf(), cur, t, aux)) { cur = t; any = true; }
            }
            if (!any) continue;
            t1 = cur;
            any = false;
            cur = Lim<R>::inf();
            const R lo2 = t1 + medium_sep(t1);
            for (int32_t k = 0; k < md.b_count; ++k) {
                const int32_t ref = sc.medium_refs[md.b_first + k];
                R t; int aux;
                cnt.prim();
                if (prim_t(sc, ref_kind(ref), ref_index(ref), bray, lo2, cur, t, aux)) { cur = t; any = true; }
            }
            if (!any) continue;
            t2 = cur;
        }
        t1 = rt_max(t1, t_min);
        t2 = rt_min(t2, closest);
        if (t1 >= t2) continue; // before any draw
        t1 = rt_max(t1, R(0));
        // the ray as the medium itself sees it: inside a transformed group, the group's object-space ray
        R ray_length = world_length;
        if constexpr (Cnt::GENERAL)
            if (md.n_outer > 0) ray_length = magnitude(dir_after<true>(sc.insts[md.inst], ray.d, md.n_outer - 1));
        R distance_inside = (t2 - t1) * ray_length;
        R hit_distance = md.neg_inv_density * rt_log(uniform01_log<R>(key, rng_ctr(bounce + 1, SLOT_MEDIUM + uint32_t(m))));
        if (hit_distance > distance_inside) continue;
        closest = t1 + rt_div(hit_distance, ray_length);
        medium = m;
        found = true;
    }
    if (!found) return false;
    if (medium >= 0) { // hittable.rs:770-789
        const MediumRec<R> md = sc.media[medium];
        rec.t = closest;
        rec.normal = V3<R>(R(1), R(0), R(0));
        rec.front_face = true;
        rec.u = R(0); rec.v = R(0);
        rec.mat = md.mat;
        rec.p = ray.at(closest);
        if constexpr (Cnt::GENERAL)
            if (md.n_outer > 0) { // a medium inside a transformed group: its record goes through the group's wrappers like any other
                const InstanceRec<R>& in = sc.insts[md.inst];
                rec.p = to_object_n<true>(in, ray, md.n_outer).at(closest);
                unwind_record<true>(in, md.n_outer, ray.d, quirks, rec.p, rec.normal, rec.front_face);
            }
    } else {
        make_record<Cnt::GENERAL, Cnt::NO_TIME>(sc, ray, best, closest, quirks, rec);
    }
    return true;
}

template <typename R, typename Stack, typename Cnt>
RT_HD bool world_hit(const SceneView<R>& sc, const Ray<R>& ray, R t_min, uint64_t key, uint32_t bounce, uint32_t quirks,
                     HitRecord<R>& rec, Stack& stack, Cnt& cnt) {
    cnt.ray();
    R closest;
    HitRef best;
    const bool found = closest_solid(sc, ray, t_min, closest, best, stack, cnt);
    return world_hit_finish(sc, ray, t_min, key, bounce, quirks, found, closest, best, rec, cnt);
}

// ---------------------------------------------------------------- textures (texture.rs, noise.rs)
template <typename R> RT_HD R perlin_noise(const R* vec, const uint8_t* perm, V3<R> p) { // noise.rs:50-94
    R fx = rt_floor(p.x), fy = rt_floor(p.y), fz = rt_floor(p.z);
    R u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int32_t i = int32_t(fx), j = int32_t(fy), k = int32_t(fz);
    R uu = u * u * (R(3) - R(2) * u), vv = v * v * (R(3) - R(2) * v), ww = w * w * (R(3) - R(2) * w);
    R acc = 0;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
            for (int dk = 0; dk < 2; ++dk) {
                uint32_t h = uint32_t(perm[(i + di) & 255]) ^ uint32_t(perm[256 + ((j + dj) & 255)]) ^ uint32_t(perm[512 + ((k + dk) & 255)]);
                V3<R> c(vec + 3 * h);
                V3<R> weight(u - R(di), v - R(dj), w - R(dk));
                acc += (R(di) * uu + R(1 - di) * (R(1) - uu)) * (R(dj) * vv + R(1 - dj) * (R(1) - vv)) *
                       (R(dk) * ww + R(1 - dk) * (R(1) - ww)) * dot(c, weight);
            }
    return acc;
}
template <typename R> RT_HD R perlin_turbulence(const R* vec, const uint8_t* perm, V3<R> p, int depth) { // noise.rs:96-108
    R acc = 0, weight = 1;
    for (int o = 0; o < depth; ++o) {
        acc = acc + weight * perlin_noise(vec, perm, p);
        weight = weight * R(0.5);
        p = p * R(2);
    }
    return acc;
}

// The texel an image texture reads at (u, v) — texture.rs:78-101: clamp, flip v, `as u32` (NaN -> 0, saturating), clamp to the last one.
template <typename R> RT_HD void image_texel(const ImageRec& im, R u, R v, uint32_t& i, uint32_t& j) {
    if (u < R(0)) u = R(0);
    if (u > R(1)) u = R(1);
    if (v < R(0)) v = R(0);
    if (v > R(1)) v = R(1);
    v = R(1) - v;
    R fi = u * R(im.w), fj = v * R(im.h);
    i = (fi == fi && fi > R(0)) ? uint32_t(fi) : 0u;
    j = (fj == fj && fj > R(0)) ? uint32_t(fj) : 0u;
    if (i >= im.w) i = im.w - 1;
    if (j >= im.h) j = im.h - 1;
}
template <typename R, typename Cnt>
RT_HD V3<R> texture_value(const SceneView<R>& sc, int32_t tex, const HitRecord<R>& rec, Cnt& cnt) {
    const V3<R> p = rec.p;
    for (;;) {
        const TextureRec<R>& t = sc.texs[tex];
        if (t.type == TEX_CHECKER) { // texture.rs:20-29
            R sines = rt_sin(R(10) * p.x) * rt_sin(R(10) * p.y) * rt_sin(R(10) * p.z);
            tex = sines < R(0) ? t.a : t.b;
            continue;
        }
        if (t.type == TEX_SOLID) return V3<R>(t.color);
        if (t.type == TEX_NOISE) { // texture.rs:54-58
            const R* vec = sc.perlin_vec + size_t(t.a) * 768;
            const uint8_t* perm = sc.perlin_perm + size_t(t.a) * 768;
            R g = R(0.5) * (R(1) + rt_sin(t.scale * p.z + R(10) * perlin_turbulence(vec, perm, p, 7)));
            return V3<R>(g, g, g);
        }
        if (t.type == TEX_IMAGE) { // texture.rs:78-101
            const ImageRec im = sc.images[t.a];
            uint32_t i, j;
            image_texel(im, rec.u, rec.v, i, j);
            cnt.texel();
            uint32_t px = sc.texels[im.offset + size_t(j) * im.w + i];
            const R s = R(1) / R(255);
            return V3<R>(R(px & 255u) * s, R((px >> 8) & 255u) * s, R((px >> 16) & 255u) * s);
        }
        return V3<R>(R(0), R(1), R(1)); // TEX_CYAN — texture.rs:102-105
    }
}

template <typename R, typename Cnt>
RT_HD V3<R> material_color(const SceneView<R>& sc, const MaterialRec<R>& m, const HitRecord<R>& rec, Cnt& cnt) {
    if (m.tex < 0) return V3<R>(m.albedo);
    return texture_value(sc, m.tex, rec, cnt);
}

// ---------------------------------------------------------------- scatter (material.rs)
RT_HD float schlick_pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
RT_HD double schlick_pow5(double x) { double x2 = x * x; return x2 * x2 * x; }
template <typename R> RT_HD R schlick(R cosine, R ri) { // material.rs:173-176
    R r0 = rt_div(R(1) - ri, R(1) + ri);
    r0 = r0 * r0;
    return r0 + (R(1) - r0) * schlick_pow5(R(1) - cosine);
}

// Returns true when the path continues; `att` and `ray` (in/out) are then the attenuation and the
// scattered ray.  `emitted` is always set (material.rs:10-12,247-249).
template <typename R, typename Cnt>
RT_HD bool shade(const SceneView<R>& sc, const HitRecord<R>& rec, uint64_t key, uint32_t bounce, Ray<R>& ray, V3<R>& att,
                 V3<R>& emitted, Cnt& cnt) {
    const MaterialRec<R> m = sc.mats[rec.mat];
    emitted = V3<R>();
    // albedo / emission texture of the three textured kinds, evaluated in ONE place (Perlin is ~350 instructions)
    V3<R> colour;
    // (round 5: a noise texture evaluated by the WHOLE wave — lane l the corner l & 7 of octave l >> 3 of one requesting lane's point, sums in the
    // reference's order, bit-identical — was built and measured: final_scene f64 +0.6 %, strict -2.4 %, cornell_box, which has no noise, -12 % through
    // the registers the extra code takes in a kernel that spills: profiles/r05/README.md.  Not kept.)
    if (m.type == MAT_DIFFUSE_LIGHT || m.type == MAT_LAMBERTIAN || m.type == MAT_ISOTROPIC) {
        if constexpr (Cnt::LEAN) colour = V3<R>(m.albedo); // (every material of a LEAN scene is a solid colour: lowered into the material record)
        else colour = material_color(sc, m, rec, cnt);
    }
    if (m.type == MAT_DIFFUSE_LIGHT) { // material.rs:242-250
        emitted = colour;
        return false;
    }
    // The unit-ball draw of Lambertian / Isotropic / Metal (the same keyed draws whichever asks) and the unit direction of
    // Metal / Dielectric, each in ONE place: three inlined rejection loops in three divergent branches ran one after the
    // other, each to the iteration count of its unluckiest lane.
    V3<R> ball, ud;
    if (m.type == MAT_LAMBERTIAN || m.type == MAT_ISOTROPIC || m.type == MAT_METAL) ball = random_in_unit_space<R>(key, bounce);
    if (m.type == MAT_METAL || m.type == MAT_DIELECTRIC) ud = unit(ray.d);
    if (m.type == MAT_LAMBERTIAN) { // material.rs:89-100
        V3<R> target = rec.p + rec.normal + ball;
        ray.d = target - rec.p;
        ray.o = rec.p;
        att = colour;
        return true;
    }
    if (m.type == MAT_ISOTROPIC) { // material.rs:256-265
        ray.o = rec.p;
        ray.d = ball;
        att = colour;
        return true;
    }
    if (m.type == MAT_METAL) { // material.rs:134-149
        V3<R> reflected = reflect(ud, rec.normal);
        ray.o = rec.p;
        ray.d = reflected + m.param * ball;
        att = V3<R>(m.albedo);
        return dot(ray.d, rec.normal) > R(0);
    }
    // MAT_DIELECTRIC — material.rs:179-203
    att = V3<R>(R(1), R(1), R(1));
    R ratio = rec.front_face ? rt_rcp(m.param) : m.param;
    R cos_theta = rt_min(dot(-ud, rec.normal), R(1));
    R sin_theta = rt_sqrt(R(1) - cos_theta * cos_theta);
    bool cannot_refract = ratio * sin_theta > R(1);
    // the draw is keyed, so evaluating it unconditionally is equivalent to the short-circuit (Q6)
    bool refl = cannot_refract || schlick(cos_theta, ratio) > uniform01<R>(key, rng_ctr(bounce + 1, SLOT_DIELECTRIC));
    ray.o = rec.p;
    ray.d = refl ? reflect(ud, rec.normal) : refract(ud, rec.normal, ratio);
    return true;
}

// ---------------------------------------------------------------- one path, lane-resident state
template <typename R> struct PathState {
    Ray<R> ray;
    V3<R> throughput; // product of attenuations so far
    V3<R> radiance;   // the finished path's value (set by the path's LAST path_shade: see there)
    uint64_t key;
    uint32_t bounce;
};

// Start sample `s` of pixel (px, row) — main.rs:212-215.  row 0 is the TOP row (j = height-1).
template <typename R>
RT_HD void path_begin(PathState<R>& ps, const CameraRec<R>& cam, const RenderConsts& rc, uint32_t px, uint32_t row, uint32_t s) {
    const uint64_t pixel = uint64_t(row) * rc.width + px;
    ps.key = sample_key(rc.seed, pixel, uint64_t(s) + rc.sample_begin);
    const uint32_t j = rc.height - 1 - row;
    R u, v; // main.rs:213-214
#if defined(__HIP_DEVICE_COMPILE__) && RT_SHARED_RECIPROCALS
    constexpr bool by_reciprocal = sizeof(R) == 8; // the contracted f64 kernels: two multiplications for two IEEE divisions by render constants
#else
    constexpr bool by_reciprocal = false;
#endif
    if constexpr (by_reciprocal) {
        u = (R(px) + uniform01<R>(ps.key, rng_ctr(0, SLOT_JITTER_U))) * R(rc.inv_width);
        v = (R(j) + uniform01<R>(ps.key, rng_ctr(0, SLOT_JITTER_V))) * R(rc.inv_height);
    } else {
        u = rt_div(R(px) + uniform01<R>(ps.key, rng_ctr(0, SLOT_JITTER_U)), R(rc.width));
        v = rt_div(R(j) + uniform01<R>(ps.key, rng_ctr(0, SLOT_JITTER_V)), R(rc.height));
    }
    ps.ray = camera_ray(cam, u, v, ps.key, (rc.scene_flags & SCENE_NO_TIME) == 0u);
    ps.throughput = V3<R>(R(1), R(1), R(1));
    ps.bounce = 0;
}

// One world.hit + shade: the body of color() (main.rs:26-45) unrolled into a loop:
//   L = sum_k (prod_{i<k} att_i) * emitted_k  (+ throughput * background on a miss).
// Only the LAST term of that sum can be non-zero — nothing that emits scatters (material.rs:242-250: DiffuseLight::scatter is None; every other
// material emits black) — so the sum is not carried through the path: every call sets ps.radiance to ITS term, and what the last call of a
// path leaves there is the path's value, bit for bit what the carried sum was (0 + x = x; the earlier terms were throughput * 0).  Three
// reals fewer alive across the BVH walk (six registers in the f64 kernels) and out of the decoupled kernel's path-state pool.
// path_shade(): the part after the BVH walk (media, hit record, emitted/scatter, bookkeeping).
// Returns true while the path is alive.
// A path's value is a ROUNDED product: the kernels add it to their job's sum in code of their own (trace_kernels.hpp), and under -ffp-contract=fast
// the compiler would fuse `sum + throughput * emitted` into one fma in one kernel form and not in another — the forms' images would differ in the
// last place (tests/test_gpu_parity.py::test_kernel_forms_agree holds them bit-identical).  The empty asm makes the product a value of its own.
template <typename R> RT_HD void keep_rounded(V3<R>& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z));
#else
    (void)v;
#endif
}
template <typename R, typename Cnt>
RT_HD bool path_shade(PathState<R>& ps, const SceneView<R>& sc, const RenderConsts& rc, V3<R> background, R t_min, bool found,
                      R closest, HitRef best, Cnt& cnt) {
    HitRecord<R> rec;
    if (!world_hit_finish(sc, ps.ray, t_min, ps.key, ps.bounce, rc.quirks, found, closest, best, rec, cnt)) {
        ps.radiance = ps.throughput * background;
        if (closest != closest) ps.radiance = V3<R>(closest, closest, closest); // a ray that was not walked (trav_begin): a NaN path, like the reference's
        keep_rounded(ps.radiance);
        return false;
    }
    V3<R> att, emitted;
    bool cont = shade(sc, rec, ps.key, ps.bounce, ps.ray, att, emitted, cnt);
    ps.radiance = ps.throughput * emitted;
    keep_rounded(ps.radiance);
    if (!cont) return false;
    ps.throughput = ps.throughput * att;
    ps.bounce += 1;
    // color(depth == 0) returns black without tracing (main.rs:28-30): at most max_depth hits
    return ps.bounce < rc.max_depth;
}

template <int NODE_STEPS = RT_NODE_STEPS, typename R, typename Stack, typename Cnt>
RT_HD bool path_step(PathState<R>& ps, const SceneView<R>& sc, const RenderConsts& rc, V3<R> background, R t_min,
                     Stack& stack, Cnt& cnt) {
    cnt.ray();
    R closest;
    HitRef best;
    const bool found = closest_solid<NODE_STEPS>(sc, ps.ray, t_min, closest, best, stack, cnt);
    return path_shade(ps, sc, rc, background, t_min, found, closest, best, cnt);
}

// Gamma + quantise — main.rs:219-225 (`as u8` saturates, NaN -> 0)
RT_HD uint8_t quantise(double mean) {
    double x = sqrt(mean);
    if (x < 0.0) x = 0.0;
    if (x > 0.999) x = 0.999;
    x = x * 256.;
    if (!(x == x) || x <= 0.0) return 0;
    if (x >= 255.0) return 255;
    return uint8_t(x);
}
RT_HD uint8_t quantise(float mean) {
    float x = sqrtf(mean);
    if (x < 0.0f) x = 0.0f;
    if (x > 0.999f) x = 0.999f;
    x = x * 256.f;
    if (!(x == x) || x <= 0.0f) return 0;
    if (x >= 255.0f) return 255;
    return uint8_t(x);
}

// Tile partition (include/rttnw_hip.h rttnw_tile_layout): rows are rotated by ty so that a column of
// tiles is spread over all ranks.
RT_HD uint32_t tile_permuted(uint32_t tx, uint32_t ty, uint32_t tiles_x) { return ty * tiles_x + (tx + ty) % tiles_x; }
RT_HD void tile_unpermute(uint32_t permuted, uint32_t tiles_x, uint32_t& tx, uint32_t& ty) {
    ty = permuted / tiles_x;
    uint32_t c = permuted % tiles_x;
    tx = (c + tiles_x - ty % tiles_x) % tiles_x;
}
// The same with the divisions by tiles_x done by multiply-shift (kernels: once per job).
RT_HD void tile_unpermute(uint32_t permuted, FastDiv div_tiles_x, uint32_t& tx, uint32_t& ty) {
    const uint32_t tiles_x = div_tiles_x.d;
    ty = fdiv(permuted, div_tiles_x);
    const uint32_t c = permuted - ty * tiles_x;
    const uint32_t r = ty - fdiv(ty, div_tiles_x) * tiles_x; // ty % tiles_x
    tx = c >= r ? c - r : c + tiles_x - r;
}
// Job index -> (pixel, samples).  64 consecutive jobs — what the lanes of a wave take together — are a 2x2 PIXEL BLOCK
// x 16 CONSECUTIVE CHUNKS of its pixels, not 64 different pixels: the rays a wave starts together are then nearly
// the same ray, and the lanes stay in step longer (measured on final_scene against an 8x8-pixel tile x 1 chunk:
// 4x4 x 4 +0.9 %, 2x2 x 16 +2.7 %, 1 pixel x 64 +3.2 % but 17 % padding jobs; cornell_box -0.5 % for all).
// Jobs are numbered group-major: group g = chunks [16 g, 16 g + 16) of every pixel of the rank, then inside a group
// the 2x2 blocks in tile order.  Chunks past the last one are padding (empty jobs).  `sum_index` is where the job's
// sequential sum goes: chunk-major over (tile, pixel-in-tile), the layout resolve_kernel reads.
#ifndef RT_JOB_BLOCK_LG
#define RT_JOB_BLOCK_LG 1 // (re-measured under round 5's shade phases, final_scene f64 / cornell_box f64: 0 = 1 pixel x 64 chunks, 2 = 4x4 x 4: profiles/r05/README.md)
#endif
constexpr uint32_t JOB_BLOCK_LG = RT_JOB_BLOCK_LG;                    // 2x2 pixels
constexpr uint32_t JOB_GROUP_CHUNKS = 64u >> (2u * JOB_BLOCK_LG);      // 16 chunks
struct JobInfo {
    uint32_t px, row, s, s_end;
    uint32_t sum_index;
    bool real; // false: padding, nothing to trace and no sum to write
};
RT_HD JobInfo job_decode(const RenderConsts& rc, uint32_t job) {
    constexpr uint32_t side = 1u << JOB_BLOCK_LG, lg_px = 2u * JOB_BLOCK_LG, blocks_lg = 6u - lg_px, per_row_lg = 3u - JOB_BLOCK_LG;
    const uint32_t group = fdiv(job, rc.div_jobs_per_group);
    const uint32_t rem = job - group * rc.div_jobs_per_group.d;
    const uint32_t b = rem >> 6, l = rem & 63u;                         // pixel block of the rank, slot in the wave's 64
    const uint32_t tile = b >> blocks_lg, q = b & ((1u << blocks_lg) - 1u), pp = l & ((1u << lg_px) - 1u);
    const uint32_t chunk = group * JOB_GROUP_CHUNKS + (l >> lg_px);
    uint32_t tx, ty;
    tile_unpermute(rc.tile_rank + tile * rc.tile_world, rc.div_tiles_x, tx, ty);
    const uint32_t x = (q & ((1u << per_row_lg) - 1u)) * side + (pp & (side - 1u)), y = (q >> per_row_lg) * side + (pp >> JOB_BLOCK_LG);
    JobInfo j;
    j.px = tx * 8u + x;
    j.row = ty * 8u + y;
    j.real = chunk < rc.n_chunks;
    j.s = j.s_end = 0;
    if (j.real) chunk_samples(rc, rc.chunk_base + chunk, j.s, j.s_end);
    if (j.px >= rc.width || j.row >= rc.height) j.s = j.s_end; // a tile pixel outside the image: an empty job (its sum is 0)
    j.sum_index = chunk * rc.jobs_per_chunk + tile * 64u + y * 8u + x;
    return j;
}
// Host: fill in the job numbering of `rc` (after plan_chunks); false when there are 2^32 or more jobs.
inline bool plan_jobs(RenderConsts& rc) {
    rc.jobs_per_chunk = rc.my_tiles * 64u;
    const uint64_t groups = (uint64_t(rc.n_chunks) + JOB_GROUP_CHUNKS - 1) / JOB_GROUP_CHUNKS;
    const uint64_t per_group = uint64_t(rc.jobs_per_chunk) * JOB_GROUP_CHUNKS, total = groups * per_group;
    if (per_group >= (1ull << 32) || total >= (1ull << 32)) return false;
    rc.div_jobs_per_group = make_fastdiv(uint32_t(std::max<uint64_t>(1, per_group)));
    rc.n_jobs = uint32_t(total);
    return true;
}

} // namespace RT_ARITH_NS
} // namespace rt
