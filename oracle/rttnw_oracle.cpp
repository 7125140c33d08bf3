// =====================================================================================
// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// CPU restatement (C++17, f64) of luliic2/rttnw's per-pixel sample loop and BVH/primitive
// intersection path.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
// load this library; the product (rttnw_amd/) never links, imports or calls it.
//
// PARITY PINNING: the reference is a Rust binary that (a) cannot be compiled in this image (no
// rustc/cargo; scenes.rs has unresolved imports) and (b) seeds every random draw from the OS
// (`rand::thread_rng()`), so it has no tests, no golden vectors and no reproducible output.  This
// restatement is therefore pinned by (1) closed-form known answers derived from the cited formulas
// (tests/test_oracle_kat.py) and (2) block statistics of the two renders the
// reference commits (cornel_box.png / image.png -> tests/golden/reference_png_stats.json, made by
// tests/golden/make_reference_png_stats.py).  The RNG *stream* (rand 0.8.3 / rand_chacha 0.3.0,
// Cargo.lock:361,373) is unpinnable by construction: "RNG parity unpinned"; the keyed generator
// below replaces it (DESIGN.md "RNG").
//
// Every function cites the reference file:line it follows (paths relative to the reference root).
// The arithmetic is written in the reference's operation order; nothing here is shared with the
// product's sources except the public boundary header include/rttnw_hip.h (types only).
// =====================================================================================
#include "../include/rttnw_hip.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr double kPi = 3.14159265358979323846264338327950288; // std::f64::consts::PI
constexpr double kInf = std::numeric_limits<double>::infinity();

// ---------------------------------------------------------------------------------------------
// Keyed RNG (replaces rand::thread_rng(); spec in DESIGN.md "RNG", shared by definition — not by
// source — with rttnw_amd/csrc/rt_core.hpp).
// ---------------------------------------------------------------------------------------------
constexpr uint64_t GAMMA = 0x9E3779B97F4A7C15ull;
inline uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31; return z;
}
inline uint64_t sample_key(uint64_t seed, uint64_t pixel, uint64_t sample) {
    uint64_t k0 = mix64(seed + GAMMA);
    uint64_t k1 = mix64(k0 + pixel * 0xD1B54A32D192ED03ull);
    return mix64(k1 + sample * 0x8CB92BA72F3D8DD7ull);
}
// The word of a draw: the Weyl step on the sample's key, then a mixer of two multiplications by 32-bit
// constants between three folds of the high half into the low one (the sample keys and the scene streams keep SplitMix64's finaliser).
inline uint64_t draw_mix(uint64_t z) {
    z ^= z >> 32; z *= 0x9E3779B1ull;
    z ^= z >> 32; z *= 0x85EBCA6Bull;
    z ^= z >> 32; return z;
}
inline uint64_t keyed_word(uint64_t key, uint32_t ctr) { return draw_mix(key + (uint64_t(ctr) + 1) * GAMMA); }
inline double keyed_uniform(uint64_t key, uint32_t ctr) {
    return double(keyed_word(key, ctr) >> 11) * (1.0 / 9007199254740992.0); // 53 bits, [0,1) like rand's gen::<f64>()
}
// counter layout: block 0 = camera, block b+1 = bounce b
constexpr uint32_t SLOT_JITTER_U = 0, SLOT_JITTER_V = 1, SLOT_TIME = 2, SLOT_LENS = 8;
constexpr uint32_t SLOT_MEDIUM = 0, SLOT_DIELECTRIC = 16, SLOT_SCATTER = 32;
inline uint32_t ctr_of(uint32_t block, uint32_t slot) { return block * 1024u + slot; }

// Scene-construction stream (Perlin tables, reference-style BVH axes, host scene builders).
struct SceneRng {
    uint64_t s;
    SceneRng(uint64_t seed, uint64_t stream)
        : s(mix64(seed + GAMMA) ^ mix64((stream + 1) * 0xD1B54A32D192ED03ull)) {}
    uint64_t next_u64() { s += GAMMA; return mix64(s); }
    double next_f64() { return double(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
    double range(double a, double b) { return a + (b - a) * next_f64(); }
    uint32_t below(uint32_t n) { return uint32_t(next_f64() * double(n)); }
};
constexpr uint64_t STREAM_PERLIN = 0x100, STREAM_BVH = 0x200;

// ---------------------------------------------------------------------------------------------
// Vec3f — src/math/vec3.rs:27-262
// ---------------------------------------------------------------------------------------------
struct V3 {
    double x = 0, y = 0, z = 0;
    V3() = default;
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    double& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }       // vec3.rs:196-205
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }       // vec3.rs:206-211
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }       // vec3.rs:213-218
inline V3 operator*(V3 a, double k) { return {a.x * k, a.y * k, a.z * k}; }         // vec3.rs:219-224
inline V3 operator*(double k, V3 a) { return {a.x * k, a.y * k, a.z * k}; }         // vec3.rs:226-231
inline V3 operator/(V3 a, double k) { return {a.x / k, a.y / k, a.z / k}; }         // vec3.rs:239-244
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }                            // vec3.rs:251-257
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }         // vec3.rs:77-79
inline V3 cross(V3 a, V3 b) {                                                        // vec3.rs:82-88
    return {a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x};
}
inline double squared_length(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }    // vec3.rs:93-95 (powf(2.0) == x*x)
inline double magnitude(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); } // vec3.rs:90-92
inline V3 unit(V3 a) { double k = 1.0 / magnitude(a); return a * k; }               // vec3.rs:97-100
inline V3 reflect(V3 v, V3 n) { return v - 2.0 * dot(v, n) * n; }                   // vec3.rs:112-114
inline V3 refract(V3 v, V3 n, double etai_over_etat) {                              // vec3.rs:116-121
    double cos_theta = std::fmin(dot(-v, n), 1.0);
    V3 perp = etai_over_etat * (v + cos_theta * n);
    V3 par = -std::sqrt(std::fabs(1. - squared_length(perp))) * n;
    return perp + par;
}

// ---------------------------------------------------------------------------------------------
// Ray — src/math/ray.rs:9-27
// ---------------------------------------------------------------------------------------------
struct Ray {
    V3 a, b;
    double time = 0;
    V3 origin() const { return a; }
    V3 direction() const { return b; }
    V3 point_at_parameter(double t) const { return a + t * b; }
};

// Per-path context: the keyed RNG position + work counters (counters are instrumentation only).
struct Counters {
    uint64_t rays = 0, nodes = 0, prims = 0, texels = 0;
};
struct PathCtx {
    uint64_t key = 0;
    uint32_t bounce = 0; // 0-based index of the current world.hit() call along the path
    uint32_t quirks = RTTNW_QUIRKS_REFERENCE;
    Counters* cnt = nullptr;
    double uniform(uint32_t slot) const { return keyed_uniform(key, ctr_of(bounce + 1, slot)); }
};

// Vec3f::random_in_unit_space — vec3.rs:149-160 (rejection in the unit BALL, three uniforms per iteration).
// Draw spec of the keyed generator for these: iteration `it` owns slots 32 + 4 it .. 32 + 4 it + 2; the word
// of the first gives the 21 leading bits of the three uniforms, the words of the other two their remaining 32 bits each (the high
// and the low half of the second word, the high half of the third).
inline double ball_uniform(const PathCtx& ctx, uint32_t it, int c) {
    const uint64_t h = keyed_word(ctx.key, ctr_of(ctx.bounce + 1, SLOT_SCATTER + 4 * it));
    const uint64_t second = keyed_word(ctx.key, ctr_of(ctx.bounce + 1, SLOT_SCATTER + 4 * it + 1));
    const uint64_t third = keyed_word(ctx.key, ctr_of(ctx.bounce + 1, SLOT_SCATTER + 4 * it + 2));
    const uint64_t field = c == 0 ? h >> 43 : (c == 1 ? (h >> 22) & 0x1FFFFFull : (h >> 1) & 0x1FFFFFull);
    const uint64_t low = c == 0 ? second >> 32 : (c == 1 ? second & 0xFFFFFFFFull : third >> 32);
    return double((field << 32) | low) * (1.0 / 9007199254740992.0);
}
inline V3 random_in_unit_space(const PathCtx& ctx) {
    for (uint32_t it = 0;; ++it) {
        V3 r(ball_uniform(ctx, it, 0), ball_uniform(ctx, it, 1), ball_uniform(ctx, it, 2));
        V3 v = 2.0 * r - V3(1.0, 1.0, 1.0);
        if (squared_length(v) < 1.0) return v;
    }
}

// ---------------------------------------------------------------------------------------------
// Bound — src/math/bound.rs:5-46
// ---------------------------------------------------------------------------------------------
struct Bound {
    V3 min, max;
    bool hit(const Ray& ray, double tmin, double tmax) const { // bound.rs:13-32
        for (int d = 0; d < 3; ++d) {
            double inv = 1.0 / ray.b[d];
            double t0 = (min[d] - ray.a[d]) * inv;
            double t1 = (max[d] - ray.a[d]) * inv;
            if (inv < 0.0) std::swap(t0, t1);
            // Rust f64::max/min return the non-NaN operand; fmax/fmin do the same.
            tmin = std::fmax(t0, tmin);
            tmax = std::fmin(t1, tmax);
            if (tmax < tmin) return false;
        }
        return true;
    }
    Bound surrounding(const Bound& o) const { // bound.rs:34-46
        return {V3(std::fmin(min.x, o.min.x), std::fmin(min.y, o.min.y), std::fmin(min.z, o.min.z)),
                V3(std::fmax(max.x, o.max.x), std::fmax(max.y, o.max.y), std::fmax(max.z, o.max.z))};
    }
};

// ---------------------------------------------------------------------------------------------
// Perlin — src/math/noise.rs:5-108
// ---------------------------------------------------------------------------------------------
struct Perlin {
    V3 random_points[256];
    uint32_t px[256], py[256], pz[256];
    static void permutation(SceneRng& rng, uint32_t* p) { // noise.rs:21-29 (SliceRandom::shuffle)
        for (uint32_t i = 0; i < 256; ++i) p[i] = i;
        for (uint32_t i = 255; i >= 1; --i) {
            uint32_t j = rng.below(i + 1);
            std::swap(p[i], p[j]);
        }
    }
    explicit Perlin(SceneRng& rng) { // noise.rs:40-47: points, then x, y, z permutations
        for (auto& v : random_points) { // noise.rs:15-19: Vec3f::random(-1..1), NOT normalised
            double x = rng.range(-1., 1.), y = rng.range(-1., 1.), z = rng.range(-1., 1.);
            v = V3(x, y, z);
        }
        permutation(rng, px); permutation(rng, py); permutation(rng, pz);
    }
    double noise(V3 p) const { // noise.rs:50-75
        double u = p.x - std::floor(p.x), v = p.y - std::floor(p.y), w = p.z - std::floor(p.z);
        int32_t i = int32_t(std::floor(p.x)), j = int32_t(std::floor(p.y)), k = int32_t(std::floor(p.z));
        V3 c[2][2][2];
        for (int di = 0; di < 2; ++di)
            for (int dj = 0; dj < 2; ++dj)
                for (int dk = 0; dk < 2; ++dk)
                    c[di][dj][dk] = random_points[px[(i + di) & 255] ^ py[(j + dj) & 255] ^ pz[(k + dk) & 255]];
        return interpolation(c, u, v, w);
    }
    static double interpolation(const V3 c[2][2][2], double u, double v, double w) { // noise.rs:77-94
        double uu = u * u * (3. - 2. * u), vv = v * v * (3. - 2. * v), ww = w * w * (3. - 2. * w);
        double acc = 0.0;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    V3 weight(u - i, v - j, w - k);
                    acc += (i * uu + (1 - i) * (1. - uu)) * (j * vv + (1 - j) * (1. - vv)) *
                           (k * ww + (1 - k) * (1. - ww)) * dot(c[i][j][k], weight);
                }
        return acc;
    }
    double turbulence(V3 p, uint32_t depth) const { // noise.rs:96-108: signed sum, no abs
        double acc = 0.0, weight = 1.0;
        for (uint32_t o = 0; o < depth; ++o) {
            acc = acc + weight * noise(p);
            weight = weight * 0.5;
            p = p * 2.0;
        }
        return acc;
    }
};

// ---------------------------------------------------------------------------------------------
// Textures — src/math/texture.rs
// ---------------------------------------------------------------------------------------------
struct Texture {
    virtual ~Texture() = default;
    virtual V3 value(double u, double v, V3 p, Counters* cnt) const = 0;
};
struct SolidTexture : Texture { // texture.rs:9-13
    V3 c;
    explicit SolidTexture(V3 c_) : c(c_) {}
    V3 value(double, double, V3, Counters*) const override { return c; }
};
struct CheckerTexture : Texture { // texture.rs:15-30
    std::shared_ptr<Texture> odd, even;
    V3 value(double u, double v, V3 p, Counters* cnt) const override {
        double sines = std::sin(10.0 * p.x) * std::sin(10.0 * p.y) * std::sin(10.0 * p.z);
        return sines < 0.0 ? odd->value(u, v, p, cnt) : even->value(u, v, p, cnt);
    }
};
struct NoiseTexture : Texture { // texture.rs:32-59
    Perlin noise;
    double scale;
    NoiseTexture(SceneRng& rng, double s) : noise(rng), scale(s) {}
    V3 value(double, double, V3 p, Counters*) const override {
        return V3(1.0, 1.0, 1.0) * 0.5 * (1. + std::sin(scale * p.z + 10. * noise.turbulence(p, 7)));
    }
};
struct ImageTexture : Texture { // texture.rs:64-107
    std::vector<uint8_t> data; // RGBA8; empty == load failure
    uint32_t w = 0, h = 0;
    V3 value(double u, double v, V3, Counters* cnt) const override {
        if (data.empty()) return V3(0., 1., 1.); // texture.rs:102-105
        // f64::clamp(0., 1.): a NaN stays NaN (and then casts to texel 0)
        if (u < 0.) u = 0.;
        if (u > 1.) u = 1.;
        if (v < 0.) v = 0.;
        if (v > 1.) v = 1.;
        v = 1. - v;
        // Rust `as u32` saturates and maps NaN to 0.
        auto to_u32 = [](double x) -> uint32_t {
            if (!(x == x)) return 0u;
            if (x <= 0.0) return 0u;
            if (x >= 4294967295.0) return 4294967295u;
            return uint32_t(x);
        };
        uint32_t i = to_u32(u * double(w)), j = to_u32(v * double(h));
        if (i >= w) i = w - 1;
        if (j >= h) j = h - 1;
        if (cnt) cnt->texels++;
        const uint8_t* px = &data[(size_t(j) * w + i) * 4];
        const double s = 1.0 / 255.0;
        return V3(px[0] * s, px[1] * s, px[2] * s); // Vec3f::scaled — vec3.rs:44-51
    }
};

// ---------------------------------------------------------------------------------------------
// HitRecord / Material — hittable.rs:15-44, material.rs
// ---------------------------------------------------------------------------------------------
struct Material;
struct HitRecord {
    double t = 0;
    V3 p, normal;
    const Material* material = nullptr;
    double u = 0, v = 0;
    bool front_face = false;
};
inline void face_normal(const Ray& ray, V3 outward, V3& normal, bool& front_face) { // hittable.rs:30-44
    front_face = dot(ray.direction(), outward) < 0.;
    normal = front_face ? outward : -outward;
}

struct Material {
    int id = -1;
    virtual ~Material() = default;
    virtual bool scatter(const Ray& ray, const HitRecord& rec, const PathCtx& ctx, V3& att, Ray& out) const = 0;
    virtual V3 emitted(double, double, V3, Counters*) const { return V3(0., 0., 0.); } // material.rs:10-12
};
struct Lambertian : Material { // material.rs:89-100
    std::shared_ptr<Texture> albedo;
    bool scatter(const Ray& ray, const HitRecord& rec, const PathCtx& ctx, V3& att, Ray& out) const override {
        V3 target = rec.p + rec.normal + random_in_unit_space(ctx);
        out.a = rec.p;
        out.b = target - rec.p;
        out.time = ray.time;
        att = albedo->value(rec.u, rec.v, rec.p, ctx.cnt);
        return true;
    }
};
struct Metal : Material { // material.rs:134-149
    V3 albedo;
    double fuzz;
    bool scatter(const Ray& ray, const HitRecord& rec, const PathCtx& ctx, V3& att, Ray& out) const override {
        V3 reflected = reflect(unit(ray.direction()), rec.normal);
        out.a = rec.p;
        out.b = reflected + fuzz * random_in_unit_space(ctx); // drawn even when fuzz == 0 (Q5)
        out.time = ray.time;
        att = albedo;
        return dot(out.direction(), rec.normal) > 0.0;
    }
};
struct Dielectric : Material { // material.rs:173-204
    double ri;
    static double schlick(double cosine, double ri) { // material.rs:173-176
        double r0 = (1.0 - ri) / (1.0 + ri);
        r0 = r0 * r0;
        return r0 + (1.0 - r0) * std::pow(1.0 - cosine, 5.0);
    }
    bool scatter(const Ray& ray, const HitRecord& rec, const PathCtx& ctx, V3& att, Ray& out) const override {
        att = V3(1.0, 1.0, 1.0);
        double ratio = rec.front_face ? 1.0 / ri : ri;
        // ORACLE-ONLY experiment switch (tests/test_oracle_png_pins.py): image.png shows its two dielectric spheres as
        // upright see-through "bubbles" with a reflecting ring, not as the inverting glass balls material.rs:179-203
        // renders; bit 0x100 evaluates the variant with the two ratios swapped to test that reading of image.png.
        if (ctx.quirks & 0x100u) ratio = rec.front_face ? ri : 1.0 / ri;
        V3 ud = unit(ray.direction());
        double cos_theta = std::fmin(dot(-ud, rec.normal), 1.);
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        bool cannot_refract = ratio * sin_theta > 1.0;
        // short-circuit `||`: the uniform is consumed only when refraction is possible (Q6)
        V3 dir = (cannot_refract || schlick(cos_theta, ratio) > ctx.uniform(SLOT_DIELECTRIC))
                     ? reflect(ud, rec.normal)
                     : refract(ud, rec.normal, ratio);
        out.a = rec.p;
        out.b = dir;
        out.time = ray.time;
        return true;
    }
};
struct DiffuseLight : Material { // material.rs:242-250 (emits from both faces, Q7)
    std::shared_ptr<Texture> emit;
    bool scatter(const Ray&, const HitRecord&, const PathCtx&, V3&, Ray&) const override { return false; }
    V3 emitted(double u, double v, V3 p, Counters* cnt) const override { return emit->value(u, v, p, cnt); }
};
struct Isotropic : Material { // material.rs:252-266
    std::shared_ptr<Texture> albedo;
    bool scatter(const Ray& ray, const HitRecord& rec, const PathCtx& ctx, V3& att, Ray& out) const override {
        out.a = rec.p;
        out.b = random_in_unit_space(ctx);
        out.time = ray.time;
        att = albedo->value(rec.u, rec.v, rec.p, ctx.cnt);
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// Hittables — src/math/hittable.rs
// ---------------------------------------------------------------------------------------------
struct Hittable {
    virtual ~Hittable() = default;
    virtual bool hit(const Ray& ray, double t_min, double t_max, const PathCtx& ctx, HitRecord& rec) const = 0;
    virtual bool bounding_box(double t0, double t1, Bound& out) const = 0;
};
using HittablePtr = std::shared_ptr<Hittable>;

inline void sphere_uv(V3 p, double& u, double& v) { // hittable.rs:77-83
    double theta = std::acos(-p.y);
    double phi = std::atan2(-p.z, p.x) + kPi;
    u = phi / (2.0 * kPi);
    v = theta / kPi;
}

struct Sphere : Hittable { // hittable.rs:86-131
    V3 center;
    double radius;
    std::shared_ptr<Material> material;
    bool hit(const Ray& ray, double t_min, double t_max, const PathCtx& ctx, HitRecord& rec) const override {
        if (ctx.cnt) ctx.cnt->prims++;
        V3 oc = ray.origin() - center;
        double a = dot(ray.direction(), ray.direction());
        double half_b = dot(oc, ray.direction());
        double c = dot(oc, oc) - radius * radius;
        double discriminant = half_b * half_b - a * c;
        if (discriminant < 0.0) return false;
        double sqrtd = std::sqrt(discriminant);
        double root = (-half_b - sqrtd) / a;
        if (root < t_min || t_max < root) {
            root = (-half_b + sqrtd) / a;
            if (root < t_min || t_max < root) return false;
        }
        rec.t = root;
        rec.p = ray.point_at_parameter(root);
        V3 normal = (rec.p - center) / radius;
        sphere_uv(normal, rec.u, rec.v);
        face_normal(ray, normal, rec.normal, rec.front_face);
        rec.material = material.get();
        return true;
    }
    bool bounding_box(double, double, Bound& out) const override {
        out = {center - V3(radius, radius, radius), center + V3(radius, radius, radius)};
        return true;
    }
};

struct List : Hittable { // hittable.rs:134-177
    std::vector<HittablePtr> list;
    bool hit(const Ray& ray, double t_min, double t_max, const PathCtx& ctx, HitRecord& rec) const override {
        bool any = false;
        double closest = t_max;
        HitRecord tmp;
        for (const auto& i : list) {
            if (i->hit(ray, t_min, closest, ctx, tmp)) {
                closest = tmp.t;
                rec = tmp;
                any = true;
            }
        }
        return any;
    }
    bool bounding_box(double t0, double t1, Bound& out) const override { // hittable.rs:165-176
        if (list.empty()) return false;
        Bound acc;
        if (!list[0]->bounding_box(t0, t1, acc)) return false;
        for (size_t k = 1; k < list.size(); ++k) {
            Bound b;
            if (!list[k]->bounding_box(t0, t1, b)) return false;
            acc = b.surrounding(acc);
        }
        out = acc;
        return true;
    }
};

struct MovingSphere : Hittable { // hittable.rs:179-245
    V3 c0, c1;
    double time0, time1, radius;
    std::shared_ptr<Material> material;
    V3 center(double time) const { return c0 + ((time - time0) / (time1 - time0)) * (c1 - c0); } // :187-191
    bool hit(const Ray& ray, double t_min, double t_max, const PathCtx& ctx, HitRecord& rec) const override {
        if (ctx.cnt) ctx.cnt->prims++;
        V3 oc = ray.origin() - center(ray.time);
        double a = dot(ray.direction(), ray.direction());
        double half_b = dot(oc, ray.direction());
        double c = dot(oc, oc) - radius * radius;
        double discriminant = half_b * half_b - a * c;
        if (discriminant < 0.0) return false;
        double sqrtd = std::sqrt(discriminant);
        double root = (-half_b - sqrtd) / a;
        if (root < t_min || t_max < root) {
            root = (-half_b + sqrtd) / a;
            if (root < t_min || t_max < root) return false;
        }
        rec.t = root;
        rec.p = ray.point_at_parameter(root);
        V3 outward = (rec.p - center(ray.time)) / radius;
        rec.u = 0.0; rec.v = 0.0; // hittable.rs:220
        face_normal(ray, outward, rec.normal, rec.front_face);
        rec.material = material.get();
        return true;
    }
    bool bounding_box(double t0, double t1, Bound& out) const override {
        V3 r(radius, radius, radius);
        Bound b0{center(t0) - r, center(t0) + r}, b1{center(t1) - r, center(t1) + r};
        out = b0.surrounding(b1);
        return true;
    }
};

struct Rectangle : Hittable { // hittable.rs:434-547; plane -> (axis0, axis1, k): hittable.rs:450-488
    int axis0, axis1, kaxis;
    double p0s, p0e, p1s, p1e, k;
    std::shared_ptr<Material> material;
    Rectangle(int plane, double a0, double a1, double b0, double b1, double k_, std::shared_ptr<Material> m)
        : p0s(a0), p0e(a1), p1s(b0), p1e(b1), k(k_), material(std::move(m)) {
        if (plane == RTTNW_XY) { axis0 = 0; axis1 = 1; kaxis = 2; }
        else if (plane == RTTNW_XZ) { axis0 = 0; axis1 = 2; kaxis = 1; }
        else { axis0 = 1; axis1 = 2; kaxis = 0; }
    }
    bool hit(const Ray& ray, double t_min, double t_max, const PathCtx& ctx, HitRecord& rec) const override {
        if (ctx.cnt) ctx.cnt->prims++;
        double t = (k - ray.a[kaxis]) / ray.b[kaxis];
        if (t < t_min || t > t_max) return false;
        double a = ray.a[axis0] + t * ray.b[axis0];
        double b = ray.a[axis1] + t * ray.b[axis1];
        // Range::contains: start <= x && x < end (half-open, Q9) — hittable.rs:511
        if (!(p0s <= a && a < p0e) || !(p1s <= b && b < p1e)) return false;
        rec.u = (a - p0s) / (p0e - p0s);
        rec.v = (b - p1s) / (p1e - p1s);
        V3 outward(0, 0, 0);
        outward.at(kaxis) = 1.;
        face_normal(ray, outward, rec.normal, rec.front_face);
        rec.p = ray.point_at_parameter(t);
        rec.t = t;
        rec.material = material.get();
        return true;
    }
    bool bounding_box(double, double, Bound& out) const override { // hittable.rs:532-546
        V3 mn, mx;
        mn.at(axis0) = p0s; mn.at(axis1) = p1s; mn.at(kaxis) = k - 0.0001;
        mx.at(axis0) = p0e; mx.at(axis1) = p1e; mx.at(kaxis) = k + 0.0001;
        out = {mn, mx};
        return true;
    }
};

struct Cube : Hittable { // hittable.rs:549-592
    V3 box_min, box_max;
    List sides;
    Cube(V3 mn, V3 mx, const std::shared_ptr<Material>& m) : box_min(mn), box_max(mx) {
        // Plane::points / rectangles — hittable.rs:414-426,451-453,464-466,477-479
        sides.list.push_back(std::make_shared<Rectangle>(RTTNW_XY, mn.x, mx.x, mn.y, mx.y, mn.z, m));
        sides.list.push_back(std::make_shared<Rectangle>(RTTNW_XY, mn.x, mx.x, mn.y, mx.y, mx.z, m));
        sides.list.push_back(std::make_shared<Rectangle>(RTTNW_XZ, mn.x, mx.x, mn.z, mx.z, mn.y, m));
        sides.list.push_back(std::make_shared<Rectangle>(RTTNW_XZ, mn.x, mx.x, mn.z, mx.z, mx.y, m));
        sides.list.push_back(std::make_shared<Rectangle>(RTTNW_YZ, mn.y, mx.y, mn.z, mx.z, mn.x, m));
        sides.list.push_back(std::make_shared<Rectangle>(RTTNW_YZ, mn.y, mx.y, mn.z, mx.z, mx.x, m));
    }
    bool hit(const Ray& ray, double t_min, double t_max, const PathCtx& ctx, HitRecord& rec) const override {
        return sides.hit(ray, t_min, t_max, ctx, rec);
    }
    bool bounding_box(double, double, Bound& out) const override { out = {box_min, box_max}; return true; }
};

struct BvhTree : Hittable { // hittable.rs:248-373
    HittablePtr left, right;
    Bound bound;
    static double key_of(const Hittable& h, int axis) { // hittable.rs:323-333
        Bound b;
        if (!h.bounding_box(0.0, 0.0, b)) { std::fprintf(stderr, "No bounding box in BvhTree constructor\n"); b = Bound{}; }
        return b.min[axis];
    }
    // hittable.rs:265-321 — reproduces the degenerate builder: random axis per node, the WHOLE
    // remaining vector is re-sorted, leaves are removed from index 0.
    BvhTree(std::vector<HittablePtr>& objects, size_t start, size_t end, double t0, double t1, SceneRng& rng) {
        int axis = int(rng.below(3));
        size_t span = end - start;
        if (span == 1) {
            left = right = objects.front();
            objects.erase(objects.begin());
        } else if (span == 2) {
            HittablePtr first = objects.front(); objects.erase(objects.begin());
            HittablePtr second = objects.front(); objects.erase(objects.begin());
            if (key_of(*first, axis) < key_of(*second, axis)) { left = first; right = second; }
            else { left = second; right = first; }
        } else {
            std::stable_sort(objects.begin(), objects.end(), [axis](const HittablePtr& x, const HittablePtr& y) {
                return key_of(*x, axis) < key_of(*y, axis);
            });
            size_t mid = start + span / 2;
            left = std::make_shared<BvhTree>(objects, start, mid, t0, t1, rng);
            right = std::make_shared<BvhTree>(objects, mid, end, t0, t1, rng);
        }
        Bound bl{}, br{};
        if (!left->bounding_box(t0, t1, bl)) std::fprintf(stderr, "No bounding box in BvhTree constructor\n");
        if (!right->bounding_box(t0, t1, br)) std::fprintf(stderr, "No bounding box in BvhTree constructor\n");
        bound = bl.surrounding(br);
    }
    bool hit(const Ray& ray, double t_min, double t_max, const PathCtx& ctx, HitRecord& rec) const override { // :356-368
        if (ctx.cnt) ctx.cnt->nodes++;
        if (!bound.hit(ray, t_min, t_max)) return false;
        HitRecord l, r;
        bool hl = left->hit(ray, t_min, t_max, ctx, l);
        double t = hl ? l.t : t_max;
        bool hr = right->hit(ray, t_min, t, ctx, r);
        if (hr) { rec = r; return true; }
        if (hl) { rec = l; return true; }
        return false;
    }
    bool bounding_box(double, double, Bound& out) const override { out = bound; return true; }

    // ---- second builder, ORACLE-ONLY (rto_scene_set_bvh_builder(scene, 1)); no counterpart in the reference --------------
    // The reference's builder above re-sorts the whole remaining vector at every node (hittable.rs:300-303): O(n^2 log n), it
    // cannot make config 5's tree of 10^6 spheres.  Closest-hit results do not depend on the tree's topology (SURVEY Q12:
    // BvhTree::hit, :356-368, returns the nearest hit of its two sides whatever they hold; an exact tie between two DIFFERENT
    // objects has measure zero in the sphere cloud), so for big lists the oracle may hang the same `hit` on another
    // topology: a plain recursive median split along the longest axis of the node's box.  A node is still a BvhTree with the
    // reference's `hit`; a range of one object is the object itself.  tests/test_oracle_kat.py holds this builder against
    // the reference-shaped one (bit-identical images at 3 000 spheres) before anything is pinned to it.
    struct Item { Bound box; HittablePtr obj; };
    BvhTree(HittablePtr l, HittablePtr r, const Bound& b) : left(std::move(l)), right(std::move(r)), bound(b) {}
    static HittablePtr median_split(std::vector<Item>& items, size_t lo, size_t hi) {
        if (hi - lo == 1) return items[lo].obj;
        Bound all = items[lo].box;
        for (size_t k = lo + 1; k < hi; ++k) all = all.surrounding(items[k].box);
        int axis = 0;
        for (int d = 1; d < 3; ++d)
            if (all.max[d] - all.min[d] > all.max[axis] - all.min[axis]) axis = d;
        const size_t mid = lo + (hi - lo) / 2;
        std::nth_element(items.begin() + lo, items.begin() + mid, items.begin() + hi, [axis](const Item& x, const Item& y) {
            return x.box.min[axis] + x.box.max[axis] < y.box.min[axis] + y.box.max[axis];
        });
        HittablePtr l = median_split(items, lo, mid), r = median_split(items, mid, hi);
        return std::make_shared<BvhTree>(std::move(l), std::move(r), all);
    }
    static HittablePtr build_median_split(const std::vector<HittablePtr>& objects, double t0, double t1) {
        std::vector<Item> items(objects.size());
        for (size_t k = 0; k < objects.size(); ++k) {
            items[k].obj = objects[k];
            if (!objects[k]->bounding_box(t0, t1, items[k].box)) std::fprintf(stderr, "No bounding box in BvhTree constructor\n");
        }
        HittablePtr root = median_split(items, 0, items.size());
        if (items.size() == 1) { // a one-object list still gets its node, like the reference's span == 1 (hittable.rs:271-274)
            return std::make_shared<BvhTree>(root, root, items[0].box);
        }
        return root;
    }
};

struct Translate : Hittable { // hittable.rs:594-629
    HittablePtr item;
    V3 offset;
    bool hit(const Ray& ray, double t_min, double t_max, const PathCtx& ctx, HitRecord& rec) const override {
        Ray moved{ray.origin() - offset, ray.direction(), ray.time};
        if (!item->hit(moved, t_min, t_max, ctx, rec)) return false;
        V3 n; bool ff;
        face_normal(moved, rec.normal, n, ff); // re-flips an already flipped normal (Q3)
        rec.normal = n;
        rec.front_face = ff;
        rec.p = rec.p + offset;
        return true;
    }
    bool bounding_box(double t0, double t1, Bound& out) const override {
        Bound b;
        if (!item->bounding_box(t0, t1, b)) return false;
        out = {b.min + offset, b.max + offset};
        return true;
    }
};

struct YRotate : Hittable { // hittable.rs:631-722
    HittablePtr item;
    double sin_theta, cos_theta;
    bool has_bound;
    Bound bound;
    YRotate(HittablePtr it, double angle) : item(std::move(it)) { // hittable.rs:640-682
        double radians = angle * (kPi / 180.0); // f64::to_radians
        sin_theta = std::sin(radians);
        cos_theta = std::cos(radians);
        Bound b{};
        has_bound = item->bounding_box(0., 1., b);
        V3 mn(kInf, kInf, kInf), mx(-kInf, -kInf, -kInf);
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    double x = i * b.max.x + (1 - i) * b.min.x;
                    double y = j * b.max.y + (1 - j) * b.min.y;
                    double z = k * b.max.z + (1 - k) * b.min.z;
                    double x2 = cos_theta * x + sin_theta * z;
                    double z2 = -sin_theta * x2 + cos_theta * z; // shadowed-x bug, latent (Q2) :661-662
                    V3 tmp(x2, y, z2);
                    for (int c = 0; c < 3; ++c) {
                        mn.at(c) = std::fmin(mn[c], tmp[c]);
                        mx.at(c) = std::fmax(mx[c], tmp[c]);
                    }
                }
        bound = {mn, mx};
    }
    bool hit(const Ray& ray, double t_min, double t_max, const PathCtx& ctx, HitRecord& rec) const override { // :686-716
        V3 origin = ray.origin(), direction = ray.direction();
        origin.x = cos_theta * ray.a.x - sin_theta * ray.a.z;
        origin.z = sin_theta * ray.a.x + cos_theta * ray.a.z;
        direction.x = cos_theta * ray.b.x - sin_theta * ray.b.z;
        direction.z = sin_theta * ray.b.x + cos_theta * ray.b.z;
        Ray rotated{origin, direction, ray.time};
        if (!item->hit(rotated, t_min, t_max, ctx, rec)) return false;
        if (ctx.quirks & RTTNW_QUIRK_YROTATE_BACKROT) {
            // Q1 — hittable.rs:700-705: the z line reads the ALREADY OVERWRITTEN x component.
            rec.p.x = cos_theta * rec.p.x + sin_theta * rec.p.z;
            rec.p.z = -sin_theta * rec.p.x + cos_theta * rec.p.z;
            rec.normal.x = cos_theta * rec.normal.x + sin_theta * rec.normal.z;
            rec.normal.z = -sin_theta * rec.normal.x + cos_theta * rec.normal.z;
        } else {
            double px = rec.p.x, nx = rec.normal.x;
            rec.p.x = cos_theta * px + sin_theta * rec.p.z;
            rec.p.z = -sin_theta * px + cos_theta * rec.p.z;
            rec.normal.x = cos_theta * nx + sin_theta * rec.normal.z;
            rec.normal.z = -sin_theta * nx + cos_theta * rec.normal.z;
        }
        V3 n; bool ff;
        face_normal(rotated, rec.normal, n, ff);
        rec.normal = n;
        rec.front_face = ff;
        return true;
    }
    bool bounding_box(double, double, Bound& out) const override { out = bound; return true; }
};

struct ConstantMedium : Hittable { // hittable.rs:724-801
    HittablePtr boundary;
    std::shared_ptr<Isotropic> phase_function;
    double neg_inv_density;
    uint32_t medium_index = 0; // RNG slot of this medium (creation order)
    bool hit(const Ray& ray, double t_min, double t_max, const PathCtx& ctx, HitRecord& rec) const override {
        HitRecord r1, r2;
        // the debug branch (hittable.rs:743-744) short-circuits: no draw is consumed
        if (!boundary->hit(ray, -kInf, kInf, ctx, r1)) return false;
        if (!boundary->hit(ray, r1.t + 0.0001, kInf, ctx, r2)) return false;
        r1.t = std::fmax(r1.t, t_min);
        r2.t = std::fmin(r2.t, t_max);
        if (r1.t >= r2.t) return false; // before any draw (Q13)
        r1.t = std::fmax(r1.t, 0.);
        double ray_length = magnitude(ray.direction());
        double distance_inside = (r2.t - r1.t) * ray_length;
        double hit_distance = neg_inv_density * std::log(ctx.uniform(SLOT_MEDIUM + medium_index));
        if (hit_distance > distance_inside) return false;
        rec.t = r1.t + hit_distance / ray_length;
        rec.p = ray.point_at_parameter(rec.t);
        rec.normal = V3(1., 0., 0.);
        rec.front_face = true;
        rec.material = phase_function.get();
        rec.u = 0.0; rec.v = 0.0;
        return true;
    }
    bool bounding_box(double t0, double t1, Bound& out) const override { return boundary->bounding_box(t0, t1, out); }
};

// ---------------------------------------------------------------------------------------------
// Camera — src/math/camera.rs:18-84
// ---------------------------------------------------------------------------------------------
struct Camera {
    V3 origin, lower_left_corner, horizontal, vertical, u, v, w;
    double lens_radius, open_time, close_time;
    explicit Camera(const rttnw_camera_desc& d) { // camera.rs:32-61
        V3 lookfrom(d.lookfrom[0], d.lookfrom[1], d.lookfrom[2]);
        V3 lookat(d.lookat[0], d.lookat[1], d.lookat[2]);
        V3 vup(d.view_up[0], d.view_up[1], d.view_up[2]);
        lens_radius = d.aperture / 2.0;
        double theta = d.vertical_fov * kPi / 180.0;
        double half_height = std::tan(theta / 2.0);
        double half_width = d.aspect_ratio * half_height;
        origin = lookfrom;
        w = unit(lookfrom - lookat);
        u = unit(cross(vup, w));
        v = cross(w, u);
        lower_left_corner = origin - half_width * d.focus_distance * u - half_height * d.focus_distance * v -
                            d.focus_distance * w;
        horizontal = 2.0 * half_width * d.focus_distance * u;
        vertical = 2.0 * half_height * d.focus_distance * v;
        open_time = d.open_time;
        close_time = d.close_time;
    }
    Ray ray(double s, double t, uint64_t key) const { // camera.rs:63-84
        // random_in_unit_disk: the loop always runs, even when lens_radius == 0 (Q15)
        V3 p;
        for (uint32_t it = 0;; ++it) {
            double a = keyed_uniform(key, ctr_of(0, SLOT_LENS + 2 * it));
            double b = keyed_uniform(key, ctr_of(0, SLOT_LENS + 2 * it + 1));
            p = 2.0 * V3(a, b, 0.0) - V3(1.0, 1.0, 0.0);
            if (dot(p, p) < 1.0) break;
        }
        V3 rd = lens_radius * p;
        V3 offset = u * rd.x + v * rd.y;
        Ray r;
        r.a = origin + offset;
        r.b = lower_left_corner + s * horizontal + t * vertical - origin - offset;
        // gen_range(open..close): uniform in [open, close)
        r.time = open_time + (close_time - open_time) * keyed_uniform(key, ctr_of(0, SLOT_TIME));
        return r;
    }
};

// color() — main.rs:26-45 (recursive, as written)
V3 color(const Ray& ray, V3 background, const Hittable& world, int depth, PathCtx& ctx, double t_min) {
    if (depth <= 0) return V3(0., 0., 0.);
    HitRecord rec;
    if (ctx.cnt) ctx.cnt->rays++;
    if (world.hit(ray, t_min, std::numeric_limits<double>::max(), ctx, rec)) {
        V3 emitted = rec.material->emitted(rec.u, rec.v, rec.p, ctx.cnt);
        V3 att;
        Ray scattered;
        if (rec.material->scatter(ray, rec, ctx, att, scattered)) {
            ctx.bounce += 1;
            return emitted + att * color(scattered, background, world, depth - 1, ctx, t_min);
        }
        return emitted;
    }
    return background;
}

// Gamma + quantise — main.rs:219-225.  Rust `as u8` saturates and maps NaN to 0.
inline uint8_t quantise(double mean) {
    double x = std::sqrt(mean);
    // f64::clamp(0.0, 0.999): NaN stays NaN
    if (x < 0.0) x = 0.0;
    if (x > 0.999) x = 0.999;
    x = x * 256.;
    if (!(x == x)) return 0;
    if (x <= 0.0) return 0;
    if (x >= 255.0) return 255;
    return uint8_t(x);
}

} // namespace

// =============================================================================================
// Scene object table + C API (same shape as include/rttnw_hip.h, prefix rto_)
// =============================================================================================
struct rttnw_scene {
    uint64_t seed = 0;
    enum Kind { TEX, MAT, HIT, LIST };
    struct Obj {
        Kind kind;
        std::shared_ptr<Texture> tex;
        std::shared_ptr<Material> mat;
        HittablePtr hit;
        std::shared_ptr<List> list; // LIST objects are also hittable
        bool consumed = false;      // list moved into a BvhTree
    };
    std::vector<Obj> objs;
    std::shared_ptr<List> world;
    uint32_t n_noise = 0, n_bvh = 0, n_media = 0;
    uint32_t bvh_builder = 0; // 0: the reference's builder (hittable.rs:265-321); 1: BvhTree::build_median_split (oracle-only, big lists)
};

namespace {
thread_local std::string g_err;
int fail(int code, const char* msg) { g_err = msg; return code; }

Texture* get_tex(rttnw_scene* s, rttnw_id id, std::shared_ptr<Texture>* out = nullptr) {
    if (!s || id < 0 || size_t(id) >= s->objs.size() || s->objs[id].kind != rttnw_scene::TEX) return nullptr;
    if (out) *out = s->objs[id].tex;
    return s->objs[id].tex.get();
}
std::shared_ptr<Material> get_mat(rttnw_scene* s, rttnw_id id) {
    if (!s || id < 0 || size_t(id) >= s->objs.size() || s->objs[id].kind != rttnw_scene::MAT) return nullptr;
    return s->objs[id].mat;
}
HittablePtr get_hit(rttnw_scene* s, rttnw_id id) {
    if (!s || id < 0 || size_t(id) >= s->objs.size()) return nullptr;
    auto& o = s->objs[id];
    if (o.kind == rttnw_scene::HIT) return o.hit;
    if (o.kind == rttnw_scene::LIST && !o.consumed) return o.list;
    return nullptr;
}
rttnw_id push_obj(rttnw_scene* s, rttnw_scene::Obj o) {
    s->objs.push_back(std::move(o));
    return rttnw_id(s->objs.size() - 1);
}
rttnw_id push_tex(rttnw_scene* s, std::shared_ptr<Texture> t) {
    rttnw_scene::Obj o; o.kind = rttnw_scene::TEX; o.tex = std::move(t); return push_obj(s, std::move(o));
}
rttnw_id push_mat(rttnw_scene* s, std::shared_ptr<Material> m) {
    rttnw_scene::Obj o; o.kind = rttnw_scene::MAT; o.mat = std::move(m);
    rttnw_id id = push_obj(s, std::move(o));
    s->objs[id].mat->id = id;
    return id;
}
rttnw_id push_hit(rttnw_scene* s, HittablePtr h) {
    rttnw_scene::Obj o; o.kind = rttnw_scene::HIT; o.hit = std::move(h); return push_obj(s, std::move(o));
}
} // namespace

extern "C" {

int rto_scene_create(uint64_t scene_seed, rttnw_scene** out) {
    if (!out) return fail(RTTNW_ERR_INVALID, "out is NULL");
    *out = new rttnw_scene();
    (*out)->seed = scene_seed;
    return RTTNW_OK;
}
void rto_scene_destroy(rttnw_scene* s) { delete s; }

rttnw_id rto_tex_solid(rttnw_scene* s, double r, double g, double b) {
    if (!s) return fail(RTTNW_ERR_INVALID, "scene is NULL");
    return push_tex(s, std::make_shared<SolidTexture>(V3(r, g, b)));
}
rttnw_id rto_tex_checker(rttnw_scene* s, rttnw_id odd, rttnw_id even) {
    std::shared_ptr<Texture> o, e;
    if (!get_tex(s, odd, &o) || !get_tex(s, even, &e)) return fail(RTTNW_ERR_INVALID, "checker: bad texture id");
    auto c = std::make_shared<CheckerTexture>();
    c->odd = o; c->even = e;
    return push_tex(s, c);
}
rttnw_id rto_tex_noise(rttnw_scene* s, double scale) {
    if (!s) return fail(RTTNW_ERR_INVALID, "scene is NULL");
    SceneRng rng(s->seed, STREAM_PERLIN + s->n_noise++);
    return push_tex(s, std::make_shared<NoiseTexture>(rng, scale));
}
rttnw_id rto_tex_image_rgba8(rttnw_scene* s, const uint8_t* rgba, uint32_t w, uint32_t h) {
    if (!s) return fail(RTTNW_ERR_INVALID, "scene is NULL");
    auto t = std::make_shared<ImageTexture>();
    if (rgba && w && h) { t->data.assign(rgba, rgba + size_t(w) * h * 4); t->w = w; t->h = h; }
    return push_tex(s, t);
}
rttnw_id rto_mat_lambertian(rttnw_scene* s, rttnw_id tex) {
    std::shared_ptr<Texture> t;
    if (!get_tex(s, tex, &t)) return fail(RTTNW_ERR_INVALID, "lambertian: bad texture id");
    auto m = std::make_shared<Lambertian>(); m->albedo = t; return push_mat(s, m);
}
rttnw_id rto_mat_metal(rttnw_scene* s, double r, double g, double b, double fuzz) {
    if (!s) return fail(RTTNW_ERR_INVALID, "scene is NULL");
    auto m = std::make_shared<Metal>(); m->albedo = V3(r, g, b); m->fuzz = std::fmin(fuzz, 1.0); // material.rs:129
    return push_mat(s, m);
}
rttnw_id rto_mat_dielectric(rttnw_scene* s, double ri) {
    if (!s) return fail(RTTNW_ERR_INVALID, "scene is NULL");
    auto m = std::make_shared<Dielectric>(); m->ri = ri; return push_mat(s, m);
}
rttnw_id rto_mat_diffuse_light(rttnw_scene* s, rttnw_id tex) {
    std::shared_ptr<Texture> t;
    if (!get_tex(s, tex, &t)) return fail(RTTNW_ERR_INVALID, "diffuse_light: bad texture id");
    auto m = std::make_shared<DiffuseLight>(); m->emit = t; return push_mat(s, m);
}
rttnw_id rto_mat_isotropic(rttnw_scene* s, rttnw_id tex) {
    std::shared_ptr<Texture> t;
    if (!get_tex(s, tex, &t)) return fail(RTTNW_ERR_INVALID, "isotropic: bad texture id");
    auto m = std::make_shared<Isotropic>(); m->albedo = t; return push_mat(s, m);
}
rttnw_id rto_sphere(rttnw_scene* s, const double c[3], double radius, rttnw_id mat) {
    auto m = get_mat(s, mat);
    if (!m || !c) return fail(RTTNW_ERR_INVALID, "sphere: bad material id");
    auto h = std::make_shared<Sphere>(); h->center = V3(c[0], c[1], c[2]); h->radius = radius; h->material = m;
    return push_hit(s, h);
}
rttnw_id rto_moving_sphere(rttnw_scene* s, const double c0[3], const double c1[3], double t0, double t1,
                           double radius, rttnw_id mat) {
    auto m = get_mat(s, mat);
    if (!m || !c0 || !c1) return fail(RTTNW_ERR_INVALID, "moving_sphere: bad material id");
    auto h = std::make_shared<MovingSphere>();
    h->c0 = V3(c0[0], c0[1], c0[2]); h->c1 = V3(c1[0], c1[1], c1[2]);
    h->time0 = t0; h->time1 = t1; h->radius = radius; h->material = m;
    return push_hit(s, h);
}
rttnw_id rto_rectangle(rttnw_scene* s, int plane, double a0, double a1, double b0, double b1, double k,
                       rttnw_id mat) {
    auto m = get_mat(s, mat);
    if (!m || plane < 0 || plane > 2) return fail(RTTNW_ERR_INVALID, "rectangle: bad material id or plane");
    return push_hit(s, std::make_shared<Rectangle>(plane, a0, a1, b0, b1, k, m));
}
rttnw_id rto_cube(rttnw_scene* s, const double mn[3], const double mx[3], rttnw_id mat) {
    auto m = get_mat(s, mat);
    if (!m || !mn || !mx) return fail(RTTNW_ERR_INVALID, "cube: bad material id");
    return push_hit(s, std::make_shared<Cube>(V3(mn[0], mn[1], mn[2]), V3(mx[0], mx[1], mx[2]), m));
}
rttnw_id rto_list(rttnw_scene* s) {
    if (!s) return fail(RTTNW_ERR_INVALID, "scene is NULL");
    rttnw_scene::Obj o; o.kind = rttnw_scene::LIST; o.list = std::make_shared<List>();
    return push_obj(s, std::move(o));
}
int rto_list_push(rttnw_scene* s, rttnw_id list, rttnw_id item) {
    if (!s || list < 0 || size_t(list) >= s->objs.size() || s->objs[list].kind != rttnw_scene::LIST ||
        s->objs[list].consumed)
        return fail(RTTNW_ERR_INVALID, "list_push: bad list id");
    auto h = get_hit(s, item);
    if (!h || item == list) return fail(RTTNW_ERR_INVALID, "list_push: bad item id");
    s->objs[list].list->list.push_back(h);
    return RTTNW_OK;
}
rttnw_id rto_bvh_tree(rttnw_scene* s, rttnw_id list) {
    if (!s || list < 0 || size_t(list) >= s->objs.size() || s->objs[list].kind != rttnw_scene::LIST ||
        s->objs[list].consumed)
        return fail(RTTNW_ERR_INVALID, "bvh_tree: bad list id");
    auto& items = s->objs[list].list->list;
    if (items.empty()) return fail(RTTNW_ERR_INVALID, "bvh_tree: empty list");
    SceneRng rng(s->seed, STREAM_BVH + s->n_bvh++);
    s->objs[list].consumed = true;
    if (s->bvh_builder == 1) return push_hit(s, BvhTree::build_median_split(items, 0., 1.));
    std::vector<HittablePtr> objects = items; // BvhTree::from consumes the list (hittable.rs:254-258)
    size_t n = objects.size();
    auto tree = std::make_shared<BvhTree>(objects, 0, n, 0., 1., rng);
    return push_hit(s, tree);
}
// ORACLE-ONLY switch (not part of include/rttnw_hip.h's builder table, whose entry of this name selects the PRODUCT's builders):
// which topology rto_bvh_tree hangs BvhTree::hit on.  0 = the reference's builder, 1 = median split for lists the
// reference's O(n^2 log n) builder cannot do.  Call before the lists are turned into trees.
int rto_scene_set_bvh_builder(rttnw_scene* s, uint32_t which) {
    if (!s || which > 1) return fail(RTTNW_ERR_INVALID, "set_bvh_builder: 0 (reference-shaped) or 1 (median split)");
    s->bvh_builder = which;
    return RTTNW_OK;
}
rttnw_id rto_translate(rttnw_scene* s, rttnw_id item, const double off[3]) {
    auto h = get_hit(s, item);
    if (!h || !off) return fail(RTTNW_ERR_INVALID, "translate: bad item id");
    auto t = std::make_shared<Translate>(); t->item = h; t->offset = V3(off[0], off[1], off[2]);
    return push_hit(s, t);
}
rttnw_id rto_rotate_y(rttnw_scene* s, rttnw_id item, double deg) {
    auto h = get_hit(s, item);
    if (!h) return fail(RTTNW_ERR_INVALID, "rotate_y: bad item id");
    return push_hit(s, std::make_shared<YRotate>(h, deg));
}
rttnw_id rto_constant_medium(rttnw_scene* s, rttnw_id boundary, double density, rttnw_id tex) {
    auto h = get_hit(s, boundary);
    std::shared_ptr<Texture> t;
    if (!h || !get_tex(s, tex, &t)) return fail(RTTNW_ERR_INVALID, "constant_medium: bad boundary or texture id");
    auto m = std::make_shared<ConstantMedium>();
    m->boundary = h;
    m->phase_function = std::make_shared<Isotropic>();
    m->phase_function->albedo = t;
    m->neg_inv_density = -1. / density; // hittable.rs:733
    m->medium_index = s->n_media++;
    const rttnw_id id = push_hit(s, m);
    m->phase_function->id = id; // probes report the medium's own object id for its phase function (created in place, hittable.rs:733)
    return id;
}
int rto_scene_set_world(rttnw_scene* s, rttnw_id world) {
    if (!s || world < 0 || size_t(world) >= s->objs.size() || s->objs[world].kind != rttnw_scene::LIST ||
        s->objs[world].consumed)
        return fail(RTTNW_ERR_INVALID, "set_world: bad list id");
    s->world = s->objs[world].list;
    return RTTNW_OK;
}
int rto_scene_commit(rttnw_scene* s) {
    if (!s || !s->world) return fail(RTTNW_ERR_STATE, "commit: world not set");
    return RTTNW_OK;
}
const char* rto_last_error(void) { return g_err.c_str(); }

const rttnw_builder_api* rto_builder(void) {
    static const rttnw_builder_api api = {
        rto_scene_create, rto_scene_destroy, rto_tex_solid, rto_tex_checker, rto_tex_noise,
        rto_tex_image_rgba8, rto_mat_lambertian, rto_mat_metal, rto_mat_dielectric,
        rto_mat_diffuse_light, rto_mat_isotropic, rto_sphere, rto_moving_sphere, rto_rectangle,
        rto_cube, rto_list, rto_list_push, rto_bvh_tree, rto_translate, rto_rotate_y,
        rto_constant_medium, rto_scene_set_world, rto_scene_commit, rto_last_error};
    return &api;
}

// ---------------------------------------------------------------------------------------------
// render() — main.rs:184-229.  out_linear: w*h*3 doubles (mean radiance), out_rgba8: w*h*4 bytes,
// both row-major, top row first.  n_threads <= 0: hardware_concurrency().
// Only pixels of tiles owned by (tile_rank, tile_world) are computed (others left untouched).
// ---------------------------------------------------------------------------------------------
static uint32_t oracle_tile_owner(uint32_t tx, uint32_t ty, uint32_t tiles_x, uint32_t world) {
    // same closed form as the product's partition (include/rttnw_hip.h rttnw_tile_layout):
    // rows are rotated by ty so that equal-x tiles of consecutive rows go to different ranks.
    uint32_t permuted = ty * tiles_x + (tx + ty) % tiles_x;
    return permuted % world;
}

// The per-pixel fold of main.rs:209-226 over an arbitrary set of pixels of the W x H image (same keys and jitter as in
// the full render): the window [x0, x1) x [y0, y1) when `xs == nullptr` (outputs are (y1-y0) x (x1-x0) arrays), else the
// n_list pixels (xs[k], ys[k]) (outputs are n_list records).  `out_var` (optional, 3 doubles per pixel): unbiased
// per-channel variance of the pixel's SAMPLE radiances (what the f32-vs-f64 statistical tier needs for its
// 6 sigma / sqrt(spp) bound).
static int render_pixels(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, uint32_t x0, uint32_t y0, uint32_t x1,
                         uint32_t y1, const uint32_t* xs, const uint32_t* ys, uint32_t n_list, double* out_linear,
                         uint8_t* out_rgba8, double* out_var, rttnw_stats* stats, int n_threads) {
    if (!s || !s->world || !cam || !p) return fail(RTTNW_ERR_INVALID, "render: bad arguments");
    if (!p->width || !p->height || !p->spp) return fail(RTTNW_ERR_INVALID, "render: empty image or spp");
    if (!xs && (x0 >= x1 || y0 >= y1 || x1 > p->width || y1 > p->height)) return fail(RTTNW_ERR_INVALID, "render: bad window");
    if (xs)
        for (uint32_t k = 0; k < n_list; ++k)
            if (xs[k] >= p->width || ys[k] >= p->height) return fail(RTTNW_ERR_INVALID, "render: pixel outside the image");
    const uint32_t W = p->width, H = p->height, spp = p->spp, WW = xs ? n_list : x1 - x0, WH = xs ? 1 : y1 - y0;
    const uint32_t chunk = p->spp_chunk ? p->spp_chunk : spp;
    const uint32_t world = p->tile_world ? p->tile_world : 1;
    const uint32_t tiles_x = (W + 7) / 8;
    Camera camera(*cam);
    V3 background(p->background[0], p->background[1], p->background[2]);
    if (n_threads <= 0) n_threads = int(std::thread::hardware_concurrency());
    if (n_threads <= 0) n_threads = 1;
    // work unit = 16 consecutive pixels of a row (finer than the reference's rayon row/column split needs, but it
    // keeps hundreds of host threads busy to the end)
    // (long folds — thousands of samples per pixel on a small window — get shorter units: pixel costs differ several-fold)
    const uint32_t SEG = spp >= 2048 ? 1 : (spp >= 256 ? 4 : 16), segs_per_row = (WW + SEG - 1) / SEG;
    std::atomic<uint32_t> next_unit{0};
    std::vector<Counters> counters(n_threads);
    const List& world_list = *s->world;
    auto worker = [&](int tid) {
        Counters* cnt = p->collect_counters ? &counters[tid] : nullptr;
        for (;;) {
            uint32_t unit = next_unit.fetch_add(1);
            if (unit >= WH * segs_per_row) break;
            const uint32_t wr = unit / segs_per_row, k0 = (unit % segs_per_row) * SEG, k1 = std::min(WW, k0 + SEG);
            for (uint32_t k = k0; k < k1; ++k) {
                const uint32_t i = xs ? xs[k] : x0 + k;
                const uint32_t r = xs ? ys[k] : y0 + wr; // r = 0 is the TOP row == j = height-1 (main.rs:202-205)
                const uint32_t j = H - 1 - r;
                if (world > 1 && oracle_tile_owner(i / 8, r / 8, tiles_x, world) != p->tile_rank) continue;
                uint64_t pixel = uint64_t(r) * W + i;
                V3 total(0, 0, 0), sq(0, 0, 0);
                for (uint32_t c0 = 0; c0 < spp; c0 += chunk) {
                    V3 acc(0, 0, 0);
                    uint32_t c1 = std::min(spp, c0 + chunk);
                    for (uint32_t sidx = c0; sidx < c1; ++sidx) { // main.rs:211-217
                        PathCtx ctx;
                        ctx.key = sample_key(p->seed, pixel, uint64_t(sidx) + p->sample_begin);
                        ctx.quirks = p->quirks;
                        ctx.cnt = cnt;
                        double u = (double(i) + keyed_uniform(ctx.key, ctr_of(0, SLOT_JITTER_U))) / double(W);
                        double v = (double(j) + keyed_uniform(ctx.key, ctr_of(0, SLOT_JITTER_V))) / double(H);
                        Ray ray = camera.ray(u, v, ctx.key);
                        const V3 c = color(ray, background, world_list, int(p->max_depth), ctx, p->t_min);
                        acc = acc + c;
                        if (out_var) sq = sq + c * c;
                    }
                    total = total + acc;
                }
                V3 mean = total / double(spp);
                const uint64_t o = uint64_t(wr) * WW + k; // position in the output arrays
                if (out_linear) {
                    out_linear[o * 3 + 0] = mean.x; out_linear[o * 3 + 1] = mean.y; out_linear[o * 3 + 2] = mean.z;
                }
                if (out_rgba8) {
                    out_rgba8[o * 4 + 0] = quantise(mean.x); out_rgba8[o * 4 + 1] = quantise(mean.y);
                    out_rgba8[o * 4 + 2] = quantise(mean.z); out_rgba8[o * 4 + 3] = 255;
                }
                if (out_var) {
                    const double n = double(spp), d = spp > 1 ? n - 1.0 : 1.0;
                    out_var[o * 3 + 0] = (sq.x - n * mean.x * mean.x) / d;
                    out_var[o * 3 + 1] = (sq.y - n * mean.y * mean.y) / d;
                    out_var[o * 3 + 2] = (sq.z - n * mean.z * mean.z) / d;
                }
            }
        }
    };
    std::vector<std::thread> threads;
    for (int t = 1; t < n_threads; ++t) threads.emplace_back(worker, t);
    worker(0);
    for (auto& t : threads) t.join();
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->samples = uint64_t(WW) * WH * spp;
        for (auto& c : counters) {
            stats->rays += c.rays; stats->nodes_visited += c.nodes; stats->prims_tested += c.prims;
            stats->texel_fetches += c.texels;
        }
    }
    return RTTNW_OK;
}

int rto_render(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, double* out_linear,
               uint8_t* out_rgba8, rttnw_stats* stats, int n_threads) {
    if (!p) return fail(RTTNW_ERR_INVALID, "render: bad arguments");
    return render_pixels(s, cam, p, 0, 0, p->width, p->height, nullptr, nullptr, 0, out_linear, out_rgba8, nullptr, stats, n_threads);
}
int rto_render_window(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, uint32_t x0, uint32_t y0, uint32_t x1,
                      uint32_t y1, double* out_linear, uint8_t* out_rgba8, double* out_var, rttnw_stats* stats, int n_threads) {
    return render_pixels(s, cam, p, x0, y0, x1, y1, nullptr, nullptr, 0, out_linear, out_rgba8, out_var, stats, n_threads);
}
int rto_render_pixel_list(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, const uint32_t* xs, const uint32_t* ys,
                          uint32_t n, double* out_linear, uint8_t* out_rgba8, double* out_var, rttnw_stats* stats, int n_threads) {
    if (!xs || !ys || !n) return fail(RTTNW_ERR_INVALID, "render_pixel_list: empty list");
    return render_pixels(s, cam, p, 0, 0, 0, 0, xs, ys, n, out_linear, out_rgba8, out_var, stats, n_threads);
}

// ---------------------------------------------------------------------------------------------
// Probes for known-answer tests (T0) and per-function CPU-vs-GPU vector tests.
// ---------------------------------------------------------------------------------------------

// Camera::new — out[24]: origin, lower_left_corner, horizontal, vertical, u, v, w (7x3), lens_radius,
// open_time, close_time
int rto_probe_camera(const rttnw_camera_desc* cam, double* out) {
    if (!cam || !out) return fail(RTTNW_ERR_INVALID, "probe_camera: NULL");
    Camera c(*cam);
    const V3* vs[7] = {&c.origin, &c.lower_left_corner, &c.horizontal, &c.vertical, &c.u, &c.v, &c.w};
    for (int k = 0; k < 7; ++k) { out[3 * k] = vs[k]->x; out[3 * k + 1] = vs[k]->y; out[3 * k + 2] = vs[k]->z; }
    out[21] = c.lens_radius; out[22] = c.open_time; out[23] = c.close_time;
    return RTTNW_OK;
}
// Camera::ray(s,t) with the keyed draws of (seed,pixel,sample) — out[7]: origin, direction, time
int rto_probe_camera_ray(const rttnw_camera_desc* cam, double s, double t, uint64_t seed, uint64_t pixel,
                         uint64_t sample, double* out) {
    if (!cam || !out) return fail(RTTNW_ERR_INVALID, "probe_camera_ray: NULL");
    Camera c(*cam);
    Ray r = c.ray(s, t, sample_key(seed, pixel, sample));
    out[0] = r.a.x; out[1] = r.a.y; out[2] = r.a.z; out[3] = r.b.x; out[4] = r.b.y; out[5] = r.b.z; out[6] = r.time;
    return RTTNW_OK;
}
// Hittable::hit on any hittable/list handle (id < 0: the world).  ray[7] = origin, direction, time.
// rec_out[11] = t, p(3), normal(3), u, v, front_face, material id.  Returns 1 on hit, 0 on miss.
int rto_probe_hit(rttnw_scene* s, rttnw_id id, const double* ray, double t_min, double t_max, uint64_t seed,
                  uint64_t pixel, uint64_t sample, uint32_t bounce, uint32_t quirks, double* rec_out) {
    if (!s || !ray || !rec_out) return fail(RTTNW_ERR_INVALID, "probe_hit: NULL");
    HittablePtr h = id < 0 ? HittablePtr(s->world) : get_hit(s, id);
    if (!h) return fail(RTTNW_ERR_INVALID, "probe_hit: bad id");
    Ray r{V3(ray[0], ray[1], ray[2]), V3(ray[3], ray[4], ray[5]), ray[6]};
    PathCtx ctx; ctx.key = sample_key(seed, pixel, sample); ctx.bounce = bounce; ctx.quirks = quirks;
    HitRecord rec;
    if (!h->hit(r, t_min, t_max, ctx, rec)) return 0;
    rec_out[0] = rec.t; rec_out[1] = rec.p.x; rec_out[2] = rec.p.y; rec_out[3] = rec.p.z;
    rec_out[4] = rec.normal.x; rec_out[5] = rec.normal.y; rec_out[6] = rec.normal.z;
    rec_out[7] = rec.u; rec_out[8] = rec.v; rec_out[9] = rec.front_face ? 1.0 : 0.0;
    rec_out[10] = rec.material ? double(rec.material->id) : -1.0;
    return 1;
}
// Material::scatter + emitted.  rec_in as rec_out above.  out[13] = scattered?(0/1), attenuation(3),
// scattered origin(3), direction(3), emitted(3)
int rto_probe_scatter(rttnw_scene* s, rttnw_id mat, const double* ray, const double* rec_in, uint64_t seed,
                      uint64_t pixel, uint64_t sample, uint32_t bounce, double* out) {
    auto m = get_mat(s, mat);
    if (!m || !ray || !rec_in || !out) return fail(RTTNW_ERR_INVALID, "probe_scatter: bad arguments");
    Ray r{V3(ray[0], ray[1], ray[2]), V3(ray[3], ray[4], ray[5]), ray[6]};
    HitRecord rec;
    rec.t = rec_in[0]; rec.p = V3(rec_in[1], rec_in[2], rec_in[3]);
    rec.normal = V3(rec_in[4], rec_in[5], rec_in[6]); rec.u = rec_in[7]; rec.v = rec_in[8];
    rec.front_face = rec_in[9] != 0.0; rec.material = m.get();
    PathCtx ctx; ctx.key = sample_key(seed, pixel, sample); ctx.bounce = bounce;
    V3 att; Ray sc;
    bool ok = m->scatter(r, rec, ctx, att, sc);
    V3 e = m->emitted(rec.u, rec.v, rec.p, nullptr);
    out[0] = ok ? 1.0 : 0.0; out[1] = att.x; out[2] = att.y; out[3] = att.z;
    out[4] = sc.a.x; out[5] = sc.a.y; out[6] = sc.a.z; out[7] = sc.b.x; out[8] = sc.b.y; out[9] = sc.b.z;
    out[10] = e.x; out[11] = e.y; out[12] = e.z;
    return RTTNW_OK;
}
// Texture::value
int rto_probe_tex(rttnw_scene* s, rttnw_id tex, double u, double v, const double* p, double* out) {
    Texture* t = get_tex(s, tex);
    if (!t || !p || !out) return fail(RTTNW_ERR_INVALID, "probe_tex: bad arguments");
    V3 c = t->value(u, v, V3(p[0], p[1], p[2]), nullptr);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
    return RTTNW_OK;
}
// Perlin::noise / turbulence of a noise texture
int rto_probe_perlin(rttnw_scene* s, rttnw_id tex, const double* p, uint32_t depth, double* out) {
    Texture* t = get_tex(s, tex);
    auto* nt = dynamic_cast<NoiseTexture*>(t);
    if (!nt || !p || !out) return fail(RTTNW_ERR_INVALID, "probe_perlin: not a noise texture");
    out[0] = nt->noise.noise(V3(p[0], p[1], p[2]));
    out[1] = nt->noise.turbulence(V3(p[0], p[1], p[2]), depth);
    return RTTNW_OK;
}
// Perlin tables of a noise texture: points[768] doubles, perm[768] uint32 (x, y, z)
int rto_probe_perlin_tables(rttnw_scene* s, rttnw_id tex, double* points, uint32_t* perm) {
    auto* nt = dynamic_cast<NoiseTexture*>(get_tex(s, tex));
    if (!nt) return fail(RTTNW_ERR_INVALID, "probe_perlin_tables: not a noise texture");
    for (int i = 0; i < 256; ++i) {
        if (points) { points[3 * i] = nt->noise.random_points[i].x; points[3 * i + 1] = nt->noise.random_points[i].y; points[3 * i + 2] = nt->noise.random_points[i].z; }
        if (perm) { perm[i] = nt->noise.px[i]; perm[256 + i] = nt->noise.py[i]; perm[512 + i] = nt->noise.pz[i]; }
    }
    return RTTNW_OK;
}
// Bound::hit — box[6] = min, max; returns 1/0
int rto_probe_aabb(const double* box, const double* ray, double t_min, double t_max) {
    Bound b{V3(box[0], box[1], box[2]), V3(box[3], box[4], box[5])};
    Ray r{V3(ray[0], ray[1], ray[2]), V3(ray[3], ray[4], ray[5]), 0.0};
    return b.hit(r, t_min, t_max) ? 1 : 0;
}
// bounding_box(t0,t1) of a hittable — out[6]; returns 1 when it has one
int rto_probe_bbox(rttnw_scene* s, rttnw_id id, double t0, double t1, double* out) {
    HittablePtr h = id < 0 ? HittablePtr(s->world) : get_hit(s, id);
    if (!h || !out) return fail(RTTNW_ERR_INVALID, "probe_bbox: bad id");
    Bound b;
    if (!h->bounding_box(t0, t1, b)) return 0;
    out[0] = b.min.x; out[1] = b.min.y; out[2] = b.min.z; out[3] = b.max.x; out[4] = b.max.y; out[5] = b.max.z;
    return 1;
}
double rto_probe_schlick(double cosine, double ri) { return Dielectric::schlick(cosine, ri); }
void rto_probe_refract(const double* v, const double* n, double eta, double* out) {
    V3 r = refract(V3(v[0], v[1], v[2]), V3(n[0], n[1], n[2]), eta);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void rto_probe_reflect(const double* v, const double* n, double* out) {
    V3 r = reflect(V3(v[0], v[1], v[2]), V3(n[0], n[1], n[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int rto_probe_quantise(double mean) { return int(quantise(mean)); }
double rto_probe_uniform(uint64_t seed, uint64_t pixel, uint64_t sample, uint32_t block, uint32_t slot) {
    return keyed_uniform(sample_key(seed, pixel, sample), ctr_of(block, slot));
}
uint64_t rto_probe_word(uint64_t seed, uint64_t pixel, uint64_t sample, uint32_t block, uint32_t slot) {
    return keyed_word(sample_key(seed, pixel, sample), ctr_of(block, slot));
}
// scene-construction stream: fills out[n] with next_f64() of SceneRng(seed, stream)
void rto_probe_scene_rng(uint64_t seed, uint64_t stream, uint32_t n, double* out) {
    SceneRng rng(seed, stream);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng.next_f64();
}
// color() for one sample of one pixel — out[3]
int rto_probe_sample(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, uint32_t px, uint32_t row,
                     uint32_t sample, double* out) {
    if (!s || !s->world || !cam || !p || !out) return fail(RTTNW_ERR_INVALID, "probe_sample: bad arguments");
    Camera camera(*cam);
    uint64_t pixel = uint64_t(row) * p->width + px;
    uint32_t j = p->height - 1 - row;
    PathCtx ctx; ctx.key = sample_key(p->seed, pixel, sample); ctx.quirks = p->quirks;
    double u = (double(px) + keyed_uniform(ctx.key, ctr_of(0, SLOT_JITTER_U))) / double(p->width);
    double v = (double(j) + keyed_uniform(ctx.key, ctr_of(0, SLOT_JITTER_V))) / double(p->height);
    Ray ray = camera.ray(u, v, ctx.key);
    V3 c = color(ray, V3(p->background[0], p->background[1], p->background[2]), *s->world, int(p->max_depth), ctx, p->t_min);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
    return RTTNW_OK;
}

// Per-bounce trace of one sample (the CPU twin of rttnw_debug_probe_path): out[b*12 + ..] = [0] t, [1..3] p, [4..6] normal,
// [7] material id (graph object id), [8] u, [9] v, [10] front_face, [11] scattered (1) / absorbed (0); returns #hits.
int rto_probe_path(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, uint32_t px, uint32_t row,
                   uint32_t sample, double* out, uint32_t max_out) {
    if (!s || !s->world || !cam || !p || !out) return fail(RTTNW_ERR_INVALID, "probe_path: bad arguments");
    Camera camera(*cam);
    uint64_t pixel = uint64_t(row) * p->width + px;
    uint32_t j = p->height - 1 - row;
    PathCtx ctx; ctx.key = sample_key(p->seed, pixel, sample); ctx.quirks = p->quirks;
    double u = (double(px) + keyed_uniform(ctx.key, ctr_of(0, SLOT_JITTER_U))) / double(p->width);
    double v = (double(j) + keyed_uniform(ctx.key, ctr_of(0, SLOT_JITTER_V))) / double(p->height);
    Ray ray = camera.ray(u, v, ctx.key);
    uint32_t n = 0;
    for (uint32_t depth = p->max_depth; depth > 0 && n < max_out; --depth) {
        HitRecord rec;
        if (!s->world->hit(ray, p->t_min, std::numeric_limits<double>::max(), ctx, rec)) break;
        double* o = out + size_t(n) * 12;
        o[0] = rec.t; o[1] = rec.p.x; o[2] = rec.p.y; o[3] = rec.p.z; o[4] = rec.normal.x; o[5] = rec.normal.y; o[6] = rec.normal.z;
        o[7] = double(rec.material->id);
        o[8] = rec.u; o[9] = rec.v; o[10] = rec.front_face ? 1.0 : 0.0; o[11] = 0.0;
        ++n;
        V3 att; Ray sc;
        if (!rec.material->scatter(ray, rec, ctx, att, sc)) break;
        o[11] = 1.0;
        ray = sc;
        ctx.bounce += 1;
    }
    return int(n);
}

} // extern "C"
