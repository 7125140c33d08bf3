// Host-side scene catalogue — the caller side of the boundary, mirroring the reference's
// src/scenes.rs (what gets built) and src/main.rs:66-183 (camera / size per scene).
// Everything goes through the rttnw_builder_api table; this file knows nothing about HIP.
#include "../../include/rttnw_scenes.h"

#include <cmath>
#include <cstring>
#include <string>

namespace {

// Scene stream (DESIGN.md "RNG"): SplitMix64 walk from a (seed, stream)-derived start.
struct SceneRng {
    static constexpr uint64_t GAMMA = 0x9E3779B97F4A7C15ull;
    uint64_t s;
    static uint64_t mix64(uint64_t z) {
        z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
        z ^= z >> 27; z *= 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    SceneRng(uint64_t seed, uint64_t stream)
        : s(mix64(seed + GAMMA) ^ mix64((stream + 1) * 0xD1B54A32D192ED03ull)) {}
    double gen() { s += GAMMA; return double(mix64(s) >> 11) * (1.0 / 9007199254740992.0); } // rng.gen::<f64>()
    double gen_range(double a, double b) { return a + (b - a) * gen(); }                     // rng.gen_range(a..b)
};

struct Builder {
    const rttnw_builder_api* api;
    rttnw_scene* sc;
    bool ok = true;
    rttnw_id chk(rttnw_id id) { if (id < 0) ok = false; return id; }
    void chk_rc(int rc) { if (rc != RTTNW_OK) ok = false; }

    rttnw_id solid(double r, double g, double b) { return chk(api->tex_solid(sc, r, g, b)); }
    rttnw_id lambertian_rgb(double r, double g, double b) { return chk(api->mat_lambertian(sc, solid(r, g, b))); }
    rttnw_id lambertian(rttnw_id tex) { return chk(api->mat_lambertian(sc, tex)); }
    rttnw_id metal(double r, double g, double b, double fuzz) { return chk(api->mat_metal(sc, r, g, b, fuzz)); }
    rttnw_id dielectric(double ri) { return chk(api->mat_dielectric(sc, ri)); }
    rttnw_id light(double e) { return chk(api->mat_diffuse_light(sc, solid(e, e, e))); }
    rttnw_id sphere(double x, double y, double z, double r, rttnw_id m) {
        const double c[3] = {x, y, z};
        return chk(api->sphere(sc, c, r, m));
    }
    rttnw_id moving_sphere(const double c0[3], const double c1[3], double t0, double t1, double r, rttnw_id m) {
        return chk(api->moving_sphere(sc, c0, c1, t0, t1, r, m));
    }
    rttnw_id rect(int plane, double a0, double a1, double b0, double b1, double k, rttnw_id m) {
        return chk(api->rectangle(sc, plane, a0, a1, b0, b1, k, m));
    }
    rttnw_id cube(double x0, double y0, double z0, double x1, double y1, double z1, rttnw_id m) {
        const double mn[3] = {x0, y0, z0}, mx[3] = {x1, y1, z1};
        return chk(api->cube(sc, mn, mx, m));
    }
    rttnw_id list() { return chk(api->list(sc)); }
    void push(rttnw_id l, rttnw_id item) { chk_rc(api->list_push(sc, l, item)); }
    rttnw_id bvh(rttnw_id l) { return chk(api->bvh_tree(sc, l)); }
    rttnw_id rotate_y(rttnw_id item, double deg) { return chk(api->rotate_y(sc, item, deg)); }
    rttnw_id translate(rttnw_id item, double x, double y, double z) {
        const double o[3] = {x, y, z};
        return chk(api->translate(sc, item, o));
    }
    rttnw_id medium(rttnw_id boundary, double density, rttnw_id tex) {
        return chk(api->constant_medium(sc, boundary, density, tex));
    }
};

struct EarthImage { const uint8_t* rgba; uint32_t w, h; };

// scenes.rs:11-88
rttnw_id random_scene(Builder& b, SceneRng& rng) {
    rttnw_id world = b.list();
    rttnw_id checker = b.chk(b.api->tex_checker(b.sc, b.solid(0.2, 0.3, 0.1), b.solid(0.9, 0.9, 0.9)));
    b.push(world, b.sphere(0.0, -1000.0, 0.0, 1000.0, b.lambertian(checker)));
    for (int a = -11; a < 11; ++a) {
        for (int c = -11; c < 11; ++c) {
            double choose_mat = rng.gen();
            double cx = double(a) + 0.9 + rng.gen();
            double cy = 0.2;
            double cz = double(c) + 0.9 + rng.gen();
            double dx = cx - 4.0, dy = cy - 0.2, dz = cz - 0.0;
            if (std::sqrt(dx * dx + dy * dy + dz * dz) > 0.9) {
                if (choose_mat < 0.8) { // diffuse, bouncing upwards during the shutter interval
                    double lift = rng.gen_range(0.0, 0.5);
                    const double c0[3] = {cx, cy, cz}, c1[3] = {cx + 0.0, cy + lift, cz + 0.0};
                    double r = rng.gen() * rng.gen(), g = rng.gen() * rng.gen(), bl = rng.gen() * rng.gen();
                    b.push(world, b.moving_sphere(c0, c1, 0., 1., 0.2, b.lambertian_rgb(r, g, bl)));
                } else if (choose_mat < 0.95) { // metal
                    double r = 0.5 * (1.0 - rng.gen()), g = 0.5 * (1.0 - rng.gen()), bl = 0.5 * (1.0 - rng.gen());
                    double fuzz = 0.5 * rng.gen();
                    b.push(world, b.sphere(cx, cy, cz, 0.2, b.metal(r, g, bl, fuzz)));
                } else { // glass
                    b.push(world, b.sphere(cx, cy, cz, 0.2, b.dielectric(1.5)));
                }
            }
        }
    }
    b.push(world, b.sphere(0.0, 1.0, 0.0, 1.0, b.dielectric(1.5)));
    b.push(world, b.sphere(-4.0, 1.0, 0.0, 1.0, b.lambertian_rgb(0.4, 0.2, 0.1)));
    b.push(world, b.sphere(4.0, 1.0, 0.0, 1.0, b.metal(0.7, 0.6, 0.5, 0.0)));
    return world;
}

// scenes.rs:90-108
rttnw_id two_spheres(Builder& b) {
    rttnw_id world = b.list();
    rttnw_id checker = b.chk(b.api->tex_checker(b.sc, b.solid(0.2, 0.3, 0.1), b.solid(0.9, 0.9, 0.9)));
    rttnw_id m = b.lambertian(checker); // one Arc<CheckerTexture> shared by both spheres
    b.push(world, b.sphere(0.0, -10.0, 0.0, 10.0, m));
    b.push(world, b.sphere(0.0, 10.0, 0.0, 10.0, b.lambertian(checker)));
    return world;
}

// scenes.rs:110-125
rttnw_id two_perlin_spheres(Builder& b) {
    rttnw_id world = b.list();
    rttnw_id perlin = b.chk(b.api->tex_noise(b.sc, 4.));
    b.push(world, b.sphere(0.0, -1000.0, 0.0, 1000.0, b.lambertian(perlin)));
    b.push(world, b.sphere(0.0, 2.0, 0.0, 2.0, b.lambertian(perlin)));
    return world;
}

// scenes.rs:127-136
rttnw_id earth(Builder& b, const EarthImage& img) {
    rttnw_id world = b.list();
    rttnw_id tex = b.chk(b.api->tex_image_rgba8(b.sc, img.rgba, img.w, img.h));
    b.push(world, b.sphere(0.0, 0.0, 0.0, 2., b.lambertian(tex)));
    return world;
}

// scenes.rs:138-155
rttnw_id simple_light(Builder& b) {
    rttnw_id world = two_perlin_spheres(b);
    b.push(world, b.rect(RTTNW_XY, 3., 5., 1., 3., -2.0, b.light(4.)));
    return world;
}

// Walls + ceiling light shared by the three Cornell variants — scenes.rs:157-173, :198-211
rttnw_id cornell_shell(Builder& b, double emit, double lx0, double lx1, double lz0, double lz1, rttnw_id* white_out) {
    rttnw_id world = b.list();
    rttnw_id red = b.lambertian_rgb(0.65, 0.05, 0.05);
    rttnw_id white = b.lambertian_rgb(0.73, 0.73, 0.73);
    rttnw_id green = b.lambertian_rgb(0.12, 0.45, 0.15);
    rttnw_id lamp = b.light(emit);
    b.push(world, b.rect(RTTNW_YZ, 0., 555., 0., 555., 555., green));
    b.push(world, b.rect(RTTNW_YZ, 0., 555., 0., 555., 0., red));
    b.push(world, b.rect(RTTNW_XZ, lx0, lx1, lz0, lz1, 554., lamp));
    b.push(world, b.rect(RTTNW_XZ, 0., 555., 0., 555., 555., white));
    b.push(world, b.rect(RTTNW_XZ, 0., 555., 0., 555., 0., white));
    b.push(world, b.rect(RTTNW_XY, 0., 555., 0., 555., 555., white));
    if (white_out) *white_out = white;
    return world;
}
rttnw_id empty_cornell_box(Builder& b) { return cornell_shell(b, 15., 213., 343., 227., 332., nullptr); }

// The two rotated, translated blocks — scenes.rs:180-193 and :213-222
void cornell_blocks(Builder& b, rttnw_id white, rttnw_id* tall, rttnw_id* small) {
    *tall = b.translate(b.rotate_y(b.cube(0., 0., 0., 165., 330., 165., white), 15.), 265., 0., 295.);
    *small = b.translate(b.rotate_y(b.cube(0., 0., 0., 165., 165., 165., white), -18.), 130., 0., 65.);
}

// scenes.rs:175-196
rttnw_id cornell_box(Builder& b) {
    rttnw_id world = empty_cornell_box(b);
    rttnw_id white = b.lambertian_rgb(0.73, 0.73, 0.73); // a second `white` Arc, scenes.rs:178
    rttnw_id tall, small;
    cornell_blocks(b, white, &tall, &small);
    b.push(world, tall);
    b.push(world, small);
    return world;
}

// scenes.rs:198-236
rttnw_id smoke_cornell_box(Builder& b) {
    rttnw_id white;
    rttnw_id world = cornell_shell(b, 7., 113., 443., 127., 432., &white);
    rttnw_id tall, small;
    cornell_blocks(b, white, &tall, &small);
    b.push(world, b.medium(tall, 0.01, b.solid(0., 0., 0.)));
    b.push(world, b.medium(small, 0.01, b.solid(1., 1., 1.)));
    return world;
}

// scenes.rs:238-334
rttnw_id final_scene(Builder& b, SceneRng& rng, const EarthImage& img, uint32_t ns) {
    rttnw_id boxes = b.list();
    rttnw_id ground = b.lambertian_rgb(0.48, 0.83, 0.53);
    const int boxes_per_side = 20;
    for (int i = 0; i < boxes_per_side; ++i) {
        for (int j = 0; j < boxes_per_side; ++j) {
            const double w = 100.;
            double x0 = -1000. + double(i) * w, z0 = -1000. + double(j) * w;
            double y1 = rng.gen_range(1., 101.);
            b.push(boxes, b.cube(x0, 0., z0, x0 + w, y1, z0 + w, ground));
        }
    }
    rttnw_id world = b.list();
    b.push(world, b.bvh(boxes));
    b.push(world, b.rect(RTTNW_XZ, 123., 423., 147., 412., 554., b.light(7.)));

    const double c1[3] = {400., 400., 400.}, c2[3] = {400. + 30., 400. + 0., 400. + 0.};
    b.push(world, b.moving_sphere(c1, c2, 0., 1., 50., b.lambertian_rgb(0.7, 0.3, 0.1)));
    b.push(world, b.sphere(260., 150., 45., 50.0, b.dielectric(1.5)));
    b.push(world, b.sphere(0., 150., 45., 50.0, b.metal(0.8, 0.8, 0.9, 1.)));

    // glass ball with a blue participating medium inside (boundary pushed AND used as boundary)
    rttnw_id boundary = b.sphere(360., 150., 145., 70., b.dielectric(1.5));
    b.push(world, boundary);
    b.push(world, b.medium(boundary, 0.2, b.solid(0.2, 0.4, 0.9)));
    // thin white fog filling the whole scene
    b.push(world, b.medium(b.sphere(0., 0., 0., 5000., b.dielectric(1.5)), 0.0001, b.solid(1., 1., 1.)));

    rttnw_id earth_tex = b.chk(b.api->tex_image_rgba8(b.sc, img.rgba, img.w, img.h));
    b.push(world, b.sphere(400., 200., 400., 100., b.lambertian(earth_tex)));
    rttnw_id noise = b.chk(b.api->tex_noise(b.sc, 0.1));
    b.push(world, b.sphere(220., 280., 300., 80.0, b.lambertian(noise)));

    rttnw_id cluster = b.list();
    rttnw_id white = b.lambertian_rgb(0.73, 0.73, 0.73);
    for (uint32_t k = 0; k < ns; ++k) {
        double x = rng.gen_range(0., 165.), y = rng.gen_range(0., 165.), z = rng.gen_range(0., 165.);
        b.push(cluster, b.sphere(x, y, z, 10., white));
    }
    b.push(world, b.translate(b.rotate_y(b.bvh(cluster), 15.), -100., 270., 395.));
    return world;
}

// BASELINE.md config 5 (build-defined; the reference has no such scene and could not build it):
// n spheres r=1.5, centres U in x,z in [-400,400), y in [0,800); 80/15/5 % Lambertian/Metal/
// Dielectric with the albedo recipes of scenes.rs:39-43,50-58,65; one XZ light.
rttnw_id spheres_1m(Builder& b, SceneRng& rng, uint32_t n) {
    rttnw_id cloud = b.list();
    for (uint32_t k = 0; k < n; ++k) {
        double x = rng.gen_range(-400., 400.), y = rng.gen_range(0., 800.), z = rng.gen_range(-400., 400.);
        double choose_mat = rng.gen();
        rttnw_id m;
        if (choose_mat < 0.8) {
            double r = rng.gen() * rng.gen(), g = rng.gen() * rng.gen(), bl = rng.gen() * rng.gen();
            m = b.lambertian_rgb(r, g, bl);
        } else if (choose_mat < 0.95) {
            double r = 0.5 * (1.0 - rng.gen()), g = 0.5 * (1.0 - rng.gen()), bl = 0.5 * (1.0 - rng.gen());
            double fuzz = 0.5 * rng.gen();
            m = b.metal(r, g, bl, fuzz);
        } else {
            m = b.dielectric(1.5);
        }
        b.push(cloud, b.sphere(x, y, z, 1.5, m));
    }
    rttnw_id world = b.list();
    b.push(world, b.bvh(cloud));
    b.push(world, b.rect(RTTNW_XZ, -200., 200., -200., 200., 1000., b.light(7.)));
    return world;
}

struct TableEntry {
    const char* name;
    uint32_t number;
    double bg[3], from[3], at[3], vfov, aperture;
    uint32_t width, height, spp;
};
// main.rs:66-183 (+ defaults render(400, 16/9, 100) main.rs:255 and height = (400/(16/9)) as u32 = 225)
const TableEntry kTable[] = {
    {"random_scene", 1, {0.7, 0.8, 1.}, {13., 2., 3.}, {0., 0., 0.}, 20., 0.1, 400, 225, 100},
    {"two_spheres", 2, {0.7, 0.8, 1.}, {13., 2., 3.}, {0., 0., 0.}, 20., 0., 400, 225, 100},
    {"two_perlin_spheres", 3, {0.7, 0.8, 1.}, {13., 2., 3.}, {0., 0., 0.}, 20., 0., 400, 225, 100},
    {"earth", 4, {0.7, 0.8, 1.}, {13., 2., 3.}, {0., 0., 0.}, 20., 0., 400, 225, 100},
    {"simple_light", 5, {0., 0., 0.}, {26., 3., 6.}, {0., 2., 0.}, 20., 0., 400, 225, 400},
    {"empty_cornell_box", 6, {0., 0., 0.}, {278., 278., -800.}, {278., 278., 0.}, 40., 0., 600, 600, 200},
    {"cornell_box", 7, {0., 0., 0.}, {278., 278., -800.}, {278., 278., 0.}, 40., 0., 600, 600, 200},
    {"smoke_cornell_box", 8, {0., 0., 0.}, {278., 278., -800.}, {278., 278., 0.}, 40., 0., 600, 600, 200},
    {"final_scene", 9, {0., 0., 0.}, {478., 278., -600.}, {278., 278., 0.}, 40., 0., 800, 800, 10000},
    // build-defined
    {"spheres_1m", 0, {0.7, 0.8, 1.}, {0., 400., -1600.}, {0., 400., 0.}, 40., 0., 1024, 1024, 256},
};

} // namespace

extern "C" {

const char* rttnw_scenes_name(uint32_t n) {
    for (const auto& e : kTable)
        if (e.number == n && n != 0) return e.name;
    return nullptr;
}

void rttnw_scenes_rng_f64(uint64_t seed, uint64_t stream, uint32_t n, double* out) {
    SceneRng rng(seed, stream);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng.gen();
}

int rttnw_scenes_build(const rttnw_builder_api* api, rttnw_scene* scene, const char* name, uint64_t scene_seed,
                       const uint8_t* earth_rgba, uint32_t earth_w, uint32_t earth_h, uint32_t param,
                       rttnw_scene_setup* setup_out) {
    if (!api || !scene || !name) return RTTNW_ERR_INVALID;
    const TableEntry* entry = nullptr;
    for (const auto& e : kTable)
        if (std::strcmp(e.name, name) == 0) entry = &e;
    if (!entry) return RTTNW_ERR_INVALID; // "There is no scene {}" — main.rs:179-182

    Builder b{api, scene};
    SceneRng rng(scene_seed, 1);
    EarthImage img{earth_rgba, earth_w, earth_h};
    const std::string n(name);
    rttnw_id world;
    if (n == "random_scene") world = random_scene(b, rng);
    else if (n == "two_spheres") world = two_spheres(b);
    else if (n == "two_perlin_spheres") world = two_perlin_spheres(b);
    else if (n == "earth") world = earth(b, img);
    else if (n == "simple_light") world = simple_light(b);
    else if (n == "empty_cornell_box") world = empty_cornell_box(b);
    else if (n == "cornell_box") world = cornell_box(b);
    else if (n == "smoke_cornell_box") world = smoke_cornell_box(b);
    else if (n == "final_scene") world = final_scene(b, rng, img, param ? param : 1000u);
    else world = spheres_1m(b, rng, param ? param : 1000000u);
    if (!b.ok) return RTTNW_ERR_INVALID;
    int rc = api->scene_set_world(scene, world);
    if (rc != RTTNW_OK) return rc;
    rc = api->scene_commit(scene);
    if (rc != RTTNW_OK) return rc;

    if (setup_out) {
        rttnw_scene_setup& s = *setup_out;
        std::memset(&s, 0, sizeof(s));
        for (int k = 0; k < 3; ++k) {
            s.camera.lookfrom[k] = entry->from[k];
            s.camera.lookat[k] = entry->at[k];
            s.background[k] = entry->bg[k];
        }
        s.camera.view_up[1] = 1.0;       // main.rs:185
        s.camera.vertical_fov = entry->vfov;
        s.camera.aspect_ratio = double(entry->width) / double(entry->height);
        s.camera.aperture = entry->aperture;
        s.camera.focus_distance = 10.0;  // main.rs:186
        s.camera.open_time = 0.0;        // main.rs:195-196
        s.camera.close_time = 1.0;
        s.width = entry->width; s.height = entry->height; s.spp = entry->spp;
        s.scene_number = entry->number;
    }
    return RTTNW_OK;
}

} // extern "C"
