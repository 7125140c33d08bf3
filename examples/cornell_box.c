/*
 * cornell_box.c — a plain C caller of include/rttnw_hip.h: `scenes.rs::cornell_box` (scenes.rs:157-196) and
 * `main.rs::render` (main.rs:184-233) with the camera of main.rs:137-150, written against the C ABI alone (no C++, no
 * Python, no torch): what a cgo / JNI / Rust-FFI host does.  Writes a binary PPM (P6) of the RGBA8 framebuffer.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/cornell_box.c -Lrttnw_amd/csrc -lrttnw_hip -Wl,-rpath,$PWD/rttnw_amd/csrc -o cornell_box
 *   ./cornell_box [width [spp [out.ppm [f32|f64 [ngpu]]]]]       (defaults: 200 50 image.ppm f64 1 = BASELINE configs[0])
 *
 * Exit status: 0 ok, 2 usage, 3 the library reported an error (printed with rttnw_last_error(): e.g. no HIP device —
 * there is no CPU fallback).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rttnw_hip.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        long long rc_ = (long long)(call);                                                       \
        if (rc_ < 0) {                                                                           \
            fprintf(stderr, "%s: error %lld: %s\n", #call, rc_, rttnw_last_error());             \
            return 3;                                                                            \
        }                                                                                        \
    } while (0)

/* ids are checked once, where they are used: a negative id passed on makes the next call fail with its own message */
static rttnw_id solid_lambertian(rttnw_scene* s, double r, double g, double b) {
    return rttnw_mat_lambertian(s, rttnw_tex_solid(s, r, g, b));
}

int main(int argc, char** argv) {
    const uint32_t width = argc > 1 ? (uint32_t)atoi(argv[1]) : 200u;
    const uint32_t spp = argc > 2 ? (uint32_t)atoi(argv[2]) : 50u;
    const char* out_path = argc > 3 ? argv[3] : "image.ppm";
    const int f32 = argc > 4 && strcmp(argv[4], "f32") == 0;
    const uint32_t ngpu = argc > 5 ? (uint32_t)atoi(argv[5]) : 1u;
    if (width == 0 || spp == 0 || ngpu == 0 || ngpu > 64) {
        fprintf(stderr, "usage: %s [width [spp [out.ppm [f32|f64 [ngpu]]]]]\n", argv[0]);
        return 2;
    }
    if (rttnw_abi_version() != RTTNW_ABI_VERSION) {
        fprintf(stderr, "librttnw_hip.so has ABI %d, this program was built for %d\n", rttnw_abi_version(), RTTNW_ABI_VERSION);
        return 3;
    }

    rttnw_scene* s = NULL;
    CHECK(rttnw_scene_create(0x5eed0001ull, &s));

    /* scenes.rs:157-196 */
    const rttnw_id red = solid_lambertian(s, 0.65, 0.05, 0.05);
    const rttnw_id white = solid_lambertian(s, 0.73, 0.73, 0.73);
    const rttnw_id green = solid_lambertian(s, 0.12, 0.45, 0.15);
    const rttnw_id light = rttnw_mat_diffuse_light(s, rttnw_tex_solid(s, 15.0, 15.0, 15.0));
    const rttnw_id world = rttnw_list(s);
    CHECK(world);
    CHECK(rttnw_list_push(s, world, rttnw_rectangle(s, RTTNW_YZ, 0.0, 555.0, 0.0, 555.0, 555.0, green)));
    CHECK(rttnw_list_push(s, world, rttnw_rectangle(s, RTTNW_YZ, 0.0, 555.0, 0.0, 555.0, 0.0, red)));
    CHECK(rttnw_list_push(s, world, rttnw_rectangle(s, RTTNW_XZ, 213.0, 343.0, 227.0, 332.0, 554.0, light)));
    CHECK(rttnw_list_push(s, world, rttnw_rectangle(s, RTTNW_XZ, 0.0, 555.0, 0.0, 555.0, 555.0, white)));
    CHECK(rttnw_list_push(s, world, rttnw_rectangle(s, RTTNW_XZ, 0.0, 555.0, 0.0, 555.0, 0.0, white)));
    CHECK(rttnw_list_push(s, world, rttnw_rectangle(s, RTTNW_XY, 0.0, 555.0, 0.0, 555.0, 555.0, white)));
    {
        const double lo[3] = {0.0, 0.0, 0.0}, tall[3] = {165.0, 330.0, 165.0}, cube[3] = {165.0, 165.0, 165.0};
        const double at_tall[3] = {265.0, 0.0, 295.0}, at_cube[3] = {130.0, 0.0, 65.0};
        rttnw_id b = rttnw_cube(s, lo, tall, white);       /* cube.rotate_y(15.).translate(..) — scenes.rs:180-184 */
        b = rttnw_translate(s, rttnw_rotate_y(s, b, 15.0), at_tall);
        CHECK(rttnw_list_push(s, world, b));
        b = rttnw_cube(s, lo, cube, white);                /* scenes.rs:186-190 */
        b = rttnw_translate(s, rttnw_rotate_y(s, b, -18.0), at_cube);
        CHECK(rttnw_list_push(s, world, b));
    }
    CHECK(rttnw_scene_set_world(s, world));
    {   /* world.bounding_box(0., 1.) — hittable.rs:50,165-176: the Cornell box is [0, 555]^3 (its walls 0.0001 thick) */
        double bb[6];
        const int has = rttnw_hittable_bounds(s, world, 0.0, 1.0, bb);
        if (has != 1 || bb[0] > 0.0 || bb[3] < 555.0 || bb[1] > 0.0 || bb[4] < 555.0 || bb[2] > 0.0 || bb[5] < 555.0) {
            fprintf(stderr, "rttnw_hittable_bounds: unexpected world box (%d)\n", has);
            return 3;
        }
    }
    CHECK(rttnw_scene_commit(s)); /* flatten, build the BVHs, upload: needs a HIP device */

    /* main.rs:137-150 (lookfrom, lookat, vfov 40, aperture 0) + Camera::from's shutter 0..1 */
    rttnw_camera_desc cam;
    memset(&cam, 0, sizeof cam);
    cam.lookfrom[0] = 278.0; cam.lookfrom[1] = 278.0; cam.lookfrom[2] = -800.0;
    cam.lookat[0] = 278.0; cam.lookat[1] = 278.0; cam.lookat[2] = 0.0;
    cam.view_up[1] = 1.0;
    cam.vertical_fov = 40.0;
    cam.aspect_ratio = 1.0;
    cam.aperture = 0.0;
    cam.focus_distance = 10.0;
    cam.open_time = 0.0;
    cam.close_time = 1.0;

    rttnw_params p;
    memset(&p, 0, sizeof p);
    p.width = width; p.height = width;
    p.spp = spp;
    p.max_depth = 50;               /* main.rs:216 */
    p.t_min = 0.001;                /* main.rs:33 */
    p.background[0] = p.background[1] = p.background[2] = 0.0; /* main.rs:143 */
    p.seed = 1;
    p.precision = f32 ? RTTNW_F32 : RTTNW_F64;
    p.quirks = RTTNW_QUIRKS_REFERENCE;
    p.tile_world = 1;

    uint8_t* rgba = (uint8_t*)malloc((size_t)width * width * 4);
    rttnw_stats stats[64];
    memset(stats, 0, sizeof stats);
    if (!rgba) return 3;
    if (ngpu == 1) {
        CHECK(rttnw_render(s, &cam, &p, NULL, rgba, &stats[0]));
    } else { /* the GPUs of a node in one call; with fewer devices than ranks, devices are reused (logical ranks) */
        int32_t dev[64];
        const int nd = rttnw_device_count();
        uint32_t r;
        CHECK(nd - 1);
        for (r = 0; r < ngpu; ++r) dev[r] = (int32_t)(r % (uint32_t)nd);
        CHECK(rttnw_render_multi(s, &cam, &p, ngpu, dev, NULL, rgba, stats));
    }

    FILE* f = fopen(out_path, "wb");
    if (!f) { perror(out_path); return 3; }
    fprintf(f, "P6\n%u %u\n255\n", width, width);
    {
        size_t i, n = (size_t)width * width;
        unsigned long long sum[3] = {0, 0, 0};
        for (i = 0; i < n; ++i) {
            fwrite(rgba + 4 * i, 1, 3, f); /* top row first, like main.rs:202-205 */
            sum[0] += rgba[4 * i]; sum[1] += rgba[4 * i + 1]; sum[2] += rgba[4 * i + 2];
        }
        fclose(f);
        printf("%s: %ux%u spp %u %s on %u GPU(s): %.1f ms device time, mean RGB8 %.3f %.3f %.3f\n", out_path, width, width, spp,
               f32 ? "f32" : "f64", ngpu, stats[0].kernel_ms, (double)sum[0] / n, (double)sum[1] / n, (double)sum[2] / n);
    }
    free(rgba);
    rttnw_scene_destroy(s);
    return 0;
}
