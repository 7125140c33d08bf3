/*
 * rttnw_scenes.h — host-side scene catalogue: the C++ mirror of the reference's `src/scenes.rs`
 * (nine `fn xxx() -> List` builders) and of the per-scene camera / size table in
 * `src/main.rs:66-183`, written against the entry-point table of include/rttnw_hip.h so that it
 * is a CALLER of the boundary (like scenes.rs is of math/), not part of it.
 *
 * Scene-construction randomness the reference takes from `thread_rng()` inside scenes.rs
 * (scenes.rs:12,242,321) comes from the documented scene stream `SceneRng(scene_seed, 1)`
 * (DESIGN.md "RNG"), consumed in the reference's program order.
 */
#ifndef RTTNW_SCENES_H
#define RTTNW_SCENES_H

#include "rttnw_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Per-scene values of main.rs:66-183 + the defaults of main.rs:255 / :184-197. */
typedef struct rttnw_scene_setup {
    rttnw_camera_desc camera; /* view_up (0,1,0), focus 10, shutter 0..1 — main.rs:185-196 */
    double background[3];
    uint32_t width, height;   /* the reference's default size for this scene */
    uint32_t spp;             /* the reference's default `samples` */
    uint32_t scene_number;    /* CLI number 1..9 (main.rs:241-249); 0 for build-defined scenes */
} rttnw_scene_setup;

/*
 * Build scene `name` into `scene` (created by api->scene_create with the same scene_seed), set
 * its world and commit it.  Names: "random_scene", "two_spheres", "two_perlin_spheres", "earth",
 * "simple_light", "empty_cornell_box", "cornell_box", "smoke_cornell_box", "final_scene"
 * (scenes.rs) and the build-defined stress scene "spheres_1m" (BASELINE.md config 5).
 *
 * earth_rgba/earth_w/earth_h: decoded `assets/earth.png` (scenes.rs:129,303); NULL reproduces
 * the reference's missing-file behaviour (cyan, texture.rs:102-105).
 * param: scene-specific size override (spheres_1m: number of spheres; final_scene: number of
 * cluster spheres `ns`, scenes.rs:318); 0 = the reference / BASELINE value.
 * setup_out (optional) receives the camera/size table entry with aspect_ratio = width/height.
 */
int rttnw_scenes_build(const rttnw_builder_api* api, rttnw_scene* scene, const char* name,
                       uint64_t scene_seed, const uint8_t* earth_rgba, uint32_t earth_w,
                       uint32_t earth_h, uint32_t param, rttnw_scene_setup* setup_out);

/* Name of CLI scene number 1..9 (main.rs:241-249), NULL otherwise. */
const char* rttnw_scenes_name(uint32_t scene_number);

/* Scene stream draw, exported so tests can pin the generator: fills out[n] with the first n
 * `next_f64()` of SceneRng(scene_seed, stream). */
void rttnw_scenes_rng_f64(uint64_t scene_seed, uint64_t stream, uint32_t n, double* out);

#ifdef __cplusplus
}
#endif
#endif
