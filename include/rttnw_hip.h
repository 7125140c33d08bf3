/*
 * rttnw_hip.h — C ABI of the MI355X-native path tracer that drops in behind
 * luliic2/rttnw's `scenes.rs` / `main.rs::render`.
 *
 * The reference has no FFI; the seam that exists is the Rust-level surface between
 * `src/scenes.rs` + `src/main.rs` (callers) and `src/math/` (the tracing core).  Every entry
 * point below replaces one constructor / method of that surface; the reference item is cited
 * next to it as `file:line` relative to the reference repo root.  `dyn Hittable` objects expose
 * only `hit`/`bounding_box`, so they cannot be lowered after the fact: a host describes the scene
 * through these calls instead of (or next to) building trait objects.  INTEGRATION.md shows the
 * Rust `extern "C"` block and the `scenes.rs`-shaped wrapper a maintainer would add.
 *
 * Conventions
 *   - plain C, POD only, no callbacks; doubles on the boundary (the reference is f64 throughout,
 *     src/math/vec3.rs:12); the library narrows to f32 itself when `precision == RTTNW_F32`.
 *   - objects (textures, materials, hittables) are `rttnw_id` handles owned by their scene; a
 *     negative id is an error code.  Handles may be shared (Arc semantics, e.g. one material on
 *     many spheres, scenes.rs:317-324).
 *   - functions returning `int` return RTTNW_OK (0) or a negative RTTNW_ERR_*;
 *     `rttnw_last_error()` returns a thread-local message for the last failure.
 *   - a scene is mutable until `rttnw_scene_commit`, immutable afterwards (the reference shares
 *     `&world` immutably across rayon workers, main.rs:216); one render in flight per scene.
 *     One exception, inside the library: the trees are built for the shutter interval [0, 1]
 *     (`BvhTree::from`, hittable.rs:256); the first render whose camera shutter reaches outside
 *     it rebuilds them for the wider interval (`BvhTree::from_time`, hittable.rs:261) before it
 *     enqueues anything — under a lock, device copies re-uploaded on use, and
 *     `rttnw_scene_build_info` reports the rebuilt trees from then on.
 *   - there is NO CPU fallback: every render entry point fails with RTTNW_ERR_HIP when no gfx950
 *     device is usable.
 */
#ifndef RTTNW_HIP_H
#define RTTNW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTTNW_ABI_VERSION 3 /* 2: 4-wide node records (n_nodes, debug_scene_nodes4), rttnw_render_multi
                             * 3: RTTNW_F64_STRICT, rttnw_shutdown, RTTNW_BVH_AUTO (the default builder), rttnw_stats.reserved is a bit mask
                             *    (below), validate() rejects t_min < 0 */

typedef struct rttnw_scene rttnw_scene; /* opaque */
typedef int32_t rttnw_id;

enum rttnw_status {
    RTTNW_OK = 0,
    RTTNW_ERR_INVALID = -1,     /* bad argument / unknown id / wrong object kind */
    RTTNW_ERR_STATE = -2,       /* call not allowed in this scene state (e.g. edit after commit) */
    RTTNW_ERR_UNSUPPORTED = -3, /* scene graph shape the lowering does not handle */
    RTTNW_ERR_HIP = -4,         /* HIP runtime error or no usable device */
    RTTNW_ERR_NOMEM = -5
};

/* Plane selector of `Rectangle<M,P>`: (axis0, axis1, k) — hittable.rs:450-488. */
enum rttnw_plane { RTTNW_XY = 0, RTTNW_XZ = 1, RTTNW_YZ = 2 };

/* Arithmetic type of the render kernels.
 *   RTTNW_F64  the reference's type (vec3.rs:12) and the PARITY mode: images equal the f64 restatement of the reference
 *              to <= 1e-9 per channel (RGBA8 identical at 800x800 spp 1000), every bounce's hit record to 1e-9.
 *   RTTNW_F32  a THROUGHPUT mode (about 1.5x the rate), NOT a peer precision: equal seeds share every random decision
 *              with f64 up to rounding, the estimator is unbiased (every pixel within 6 sigma / sqrt(spp) + 1/256 of
 *              the f64 image, crop means within 0.5 %), but on scenes with small specular / refractive / fuzzy spheres
 *              (final_scene: the r = 10 cluster, glass, fuzz-1 metal) an f32 path and its f64 twin diverge after a few
 *              bounces, so RGBA8 agrees within 1 LSB on 96.7 % of the pixels at spp 1000 there (cornell_box: 99 %) —
 *              outside the >= 99 % of the parity tier (tests/test_gpu_parity.py::test_T2_at_baseline_size).
 *   RTTNW_F64_STRICT  f64 with NOTHING contracted and every quotient an IEEE division — the operations of the reference's Rust in
 *              its order (rustc fuses no multiply-add): every path takes the decisions of the CPU reference bit for bit, where
 *              RTTNW_F64 (built with fused multiply-adds and shared reciprocals) agrees to rounding only — which a scene that
 *              amplifies rounding (config 5: a million small spheres, ~100x per bounce) turns into different paths after a few
 *              bounces.  It also tests every object in the frame the reference tests it in: a scene whose default lowering holds
 *              world-space copies of transformed groups' spheres is lowered a second time at its first strict render — the copies'
 *              world-space BOXES stay in the top tree (culling never shapes a result), the sphere test itself is made through the
 *              group's wrappers as in hittable.rs:599-606,686-699 (final_scene: every pixel within 1e-12 of the CPU reference at
 *              800x800 spp 5000, RGBA8 identical).  Same buffers as RTTNW_F64 (doubles); 5-10 % slower on cornell_box, 15-18 % on
 *              final_scene (what it costs, measured: IEEE quotients 6 %, no contraction 3 %, registers the rest), nothing on config 5. */
enum rttnw_precision { RTTNW_F64 = 0, RTTNW_F32 = 1, RTTNW_F64_STRICT = 2 };

/* Bit flags for `rttnw_params.quirks`. */
#define RTTNW_QUIRK_YROTATE_BACKROT 1u /* Q1: hittable.rs:700-705 reuses the overwritten x */
#define RTTNW_QUIRKS_REFERENCE RTTNW_QUIRK_YROTATE_BACKROT

/* ---------------------------------------------------------------- lifecycle ---------------- */

/* `scene_seed` feeds the library-side scene randomness the reference draws from `thread_rng()`
 * while constructing objects: Perlin tables (noise.rs:15-29,40-47).  Generator spec: DESIGN.md. */
int rttnw_scene_create(uint64_t scene_seed, rttnw_scene** out);
void rttnw_scene_destroy(rttnw_scene* scene);

/* ---------------------------------------------------------------- textures (texture.rs) ---- */

/* `impl Texture for Vec3f<Color>` — texture.rs:9-13 */
rttnw_id rttnw_tex_solid(rttnw_scene* s, double r, double g, double b);
/* `CheckerTexture { odd, even }` — texture.rs:15-30 */
rttnw_id rttnw_tex_checker(rttnw_scene* s, rttnw_id odd, rttnw_id even);
/* `NoiseTexture::scaled(scale)` (owns a fresh `Perlin::new()`) — texture.rs:45-59, noise.rs:40-47 */
rttnw_id rttnw_tex_noise(rttnw_scene* s, double scale);
/* `ImageTexture::new(path)` — texture.rs:64-107.  The host decodes the file; the library copies
 * `w*h*4` bytes (RGBA8, row-major, top row first; alpha ignored).  `rgba == NULL` reproduces the
 * load-failure behaviour: constant cyan (texture.rs:102-105). */
rttnw_id rttnw_tex_image_rgba8(rttnw_scene* s, const uint8_t* rgba, uint32_t w, uint32_t h);

/* ---------------------------------------------------------------- materials (material.rs) -- */

/* `Lambertian::arc/boxed(texture_or_colour)` — material.rs:25-100 */
rttnw_id rttnw_mat_lambertian(rttnw_scene* s, rttnw_id tex);
/* `Metal::arc(albedo, fuzz)`; fuzz is clamped to <= 1 like material.rs:114,122,129 — :103-149 */
rttnw_id rttnw_mat_metal(rttnw_scene* s, double r, double g, double b, double fuzz);
/* `Dielectric::arc(refraction_index)` — material.rs:152-204 */
rttnw_id rttnw_mat_dielectric(rttnw_scene* s, double refraction_index);
/* `DiffuseLight::arc(texture)` — material.rs:206-250 */
rttnw_id rttnw_mat_diffuse_light(rttnw_scene* s, rttnw_id tex);
/* `Isotropic { albedo }` — material.rs:252-266 (also created implicitly by constant_medium) */
rttnw_id rttnw_mat_isotropic(rttnw_scene* s, rttnw_id tex);

/* ---------------------------------------------------------------- hittables (hittable.rs) -- */

/* `Sphere { center, radius, material }` — hittable.rs:69-131 */
rttnw_id rttnw_sphere(rttnw_scene* s, const double center[3], double radius, rttnw_id mat);
/* `MovingSphere { center: c0..c1, time: t0..t1, radius, material }` — hittable.rs:179-245 */
rttnw_id rttnw_moving_sphere(rttnw_scene* s, const double center0[3], const double center1[3],
                             double time0, double time1, double radius, rttnw_id mat);
/* `XY|XZ|YZ::rectangle(material, a0..a1, b0..b1, k)` — hittable.rs:401-411,434-547 */
rttnw_id rttnw_rectangle(rttnw_scene* s, int plane, double a0, double a1, double b0, double b1,
                         double k, rttnw_id mat);
/* `Cube::new(box_min, box_max, material)` (six rectangles in a List) — hittable.rs:549-592 */
rttnw_id rttnw_cube(rttnw_scene* s, const double box_min[3], const double box_max[3], rttnw_id mat);
/* `List::new()` / `List::push(item)` — hittable.rs:134-177 */
rttnw_id rttnw_list(rttnw_scene* s);
int rttnw_list_push(rttnw_scene* s, rttnw_id list, rttnw_id item);
/* `BvhTree::from(list)` — hittable.rs:248-373.  Closest-hit results do not depend on the tree's
 * topology, so this is a grouping hint: the library builds its own flat BVH. */
rttnw_id rttnw_bvh_tree(rttnw_scene* s, rttnw_id list);
/* `Hittable::translate(offset)` — hittable.rs:51-59,594-629 */
rttnw_id rttnw_translate(rttnw_scene* s, rttnw_id item, const double offset[3]);
/* `Hittable::rotate_y(angle_degrees)` — hittable.rs:60-65,631-722 */
rttnw_id rttnw_rotate_y(rttnw_scene* s, rttnw_id item, double angle_degrees);
/* `ConstantMedium::new(boundary, density, phase_texture)` — hittable.rs:724-801 */
rttnw_id rttnw_constant_medium(rttnw_scene* s, rttnw_id boundary, double density, rttnw_id tex);

/* `Hittable::bounding_box(initial_time, final_time) -> Option<Bound>` — hittable.rs:50, the trait's second method, for any hittable id of the
 * scene's graph (before or after commit): Sphere :125-130, List :165-176, MovingSphere :233-244, BvhTree :370-372 (the bound stored by
 * `BvhTree::from` = over times 0..1, whatever is asked), Rectangle :532-546 (k -+ 0.0001), Cube :585-591, Translate :619-628, YRotate :719-721
 * (the item's box over 0..1, turned about y), ConstantMedium :798-800.  Returns 1 and writes min.xyz, max.xyz to out_min_max — 0 for `None`
 * (an empty List, a List with a member that has none) — or a negative rttnw_status.  One deliberate difference: YRotate's box is the CORRECT
 * rotation of the item's eight corners; the reference's (:661-662) rotates z with the x it has just overwritten (latent there:
 * no scene puts a YRotate into a BvhTree) and can fail to contain the object.  Nothing on the render path reads these boxes: the lowering builds
 * its own f32 boxes, rounded outward (scene_lower.cpp). */
int rttnw_hittable_bounds(const rttnw_scene* s, rttnw_id hittable, double initial_time, double final_time, double out_min_max[6]);

/* The `world: List` handed to `color()` — main.rs:47-55,216. */
int rttnw_scene_set_world(rttnw_scene* s, rttnw_id world_list);
/* Which builder `rttnw_scene_commit` uses for the flat BVHs (before commit; default RTTNW_BVH_AUTO).  Replaces the
 * reference's BvhTree::from / build (hittable.rs:300-353: recursive, random axis per level, full sort per level).
 *   RTTNW_BVH_AUTO         (ABI 3, the default) per tree: the host build below RTTNW_BVH_AUTO_DEVICE_LEAVES leaves — small trees are
 *                          tuned for the LDS-resident kernels (leaf size by what still fits) and build in well under a millisecond —,
 *                          the device binned-SAH build from there on (10^6 leaves: commit 43 instead of 165 ms at the same traversal speed)
 *   RTTNW_BVH_HOST_SAH     binned surface-area-heuristic build on the host (parallel): best traversal, 0.1 s for 10^6 leaves (commit 165 ms)
 *   RTTNW_BVH_DEVICE_LBVH  linear BVH built by HIP kernels (Morton order, Karras hierarchy, bottom-up fit):
 *                          1.4 ms for 10^6 leaves (commit 35 ms), 4-7 % slower traversal; needs a device at commit (no CPU fallback)
 *   RTTNW_BVH_DEVICE_SAH   the binned-SAH build as level-synchronous HIP kernels (binned planes for segments of more than 64
 *                          leaves, an exact sweep by one wave for smaller ones): the host builder's traversal speed at a
 *                          tenth of its build time (8.5 ms for 10^6 leaves, commit 43 ms); needs a device at commit
 * Images do not depend on the choice: the closest hit is topology independent and exact ties are resolved by
 * list order (tests/test_gpu_lbvh.py). */
#define RTTNW_BVH_HOST_SAH 0u
#define RTTNW_BVH_DEVICE_LBVH 1u
#define RTTNW_BVH_DEVICE_SAH 2u
#define RTTNW_BVH_AUTO 3u
#define RTTNW_BVH_AUTO_DEVICE_LEAVES 100000u
int rttnw_scene_set_bvh_builder(rttnw_scene* s, uint32_t builder);
/* Flatten the graph, build the flat BVHs, upload to the current HIP device.  Idempotent. */
int rttnw_scene_commit(rttnw_scene* s);

/* ---------------------------------------------------------------- render (main.rs, camera.rs) */

/* `CameraDescriptor` — camera.rs:5-15 (15 doubles, same field order). */
typedef struct rttnw_camera_desc {
    double lookfrom[3];
    double lookat[3];
    double view_up[3];
    double vertical_fov; /* degrees */
    double aspect_ratio;
    double aperture;
    double focus_distance;
    double open_time;
    double close_time;
} rttnw_camera_desc;

/* What `render()` hard-codes or takes as arguments — main.rs:58,184-197,216,33. */
typedef struct rttnw_params {
    uint32_t width;
    uint32_t height;
    uint32_t spp;        /* `samples` — main.rs:211 */
    uint32_t max_depth;  /* 50 — main.rs:216 */
    double t_min;        /* 0.001 — main.rs:33 */
    double background[3];/* main.rs:43 */
    uint64_t seed;       /* render seed of the keyed sample RNG (DESIGN.md "RNG") */
    uint32_t precision;  /* enum rttnw_precision */
    uint32_t quirks;     /* RTTNW_QUIRK_* bits; RTTNW_QUIRKS_REFERENCE reproduces the reference */
    uint32_t spp_chunk;  /* samples folded sequentially per work item; 0 = library default (4-sample chunks, the last
                            ~1/32 of the samples as single-sample chunks so that a render ends on short items).  The
                            per-pixel sum is ONE chain of chunk sums added in chunk order — a function of spp and
                            spp_chunk alone: not of the image size, the number of GPUs, or how the library splits a
                            long render into launches to keep its chunk-sum workspace within a twelfth of the GPU's memory
                            (4 .. 24 GiB; RTTNW_CHUNK_SUM_BUDGET=<bytes> overrides). */
    uint32_t tile_rank;  /* this GPU's rank in the tile partition (0 for a single GPU) */
    uint32_t tile_world; /* number of GPUs sharing the framebuffer (>= 1) */
    uint32_t collect_counters; /* 1: run the counting kernel variant and fill rttnw_stats (2: plus per-record-kind timing, 3: plus walk-length histograms; debugging) */
    uint32_t sample_begin; /* index of the first sample: this render covers samples [sample_begin, sample_begin + spp) of
                              every pixel and returns THEIR mean.  Draws are keyed by (pixel, sample), so passes over
                              disjoint ranges are independent estimates whose weighted mean is the single render of the
                              union — progressive display, checkpointed long renders (main.rs has only a progress bar). */
    uint32_t reserved0;
} rttnw_params;

typedef struct rttnw_stats {
    uint64_t samples;        /* camera paths traced by this call */
    uint64_t rays;           /* world.hit() calls (main.rs:33) */
    uint64_t nodes_visited;  /* BVH node records read */
    uint64_t prims_tested;   /* primitive records read (incl. medium boundaries) */
    uint64_t texel_fetches;  /* image-texture texel reads */
    double kernel_ms;        /* device time of the trace kernel(s), hipEvent-measured */
    uint32_t n_nodes;        /* flat scene size: 4-wide node records */
    uint32_t n_prims;
    uint32_t scene_bytes;    /* bytes of node+primitive arrays resident on the device */
    uint32_t reserved;       /* render: kernel form that ran — bit 0: decoupled (else lane-owns-path), bit 1: node records resident in LDS, bit 2: three node steps per walk trip (tiny top trees), bit 3: the instantiation whose walk never changes frames (no Translate / YRotate group with a tree of its own), bit 4: ... but tests single wrapped records in place, bit 5: ... in the LEAN flavour (the scene has no MovingSphere, no ConstantMedium and only solid colours: their code is compiled out); bit 6: the decoupled kernel walked the interleaved node + sphere buffer of a big cloud; rttnw_render_multi, rank 0 only — bit 8: the gather went through peer copies (RTTNW_MULTI_GATHER=peer), bit 9: ... because the RCCL set-up failed; scene_info: stack depth */
} rttnw_stats;

/* Framebuffer partition: 8x8-pixel tiles, tile t owned by rank
 * `rttnw_tile_owner(t) = permuted(t) % world`; each rank stores its tiles contiguously
 * ("packed" order, 64 pixels per tile, row-major inside the tile), padded to
 * `tiles_per_rank` tiles so a flat gather has a uniform count. */
typedef struct rttnw_tile_layout {
    uint32_t tiles_x, tiles_y, n_tiles;
    uint32_t tiles_per_rank; /* ceil(n_tiles / world) */
    uint32_t pixels_per_rank;/* tiles_per_rank * 64 */
} rttnw_tile_layout;
int rttnw_tile_layout_get(uint32_t width, uint32_t height, uint32_t world, rttnw_tile_layout* out);

/* `render()` for one GPU: host outputs, blocking.  Row-major, TOP ROW FIRST like main.rs:202-205.
 * `out_linear_rgb` (optional): width*height*3 doubles, the per-pixel mean radiance before gamma
 * (main.rs:217).  `out_rgba8` (optional): width*height*4 bytes after sqrt/clamp/quantise
 * (main.rs:219-225).  Requires tile_world == 1. */
int rttnw_render(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p,
                 double* out_linear_rgb, uint8_t* out_rgba8, rttnw_stats* stats);

/* `render()` on the GPUs of ONE NODE, in one call from one host thread ("library owns its HIP
 * streams / RCCL comms"): the framebuffer's 8x8 tiles are interleaved over `ngpu` ranks, rank r traces its tiles on
 * device `device_ids[r]` (the scene is replicated there on first use), the packed tiles are gathered on rank 0's device —
 * grouped ncclSend/ncclRecv over xGMI between different devices (RCCL is loaded at the first call that needs it), a
 * device-to-device copy for ranks that share rank 0's device — and un-tiled there.  A device may appear more than once
 * (logical ranks; they run one after the other on it), so any partition can be exercised on a single GPU.  The image is
 * bit-identical to rttnw_render's for every ngpu.  Outputs as rttnw_render; `p->tile_rank` / `p->tile_world` are ignored;
 * `stats` (optional) points to ngpu records: samples and device time (trace + resolve) of each rank.  Blocking.
 * Environment: RTTNW_MULTI_GATHER=rccl (default) | peer — `peer` gathers with hipMemcpyPeerAsync on each rank's stream (an event orders
 * the root's un-tile behind it) instead of RCCL; the call falls through to it by itself, with one line on stderr, when RCCL cannot be loaded
 * or ncclCommInitAll fails (stats[0].reserved bits 8 / 9). */
int rttnw_render_multi(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, uint32_t ngpu,
                       const int32_t* device_ids, double* out_linear_rgb, uint8_t* out_rgba8, rttnw_stats* stats);

/* Device-resident form, asynchronous on `hip_stream` (a hipStream_t; NULL = default stream).
 * Traces the tiles owned by (tile_rank, tile_world) and writes them in packed order into
 * `d_packed`: pixels_per_rank pixel records of 4 reals (mean r, g, b, 1) of the kernel's
 * arithmetic type — float for RTTNW_F32, double for RTTNW_F64.  Pad tiles are zero-filled.
 * With `stats != NULL` the call synchronises the stream to fill the device time / counters. */
int rttnw_render_tiles_device(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p,
                              void* d_packed, void* hip_stream, rttnw_stats* stats);

/* After the gather: scatter `world * pixels_per_rank` packed pixel records (rank-major, reals of
 * `precision`) into the row-major top-first framebuffer: `d_linear_rgb` = w*h*3 reals of
 * `precision` (main.rs:217), `d_rgba8` = w*h*4 bytes after main.rs:219-225 (sqrt, clamp 0.999,
 * *256, as u8, alpha 255).  Either output may be NULL.  Asynchronous on `hip_stream`. */
int rttnw_untile_device(uint32_t width, uint32_t height, uint32_t world, uint32_t precision,
                        const void* d_gathered, void* d_linear_rgb, uint8_t* d_rgba8,
                        void* hip_stream);

/* ---------------------------------------------------------------- introspection ------------ */

int rttnw_abi_version(void);
int rttnw_device_count(void);
/* Releases what the library keeps for the life of the process: the RCCL communicator sets rttnw_render_multi caches per list of
 * devices (the reference has no counterpart: rayon's pool goes with the process, main.rs:202).  Optional — the same runs at exit. */
void rttnw_shutdown(void);
const char* rttnw_last_error(void);

/* Debug/inspection: sizes of the lowered scene (valid after commit). */
int rttnw_scene_info(rttnw_scene* s, rttnw_stats* out);

/* What rttnw_scene_commit's build cost (valid after commit). */
typedef struct rttnw_build_info {
    uint32_t builder;     /* RTTNW_BVH_* */
    uint32_t n_nodes;     /* 128-byte 4-wide node records the kernels walk, all trees */
    uint32_t n_prims;     /* leaves of all trees */
    uint32_t stack_depth; /* traversal stack entries a lane needs */
    double lower_ms;      /* host wall time of the lowering, BVH builds included */
    double device_ms;     /* device time of the build kernels + sort (device builder only) */
} rttnw_build_info;
int rttnw_scene_build_info(const rttnw_scene* s, rttnw_build_info* out);

/* Debug/inspection: the BUILDERS' binary trees: copy up to max_nodes 64-byte node records (rt_types.hpp BvhNode: lo0[3]
 * hi0[3] lo1[3] hi1[3] child0 child1 pad pad; child >= 0 inner node, < 0 leaf bits) and the top-level root; returns the
 * node count.  (The kernels walk the 4-wide collapse of these trees, see rttnw_debug_scene_nodes4.) */
int rttnw_debug_scene_nodes(const rttnw_scene* s, void* out_nodes, uint32_t max_nodes, int32_t* top_root);

/* Debug/inspection: the records the kernels walk — the same trees collapsed to 4-WIDE nodes of 128 bytes (rt_types.hpp
 * Bvh4Node: lo[3][4] hi[3][4] (planes by axis, then child) child[4] pad[4]; an unused slot has child == INT32_MIN and an
 * inverted box).  Copies up to max_nodes records and the top-level root; returns the record count. */
int rttnw_debug_scene_nodes4(const rttnw_scene* s, void* out_nodes, uint32_t max_nodes, int32_t* top_root);

/* Debug/inspection: walk sample `sample` of pixel (px, row; row 0 = top) on the device with the kernels of
 * `p->precision` and dump every world.hit() of its path, 20 doubles per bounce:
 *   [0] t  [1..3] p  [4..6] normal  [7] material index  [8] u  [9] v  [10] front_face
 *   [11..13] ray origin  [14..16] ray direction  [17] ray time  [18] emitted.r  [19] attenuation.r (-1: absorbed)
 * `out` must hold max_out*20 + 4 doubles: out[max_out*20 .. +3] = the sample's radiance r,g,b (background taken
 * as black) and its final bounce count, computed by the same path_step() loop the trace kernel runs.
 * Returns the number of bounces written (<= max_out) or a negative error.  Blocking; test use only. */
int rttnw_debug_probe_path(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p,
                           uint32_t px, uint32_t row, uint32_t sample, double* out, uint32_t max_out);

/* ---------------------------------------------------------------- entry-point table -------- */

/* Scene-building entry points as a table, so one host-side scene catalogue (the `scenes.rs`
 * mirror in rttnw_amd/host/scenes.cpp) can drive any implementation of this boundary. */
typedef struct rttnw_builder_api {
    int (*scene_create)(uint64_t, rttnw_scene**);
    void (*scene_destroy)(rttnw_scene*);
    rttnw_id (*tex_solid)(rttnw_scene*, double, double, double);
    rttnw_id (*tex_checker)(rttnw_scene*, rttnw_id, rttnw_id);
    rttnw_id (*tex_noise)(rttnw_scene*, double);
    rttnw_id (*tex_image_rgba8)(rttnw_scene*, const uint8_t*, uint32_t, uint32_t);
    rttnw_id (*mat_lambertian)(rttnw_scene*, rttnw_id);
    rttnw_id (*mat_metal)(rttnw_scene*, double, double, double, double);
    rttnw_id (*mat_dielectric)(rttnw_scene*, double);
    rttnw_id (*mat_diffuse_light)(rttnw_scene*, rttnw_id);
    rttnw_id (*mat_isotropic)(rttnw_scene*, rttnw_id);
    rttnw_id (*sphere)(rttnw_scene*, const double*, double, rttnw_id);
    rttnw_id (*moving_sphere)(rttnw_scene*, const double*, const double*, double, double, double,
                              rttnw_id);
    rttnw_id (*rectangle)(rttnw_scene*, int, double, double, double, double, double, rttnw_id);
    rttnw_id (*cube)(rttnw_scene*, const double*, const double*, rttnw_id);
    rttnw_id (*list)(rttnw_scene*);
    int (*list_push)(rttnw_scene*, rttnw_id, rttnw_id);
    rttnw_id (*bvh_tree)(rttnw_scene*, rttnw_id);
    rttnw_id (*translate)(rttnw_scene*, rttnw_id, const double*);
    rttnw_id (*rotate_y)(rttnw_scene*, rttnw_id, double);
    rttnw_id (*constant_medium)(rttnw_scene*, rttnw_id, double, rttnw_id);
    int (*scene_set_world)(rttnw_scene*, rttnw_id);
    int (*scene_commit)(rttnw_scene*);
    const char* (*last_error)(void);
} rttnw_builder_api;

const rttnw_builder_api* rttnw_builder(void);

#ifdef __cplusplus
}
#endif
#endif /* RTTNW_HIP_H */
