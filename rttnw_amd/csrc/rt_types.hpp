// rt_types.hpp — layout of the flattened scene as it lies in HBM (and, for small scenes, in LDS).
//
// The reference keeps the scene as a graph of `Box/Arc<dyn Hittable>` trait objects
// (src/math/hittable.rs) and walks it by virtual dispatch.  Here the graph is lowered once, on the
// host (scene_lower.cpp), into flat, index-linked POD arrays, one array per primitive kind, so
// that a lane's traversal is a loop over 64-byte node records and 16..64-byte primitive records.
#pragma once
#include <stdint.h>
#include <stdlib.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt {

// ---- primitive kinds (leaf payloads of the flat BVH)
enum : uint32_t {
    PRIM_SPHERE = 0,        // Sphere — hittable.rs:69-131
    PRIM_MOVING_SPHERE = 1, // MovingSphere — hittable.rs:179-245
    PRIM_RECT = 2,          // Rectangle<M,P> — hittable.rs:434-547
    PRIM_BOX = 3,           // Cube (six rectangles, one record) — hittable.rs:549-592
    PRIM_INSTANCE = 4,      // Translate / YRotate chain over a sub-BVH — hittable.rs:594-722
    PRIM_MEDIUM = 5,        // ConstantMedium (never inside the BVH; hit-reference kind only)
    PRIM_SPHERE_WC = 6,     // leaf kind only, RTTNW_F64_STRICT's lowering only: world-space copies of a transformed group's spheres — culled by their
                            // world-space boxes, TESTED in the group's frame (rt_core.hpp sphere_wc_t); their records lie in the sphere arrays
    PRIM_NONE = 7
};

// A child slot of a node: >= 0 -> index of an inner node; < 0 -> ~(leaf bits):
//   leaf bits = kind(3) << 28 | (count-1)(2) << 26 | first(26)   (count same-kind records)
constexpr int32_t CHILD_EMPTY = INT32_MIN;        // kind 7: nothing there
constexpr int32_t STACK_SENTINEL = INT32_MIN + 1; // marks "leave the instance" on the stack
RT_HD int32_t make_leaf(uint32_t kind, uint32_t count, uint32_t first) {
    return ~int32_t((kind << 28) | ((count - 1u) << 26) | first);
}
RT_HD uint32_t leaf_kind(int32_t child) { return uint32_t(~child) >> 28; }
RT_HD uint32_t leaf_count(int32_t child) { return ((uint32_t(~child) >> 26) & 3u) + 1u; }
RT_HD uint32_t leaf_first(int32_t child) { return uint32_t(~child) & 0x3FFFFFFu; }
// hit reference: kind << 28 | record index
RT_HD int32_t make_ref(uint32_t kind, uint32_t index) { return int32_t((kind << 28) | index); }
RT_HD uint32_t ref_kind(int32_t ref) { return uint32_t(ref) >> 28; }
RT_HD uint32_t ref_index(int32_t ref) { return uint32_t(ref) & 0x0FFFFFFFu; }

// 64-byte BVH node holding the boxes of BOTH children: one record fetch decides which child(ren)
// to descend into (the reference's BvhTree::hit tests its own box, then recurses, hittable.rs:356-368).
// Boxes are f32, rounded outward from the f64 bounds; they only cull, they never shape a result.
struct alignas(16) BvhNode {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    int32_t child0, child1;
    int32_t pad0, pad1;
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");

// What the kernels walk: a 4-WIDE node, one 128-byte record (one L2 line) holding the boxes of up to FOUR children,
// made by collapsing the builders' binary tree (scene_lower.cpp collapse4).  A walk then has about half the dependent
// steps — in a wave whose 64 lanes wait for the longest walk of every bounce, the number of lockstep steps, not the
// number of box tests, is what the time goes to.  Boxes are stored by axis (lo[axis][child]) so that the four children's
// planes of one axis arrive as one 16-byte read; an unused slot has child == CHILD_EMPTY and an inverted box.
struct alignas(16) Bvh4Node {
    float lo[3][4];
    float hi[3][4];
    int32_t child[4];
    int32_t pad[4];
};
static_assert(sizeof(Bvh4Node) == 128, "Bvh4Node must be 128 bytes");
constexpr uint32_t BVH4_USED_SIXTEENTHS = 7; // 16-byte pieces of a record that carry data (the LDS copy leaves the pad out)

// What the DECOUPLED kernels walk (trees that live in HBM / the Infinity Cache): the same 4-wide record with QUANTISED boxes, 64
// bytes — two records to a 128-byte line, FOUR 16-byte pieces a visit instead of seven, the tree itself unchanged (record i = 4-wide
// record i: same children in the same slots, same stack bound).  Made from the f32 records on the device (bvh_quant.hpp).
//   piece 0  org[3]: the lower corner of the box around the children; ex[3]: the quantisation step of axis a is 2^(ex[a] - 127)
//   piece 1  q_lo[4], q_hi[4] of x, then of y: child c's box is [org + q_lo[c] step, org + q_hi[c] step], q_lo rounded DOWN and q_hi UP
//            (the boxes only cull: a dequantised box contains the f32 box it was made from); an unused slot has q_lo 255, q_hi 0
//   piece 2  q_lo[4], q_hi[4] of z;   piece 3  child[4], coded like Bvh4Node::child
// Measured (profiles/r04/README.md): spheres_1m f64 277 -> 289 Msamples/s, RTTNW_F64_STRICT 269 -> 288, half the node bytes.  The f32
// kernel first LOST 2 % to this record's per-visit set-up and conversions and walked half-precision node-local records (80 of 128 bytes, one
// v_fma_mix_f32 per plane: +4 %) — until the round's other changes left it bound by bytes alone: 433 -> 476 on these records, and the
// half-precision ones were removed.  An 8-wide quantised record (96 bytes, six pieces, 21 % fewer visits, twice the instructions per visit) lost 40 %.
struct alignas(16) Bvh4QNode {
    float org[3];
    uint8_t ex[3];
    uint8_t n_children;
    uint8_t q[3][2][4]; // [axis][0 lo / 1 hi][slot]
    uint8_t pad[8];
    int32_t child[4];
};
static_assert(sizeof(Bvh4QNode) == 64, "Bvh4QNode must be 64 bytes");
enum : int { NODES_F32X4 = 4, NODES_Q8X4 = 44 }; // what a traversal stack type walks (Stack::WIDE)

// Traversal stack: the first LDS_STACK_ENTRIES entries of a lane live in LDS, deeper ones (a 4-wide walk can have three
// pending children per level, but rarely has) in a per-lane strip of global memory.
#ifndef RT_LDS_STACK_ENTRIES // (12 / 10 / 8 entries: final_scene f32 -0.4 / -1.5 / -2.5 %)
#define RT_LDS_STACK_ENTRIES 16
#endif
constexpr uint32_t LDS_STACK_ENTRIES = RT_LDS_STACK_ENTRIES;
// LDS bytes of the lane-owns-path kernel's resident form: node records (without their pad) + the lanes' stacks.
// RenderConsts::lds_nodes of the LDS form: the node count, and above it how many Perlin tables follow the stacks in LDS
constexpr uint32_t LDS_NODES_MASK = 0x00FFFFFFu, LDS_PERLIN_SHIFT = 24u;
RT_HD size_t lds_perlin_bytes(size_t n_tables, size_t real_bytes) { return n_tables * 768u * (real_bytes + 1u); } // [n][256][3] reals + [n][3][256] bytes
RT_HD size_t lds_pad32(size_t bytes) { return (bytes + 31u) / 32u * 32u; } // staged arrays start on 32-byte boundaries and are copied in whole 32-byte units
inline size_t lds_form_bytes(uint32_t n_nodes4, uint32_t stack_depth, uint32_t block) {
    (void)stack_depth; // the LDS part of a stack has a fixed size (+ 1: the spare slot of the branch-free pushes)
    return size_t(n_nodes4) * 16 * BVH4_USED_SIXTEENTHS + size_t(LDS_STACK_ENTRIES + 1) * block * sizeof(int32_t);
}

// `seq` in the records below: position of the object in the reference's traversal order of the
// world List (depth-first).  List::hit lets a LATER item replace an earlier one at exactly equal t
// (hittable.rs:157-159) — e.g. the Cornell blocks' bottom faces coincide with the floor — so exact
// ties are resolved by the larger seq.
template <typename R> struct alignas(sizeof(R) * 4) SphereRec { R cx, cy, cz, r; };
template <typename R> struct MovingSphereRec { R c0[3], r, c1[3], t0, t1; int32_t mat; int32_t seq; };
// Records are aligned so that ONE by-value copy at the top of a test is a couple of 16-byte loads issued together
// (field-by-field reads through a reference end up as dependent loads inside the test's branches).
template <typename R> struct alignas(sizeof(R) * 4) RectRec { R a0, a1, b0, b1, k; int32_t plane; int32_t mat; int32_t seq; };
template <typename R> struct alignas(sizeof(R) * 4) BoxRec { R mn[3], mx[3]; int32_t mat; int32_t seq; };

enum : int32_t { OP_TRANSLATE = 0, OP_ROTATE_Y = 1 };
constexpr int MAX_INSTANCE_OPS = 8;
// ops[0] is the OUTERMOST wrapper (applied to the ray first): `x.rotate_y(a).translate(v)`
// = Translate(YRotate(x)) lowers to ops = { translate v, rotate a }.  A transformed object inside a transformed group is
// lowered as an instance of its own whose chain is the group's wrappers followed by its own (scene_lower.cpp), so
// instances never nest; only the first n_ops entries of a record are ever read.
template <typename R> struct alignas(16) InstanceRec {
    int32_t n_ops;
    int32_t root;        // sub-BVH root node
    int32_t single_leaf; // the wrapped group is ONE record: its leaf bits (< 0), tested in place without walking `root`; else 0
    int32_t pad0;
    struct Op { int32_t type; int32_t pad; R v[3]; } ops[MAX_INSTANCE_OPS]; // translate: offset; rotate: {sin, cos, -}
};

// The head of an instance record: what a chain of up to FAST_INSTANCE_OPS wrappers (every scene of the reference: two) needs.
// The kernels copy it by value and run unrolled code; longer chains take the general, out-of-line path over the full record.
constexpr int FAST_INSTANCE_OPS = 3;
template <typename R> struct alignas(16) InstanceHead {
    int32_t n_ops, root, single_leaf, pad0;
    typename InstanceRec<R>::Op ops[FAST_INSTANCE_OPS];
};

template <typename R> struct MediumRec {
    int32_t b_first;  // boundary = medium_refs[b_first .. b_first + b_count): make_ref(kind, index) of spheres / boxes; more than
    int32_t b_count;  //   one = a List / BvhTree boundary (ConstantMedium takes any Hittable, hittable.rs:731)
    int32_t inst;     // instance record whose ops take the world ray to the boundary's space, or -1
    int32_t n_outer;  // how many of those ops (the leading ones) wrap the MEDIUM itself (a medium inside a transformed
                      //   group): its hit record is unwound through them like any other record of the group
    int32_t mat;      // Isotropic material
    int32_t ref0;     // medium_refs[b_first], inline: the common one-primitive boundary needs no second lookup
    R neg_inv_density;
};

// A primitive record's material reference: the material index, with MAT_UV_FLAG set when that material's texture can read the
// hit's (u, v) (an image texture, possibly under a checker): the hit record then computes them; two dependent loads
// (material, texture) just to find that out were 4 % of the f64 kernel.
constexpr int32_t MAT_UV_FLAG = 1 << 30;
constexpr int32_t MAT_INDEX_MASK = MAT_UV_FLAG - 1;
// SceneView::sphere_mat only: MAT_HOME_FLAG marks the world-space copy of a sphere of a transformed group (scene_lower.cpp);
// the entry then holds, instead of a material, the object-space sphere record and the transform chain its hit record is made from.
constexpr int32_t MAT_HOME_FLAG = 1 << 29;
constexpr int32_t MAT_HOME_INST_SHIFT = 20, MAT_HOME_INST_MAX = 511;
constexpr uint32_t MAT_HOME_SPHERE_MASK = (1u << MAT_HOME_INST_SHIFT) - 1u, MAT_HOME_SPHERE_MAX = MAT_HOME_SPHERE_MASK;
enum : int32_t { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_DIFFUSE_LIGHT = 3, MAT_ISOTROPIC = 4 };
// Padded to a power of two (64 B in f64, 32 B in f32): a record never straddles two 128-byte lines.  On spheres_1m (10^6 materials, one L2-miss line
// per hit at best) 30 % of the 40-byte f64 records did (tests/hostsim/cache_model.hpp: 7.3 -> 5.8 material miss lines per sample).
#ifndef RT_MAT_PAD
#define RT_MAT_PAD 1
#endif
template <typename R> struct alignas(RT_MAT_PAD ? sizeof(R) * 8 : 8) MaterialRec {
    int32_t type;
    int32_t tex;  // texture index, or -1 when the colour below is the whole texture (solid)
    R albedo[3];  // metal albedo / inlined solid colour
    R param;      // metal fuzz / dielectric refraction index
};

enum : int32_t { TEX_SOLID = 0, TEX_CHECKER = 1, TEX_NOISE = 2, TEX_IMAGE = 3, TEX_CYAN = 4 };
template <typename R> struct TextureRec {
    int32_t type;
    int32_t a, b; // checker: odd, even texture ids; noise: perlin table index; image: image index
    int32_t pad;
    R color[3];
    R scale;
};
struct ImageRec { uint32_t offset, w, h, pad; }; // offset in texels (uint32 RGBA8) into `texels`

template <typename R> struct CameraRec { // Camera — camera.rs:18-29
    R origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3];
    R lens_radius, open_time, close_time;
};

// Everything a lane needs, by pointer.  Same struct for HBM- and LDS-resident node/primitive arrays.
template <typename R> struct SceneView {
    const Bvh4Node* nodes;
    const Bvh4QNode* nodes4q; // the same trees as quantised records, index for index (the f64 decoupled kernel; else null)
    const SphereRec<R>* spheres;
    const int32_t* sphere_mat;
    const int32_t* sphere_seq; // list-order sequence numbers, read only to break exact ties in t
    const MovingSphereRec<R>* moving;
    const RectRec<R>* rects;
    const BoxRec<R>* boxes;
    const InstanceRec<R>* insts;
    const MediumRec<R>* media;
    const int32_t* medium_refs; // boundary primitives of the media
    const MaterialRec<R>* mats;
    const TextureRec<R>* texs;
    const ImageRec* images;
    const uint32_t* texels;  // RGBA8
    const R* perlin_vec;     // [n_perlin][256][3]
    const uint8_t* perlin_perm; // [n_perlin][3][256]
    int32_t top_root;
    int32_t n_media;
};

// Division of a 32-bit value by a run-time constant as multiply-high + shifts (Granlund-Montgomery, the branch-free
// form): q = (t + ((n - t) >> 1)) >> sh,  t = mulhi(n, mul).  Set up on the host (make_fastdiv).
struct FastDiv { uint32_t mul, sh, d, pad; };
RT_HD uint32_t fdiv(uint32_t n, FastDiv f) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t = __umulhi(n, f.mul);
#else
    const uint32_t t = uint32_t((uint64_t(n) * f.mul) >> 32);
#endif
    return f.d == 1u ? n : (t + ((n - t) >> 1)) >> f.sh;
}
inline FastDiv make_fastdiv(uint32_t d) { // d >= 1
    FastDiv f{0, 0, d, 0};
    if (d == 1) return f; // fdiv() returns n itself
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;                 // l = ceil(log2 d)
    f.mul = uint32_t(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh = l - 1;
    return f;
}

struct RenderConsts {
    uint32_t width, height, spp, max_depth;
    uint32_t spp_chunk, n_chunks; // chunk schedule of the whole render, see plan_chunks(); n_chunks: the chunks of THIS launch
    uint32_t n_main;              // (whole render) chunks [0, n_main) hold spp_chunk samples, the chunks after them ONE sample each
    uint32_t tiles_x, tiles_y, n_tiles;
    uint32_t tile_rank, tile_world, my_tiles; // tiles this rank really owns
    uint32_t quirks;
    uint32_t chunk_base;          // the render's chunk that is this launch's chunk 0 (a long render is traced in several launches)
    uint32_t stack_depth;
    FastDiv div_jobs_per_group, div_tiles_x; // job index -> (chunk, tile, pixel) without integer division (job_decode)
    uint32_t jobs_per_chunk;                 // my_tiles * 64: the sums of chunk c start at c * jobs_per_chunk
    uint32_t n_jobs;                         // job indices handed out (chunk groups are padded: some jobs are empty)
    uint32_t profile;   // counting variant: 2 = also bucket the leaf clock by record kinds (atomics: perturbs the other clocks)
    uint32_t lds_nodes; // lane-owns-path kernel: number of BVH nodes resident in LDS (0 = nodes read from global memory) | Perlin tables in LDS << 24
    uint64_t seed;
    uint64_t sample_begin; // index of the first sample of this launch (rttnw_params::sample_begin + the pass's offset)
    uint32_t lds_recs[6];  // LDS form: how many insts / rects / moving / boxes / sphere_mat / (spare) records follow the Perlin tables in LDS (0: that array is read from global memory)
    uint32_t scene_flags;  // SCENE_NO_TIME: nothing in the scene reads Ray::time (no MovingSphere): the shutter draw of camera.rs:82 is keyed, so skipping it shifts nothing
    uint32_t pad0;
    double inv_width, inv_height; // 1 / width, 1 / height: the contracted f64 build multiplies where main.rs:213-214 divides (the strict build divides)
};
enum : uint32_t { SCENE_NO_TIME = 1u };

// A pixel's samples are split into CHUNKS; a job = (pixel, chunk) folds its samples sequentially (main.rs:211-216) and
// the resolve step adds a pixel's chunk sums, in chunk order, onto the pixel's running sum: ONE chain per pixel,
//     sum = (((c0 + c1) + c2) + ...),
// whatever else happens.  Jobs are handed out chunk-major, so the LAST chunks decide how long the final lanes of a render
// run alone.  Default schedule (user_chunk == 0): 4-sample chunks for the first ~31/32 of the samples, then single-sample
// chunks — the render ends on one-sample jobs whatever spp is (with uniform ceil(spp/256) chunks an 8-GPU run at spp 8000
// lost 11 % of a rank's throughput to its tail).  Short jobs also keep a wave's lanes on the same few pixels: measured on
// final_scene, main chunks of 16 / 8 / 4 / 2 samples give 1140 / 1206 / 1245 / 1250 Msamples/s (cornell_box 1589 / 1600 /
// 1595 / 1582).  The schedule is a function of spp (and user_chunk) ALONE — not of the image, the rank count or the
// workspace: the per-pixel chain, hence the image, is bit-identical for any tile_world and any launch split below.
inline void plan_chunks(RenderConsts& rc, uint32_t spp, uint32_t user_chunk) {
    if (user_chunk) {
        rc.spp_chunk = user_chunk;
        rc.n_chunks = rc.n_main = (spp + user_chunk - 1) / user_chunk;
        return;
    }
    const uint32_t tail = spp < 32u ? spp : spp / 32u;
    rc.spp_chunk = 4;
    rc.n_main = (spp - tail) / 4u;
    rc.n_chunks = rc.n_main + (spp - rc.n_main * 4u);
}
// Launches.  The chunk sums wait in a workspace until the resolve step has added them to the running sums; a rank keeps
// at most a BUDGET of them, so a render beyond it is traced in LAUNCHES over consecutive chunk ranges, each followed by its
// resolve step, of near-equal size and a multiple of the 16 chunks a job group spans (job_decode).  Because the resolve step
// CONTINUES the pixel's chain, the image does not depend on the split: any budget renders bit for bit the same image (round 2
// had "passes" with their own schedules).
// The budget is a pure speed / memory knob, and launches cost: a launch ends with its lanes running dry one by one — a lane takes
// no new job when the launch has none left, so every wave spends its last ~26 bounce rounds with fewer and fewer live lanes —
// about 10 ms per boundary on final_scene (five launches of the headline frame: f64 1139 against 1250 Msamples/s; two launches
// in flight on two streams do not help: the next kernel's workgroups take the free CUs but do not refill those lanes).  So the
// budget follows the memory the device has (render_common.hpp device_chunk_budget: a twelfth of its HBM, 4 .. 24 GiB — 24 GiB on
// an MI355X): every BASELINE configuration is ONE launch (800 x 800 spp 5000 and a rank's share of 1600 x 1600 spp 10000 hold
// 19.6 GiB of f64 sums: 2.5 % faster than six launches at 4 GiB), the whole 1600 x 1600 x 10000 frame on one GPU seven.
// CHUNK_SUM_BUDGET is the fallback where no device is asked (the host test build); RTTNW_CHUNK_SUM_BUDGET=<bytes> overrides either
// (tests: many launches on small images; a caller who wants the memory back).
constexpr uint64_t CHUNK_SUM_BUDGET = 4ull << 30;
inline uint64_t chunk_sum_budget(uint64_t device_budget = 0) {
#if !defined(__HIP_DEVICE_COMPILE__)
    if (const char* e = getenv("RTTNW_CHUNK_SUM_BUDGET")) {
        const unsigned long long v = strtoull(e, nullptr, 10);
        if (v) return v;
    }
#endif
    return device_budget ? device_budget : CHUNK_SUM_BUDGET;
}
inline uint32_t launch_chunks(uint64_t rank_tile_pixels, uint64_t bytes_per_sum, uint32_t total_chunks, uint64_t device_budget = 0) {
    if (total_chunks <= 1) return 1;
    const uint64_t per_chunk = (rank_tile_pixels ? rank_tile_pixels : 1) * (bytes_per_sum ? bytes_per_sum : 1);
    uint64_t k = chunk_sum_budget(device_budget) / per_chunk;
    // job indices are 32-bit (a launch has ceil(K / 16) * 16 * rank_tile_pixels of them)
    const uint64_t by_index = ((1ull << 32) - 1) / (rank_tile_pixels ? rank_tile_pixels : 1);
    if (by_index < 32) k = 1; else if (k > by_index - 16) k = by_index - 16;
    if (k >= total_chunks) return total_chunks; // the whole render in one launch
    if (k < 16) return uint32_t(k < 1 ? 1 : k);  // an image so large that a job group's 16 chunk planes exceed the budget
    k -= k % 16;
    const uint64_t n_launch = (total_chunks + k - 1) / k;                 // as few launches as the budget allows ...
    const uint64_t even = (total_chunks + n_launch - 1) / n_launch;        // ... of near-equal size
    return uint32_t((even + 15) / 16 * 16);                               // (<= k: k is a multiple of 16 and even <= k)
}
// Chunk `chunk` of the WHOLE render -> its samples [s, s_end).
RT_HD void chunk_samples(const RenderConsts& rc, uint32_t chunk, uint32_t& s, uint32_t& s_end) {
    if (chunk < rc.n_main) {
        s = chunk * rc.spp_chunk;
        s_end = s + rc.spp_chunk < rc.spp ? s + rc.spp_chunk : rc.spp;
    } else {
        s = rc.n_main * rc.spp_chunk + (chunk - rc.n_main);
        s_end = s + 1u;
    }
}

struct DeviceCounters {
    unsigned long long rays, nodes, prims, texels;
    unsigned long long dbg[160]; // statistics of the counting variants (RTTNW_DEBUG_SCHED prints them; meaning per kernel in trace_kernels.hpp)
};

} // namespace rt
