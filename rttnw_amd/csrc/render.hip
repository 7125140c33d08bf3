// render.hip — the device half of include/rttnw_hip.h for gfx950 (MI355X).
//
// One kernel does the whole per-pixel sample loop of main.rs:202-229:
//   * persistent workgroups (4 wave64 each) pull JOBS from one global counter.  A job is
//     (pixel, chunk of `spp_chunk` samples); 64 consecutive jobs are the 64 pixels of one 8x8 tile,
//     so a wave starts out on a coherent tile.  Idle lanes are counted with __ballot, the wave leader
//     takes that many jobs with ONE atomic and the lanes pick theirs by popcount rank — lanes never
//     wait for the longest path in the wave (path lengths run 1..50, the Cornell blocks trap rays).
//   * a lane folds its job's samples sequentially (main.rs:211) with one path per lane: regenerate a
//     camera ray when the path dies, otherwise do one world.hit + scatter (rt_core.hpp).
//   * BVH traversal keeps its stack in LDS, interleaved by lane (entry e of lane l at e*blockDim+l:
//     conflict-free ds_read/ds_write_b32).
//   * job sums go to a partial buffer; a resolve kernel adds a pixel's chunks in chunk order, so the
//     image is bit-identical whatever the scheduling, the grid size or the number of GPUs.
// No CPU fallback: every entry point needs a HIP device.
#include "../../include/rttnw_hip.h"
#include "rt_core.hpp"
#include "rt_sched.hpp"
#include "scene_handle.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace rt {

// ---------------------------------------------------------------------------------------------
// device-side helpers
// ---------------------------------------------------------------------------------------------
struct LdsStack {
    int32_t* base;   // &lds[threadIdx.x]
    uint32_t stride; // blockDim.x
    __device__ __forceinline__ void set(int i, int32_t v) { base[uint32_t(i) * stride] = v; }
    __device__ __forceinline__ int32_t get(int i) const { return base[uint32_t(i) * stride]; }
};

template <bool COUNT> struct CounterSel { using type = NoCounters; };
template <> struct CounterSel<true> { using type = LaneCounters; };

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

constexpr int TRACE_BLOCK = 256;

// The per-pixel sample loop of main.rs:202-229 as ONE persistent kernel with an intra-wave scheduler.
//
// Every lane owns one path and is, at any moment, waiting for exactly one kind of step (rt_sched.hpp ST_*).
// Each iteration the wave ballots the states and runs ONE stage for the lanes waiting on it, so the
// instruction stream of an iteration is one stage's code at (#waiting lanes / 64) utilisation, instead of node
// code + every primitive kind + shading serialised behind the slowest lane of the wave.  Lanes that are not
// picked keep their state; every executed stage advances all of its lanes, so the loop always makes progress.
// The policy keeps lanes accumulated at inner nodes (80 % of all steps) and drains the other queues at
// thresholds that adapt to the scene (rt_sched.hpp adapt_policy).
// Jobs ((pixel, sample chunk); 64 consecutive jobs = one 8x8 tile) come from one global counter: the lanes
// that ran dry are counted with __ballot, the wave leader takes that many jobs with ONE atomic and each lane
// picks its own by popcount rank.
template <typename R, bool COUNT>
__global__ __launch_bounds__(TRACE_BLOCK) void trace_kernel(SceneView<R> sc, CameraRec<R> cam, RenderConsts rc, R bg_r, R bg_g,
                                                            R bg_b, R t_min, R* __restrict__ partial,
                                                            unsigned long long* __restrict__ job_counter,
                                                            DeviceCounters* __restrict__ counters) {
    extern __shared__ int32_t lds_stack[];
    LdsStack stack{lds_stack + threadIdx.x, blockDim.x};
    typename CounterSel<COUNT>::type cnt;

    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long n_jobs = (unsigned long long)rc.my_tiles * 64ull * rc.n_chunks;
    const V3<R> background(bg_r, bg_g, bg_b);

    Lane<R> ln;
    ln.init();
    SchedPolicy pol = default_policy();
    uint32_t served[ST_COUNT] = {0, 0, 0, 0, 0, 0}; // lane-steps per stage so far (wave-uniform)
    uint32_t execs[ST_COUNT] = {0, 0, 0, 0, 0, 0};
    unsigned long long pop[ST_COUNT] = {0, 0, 0, 0, 0, 0};
    uint32_t iter = 0;

    // vote + bookkeeping of one scheduler iteration; returns the stage to run (ST_DONE: every lane is done)
    auto vote = [&]() -> uint32_t {
        uint32_t n[ST_COUNT];
#pragma unroll
        for (uint32_t k = 0; k < ST_COUNT; ++k) n[k] = uint32_t(__popcll(__ballot(ln.st == k)));
        const uint32_t pick = sched_pick(n, pol);
        if (pick == ST_DONE) return pick;
#pragma unroll
        for (uint32_t k = 0; k < ST_COUNT; ++k)
            if (pick == k) { served[k] += n[k]; execs[k] += 1; }
        if ((++iter & 127u) == 0) adapt_policy(pol, served);
        if constexpr (COUNT) { // scheduler statistics: time-average population of every queue
#pragma unroll
            for (uint32_t k = 0; k < ST_COUNT; ++k) pop[k] += n[k];
        }
        return pick;
    };

    for (;;) {
        uint32_t pick = vote();
        // inner-node visits are most of the iterations: run them in a tight loop of their own, so that only the
        // traversal registers are loop-carried there (no shuffling of the whole lane state per iteration)
        while (pick == ST_NODE) {
            if (ln.st == ST_NODE) ln.step_node(sc, t_min, stack, cnt);
            pick = vote();
        }
        if (pick == ST_DONE) break;

        if (pick == ST_SPHERE || pick == ST_BOX || pick == ST_MISC) {
            if (ln.st == pick) ln.step_leaf(sc, t_min, stack, cnt);
        } else if (pick == ST_POST) {
            if (ln.st == ST_POST) ln.step_post(sc, rc, background, t_min, cnt);
        } else { // ST_NEW
            // job hand-out for the lanes that ran dry: wave-aggregated, one atomic per refill event
            const bool need = ln.needs_job();
            const unsigned long long mask = __ballot(need);
            if (mask != 0ull) {
                if (need && ln.has_job) { // retire the finished job: its sequential sum
                    R* dst = partial + ln.job * 3ull;
                    dst[0] = ln.acc.x; dst[1] = ln.acc.y; dst[2] = ln.acc.z;
                    ln.has_job = false;
                }
                const int leader = __ffsll((long long)mask) - 1;
                unsigned long long base = 0;
                if (int(lane) == leader) base = atomicAdd(job_counter, (unsigned long long)__popcll(mask));
                const uint32_t blo = __shfl(uint32_t(base), leader, 64), bhi = __shfl(uint32_t(base >> 32), leader, 64);
                base = (unsigned long long)blo | ((unsigned long long)bhi << 32);
                if (need) ln.take_job(base + (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull)), n_jobs, rc);
            }
            if (ln.st == ST_NEW && ln.s < ln.s_end) ln.step_new(sc, cam, rc, cnt);
        }
    }

    if constexpr (COUNT) {
        uint32_t r = wave_sum(cnt.rays), nn = wave_sum(cnt.nodes), p = wave_sum(cnt.prims), t = wave_sum(cnt.texels);
        if (lane == 0) {
            atomicAdd(&counters->rays, (unsigned long long)r);
            atomicAdd(&counters->nodes, (unsigned long long)nn);
            atomicAdd(&counters->prims, (unsigned long long)p);
            atomicAdd(&counters->texels, (unsigned long long)t);
#pragma unroll
            for (uint32_t k = 0; k < ST_COUNT; ++k) {
                atomicAdd(&counters->stage_execs[k], (unsigned long long)execs[k]);
                atomicAdd(&counters->stage_lanes[k], (unsigned long long)served[k]);
                atomicAdd(&counters->thr_sum[k], (unsigned long long)pol.threshold[k]);
                atomicAdd(&counters->pop_sum[k], pop[k]);
            }
            atomicAdd(&counters->waves, 1ull);
        }
    }
}

// Sum a pixel's chunk partials in chunk order, divide by spp (main.rs:217): packed pixel records
// (r, g, b, 1).  Pad tiles (>= my_tiles) are zero-filled.
template <typename R>
__global__ void resolve_kernel(const R* __restrict__ partial, R* __restrict__ packed, RenderConsts rc, uint32_t pixels_per_rank) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pixels_per_rank) return;
    R r = 0, g = 0, b = 0, a = 0;
    const unsigned long long jobs_per_chunk = (unsigned long long)rc.my_tiles * 64ull;
    if (p < jobs_per_chunk) {
        for (uint32_t c = 0; c < rc.n_chunks; ++c) {
            const R* src = partial + ((unsigned long long)c * jobs_per_chunk + p) * 3ull;
            r = r + src[0]; g = g + src[1]; b = b + src[2];
        }
        const R spp = R(rc.spp);
        r = r / spp; g = g / spp; b = b / spp;
        a = R(1);
    }
    R* dst = packed + (unsigned long long)p * 4ull;
    dst[0] = r; dst[1] = g; dst[2] = b; dst[3] = a;
}

// Gathered packed records (rank-major) -> row-major top-first framebuffer + RGBA8 (main.rs:219-225).
template <typename R>
__global__ void untile_kernel(const R* __restrict__ gathered, R* __restrict__ linear_rgb, uint8_t* __restrict__ rgba8, uint32_t width,
                              uint32_t height, uint32_t tiles_x, uint32_t world, uint32_t pixels_per_rank) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= width || y >= height) return;
    const uint32_t permuted = tile_permuted(x >> 3, y >> 3, tiles_x);
    const uint32_t owner = permuted % world, local_tile = permuted / world;
    const unsigned long long src = (unsigned long long)owner * pixels_per_rank + local_tile * 64ull + ((y & 7u) << 3) + (x & 7u);
    const R r = gathered[src * 4], g = gathered[src * 4 + 1], b = gathered[src * 4 + 2];
    const unsigned long long o = (unsigned long long)y * width + x;
    if (linear_rgb) { linear_rgb[o * 3] = r; linear_rgb[o * 3 + 1] = g; linear_rgb[o * 3 + 2] = b; }
    if (rgba8) {
        rgba8[o * 4] = quantise(r); rgba8[o * 4 + 1] = quantise(g); rgba8[o * 4 + 2] = quantise(b); rgba8[o * 4 + 3] = 255;
    }
}

// Debug probe: one lane walks one sample's path and dumps every hit record (t, p, normal, material, u, v,
// front_face) plus the ray it was found with — the device half of the per-bounce CPU-vs-GPU vector tests.
constexpr int PROBE_STRIDE = 20;
template <typename R>
__global__ void probe_path_kernel(SceneView<R> sc, CameraRec<R> cam, RenderConsts rc, R t_min, uint32_t px, uint32_t row,
                                  uint32_t sample, double* __restrict__ out, uint32_t max_out, int32_t* __restrict__ n_out) {
    extern __shared__ int32_t lds_stack[];
    if (threadIdx.x != 0) return;
    LdsStack stack{lds_stack, blockDim.x};
    NoCounters cnt;
    PathState<R> ps;
    path_begin(ps, cam, rc, px, row, sample);
    uint32_t n = 0;
    while (n < max_out) {
        HitRecord<R> rec;
        const Ray<R> ray = ps.ray;
        if (!world_hit(sc, ps.ray, t_min, ps.key, ps.bounce, rc.quirks, rec, stack, cnt)) break;
        double* o = out + size_t(n) * PROBE_STRIDE;
        o[0] = rec.t; o[1] = rec.p.x; o[2] = rec.p.y; o[3] = rec.p.z;
        o[4] = rec.normal.x; o[5] = rec.normal.y; o[6] = rec.normal.z; o[7] = double(rec.mat);
        o[8] = rec.u; o[9] = rec.v; o[10] = rec.front_face ? 1.0 : 0.0;
        o[11] = ray.o.x; o[12] = ray.o.y; o[13] = ray.o.z; o[14] = ray.d.x; o[15] = ray.d.y; o[16] = ray.d.z; o[17] = ray.time;
        ++n;
        V3<R> att, em;
        const bool cont = shade(sc, rec, ps.key, ps.bounce, ps.ray, att, em, cnt);
        o[18] = em.x; o[19] = cont ? att.x : -1.0;
        if (!cont) break;
        ps.bounce += 1;
        if (ps.bounce >= rc.max_depth) break;
    }
    *n_out = int32_t(n);
    // the same sample again through path_step(), exactly as the trace kernel runs it: radiance after `out`
    path_begin(ps, cam, rc, px, row, sample);
    while (path_step(ps, sc, rc, V3<R>(), t_min, stack, cnt)) {}
    double* tail = out + size_t(max_out) * PROBE_STRIDE;
    tail[0] = ps.radiance.x; tail[1] = ps.radiance.y; tail[2] = ps.radiance.z; tail[3] = double(ps.bounce);
}

// ---------------------------------------------------------------------------------------------
// host-side state
// ---------------------------------------------------------------------------------------------
#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_));             \
            return RTTNW_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)

template <typename T> struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    int upload(const std::vector<T>& v) {
        release();
        n = v.size();
        const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
        HIP_TRY(hipMalloc((void**)&p, bytes));
        if (n) HIP_TRY(hipMemcpy(p, v.data(), n * sizeof(T), hipMemcpyHostToDevice));
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

template <typename R> struct DeviceScene {
    bool ready = false;
    DevBuf<BvhNode> nodes;
    DevBuf<SphereRec<R>> spheres;
    DevBuf<int32_t> sphere_mat, sphere_seq;
    DevBuf<MovingSphereRec<R>> moving;
    DevBuf<RectRec<R>> rects;
    DevBuf<BoxRec<R>> boxes;
    DevBuf<InstanceRec<R>> insts;
    DevBuf<MediumRec<R>> media;
    DevBuf<MaterialRec<R>> mats;
    DevBuf<TextureRec<R>> texs;
    DevBuf<ImageRec> images;
    DevBuf<uint32_t> texels;
    DevBuf<R> perlin_vec;
    DevBuf<uint8_t> perlin_perm;
    SceneView<R> view{};
    size_t bytes = 0;

    int upload(const FlatScene& f) {
        std::vector<SphereRec<R>> sp;
        for (auto& s : f.spheres) sp.push_back({R(s.cx), R(s.cy), R(s.cz), R(s.r)});
        std::vector<MovingSphereRec<R>> mv;
        for (auto& m : f.moving) {
            MovingSphereRec<R> o{};
            for (int k = 0; k < 3; ++k) { o.c0[k] = R(m.c0[k]); o.c1[k] = R(m.c1[k]); }
            o.r = R(m.r); o.t0 = R(m.t0); o.t1 = R(m.t1); o.mat = m.mat; o.seq = m.seq;
            mv.push_back(o);
        }
        std::vector<RectRec<R>> rc_;
        for (auto& r : f.rects) rc_.push_back({R(r.a0), R(r.a1), R(r.b0), R(r.b1), R(r.k), r.plane, r.mat, r.seq, 0});
        std::vector<BoxRec<R>> bx;
        for (auto& b : f.boxes) {
            BoxRec<R> o{};
            for (int k = 0; k < 3; ++k) { o.mn[k] = R(b.mn[k]); o.mx[k] = R(b.mx[k]); }
            o.mat = b.mat; o.seq = b.seq;
            bx.push_back(o);
        }
        std::vector<InstanceRec<R>> in;
        for (auto& i : f.insts) {
            InstanceRec<R> o{};
            o.n_ops = i.n_ops; o.root = i.root;
            for (int k = 0; k < MAX_INSTANCE_OPS; ++k) {
                o.ops[k].type = i.ops[k].type;
                for (int c = 0; c < 3; ++c) o.ops[k].v[c] = R(i.ops[k].v[c]);
            }
            in.push_back(o);
        }
        std::vector<MediumRec<R>> md;
        for (auto& m : f.media) md.push_back({m.boundary, m.inst, m.mat, 0, R(m.neg_inv_density)});
        std::vector<MaterialRec<R>> mt;
        for (auto& m : f.mats) mt.push_back({m.type, m.tex, {R(m.albedo[0]), R(m.albedo[1]), R(m.albedo[2])}, R(m.param)});
        std::vector<TextureRec<R>> tx;
        for (auto& t : f.texs) tx.push_back({t.type, t.a, t.b, 0, {R(t.color[0]), R(t.color[1]), R(t.color[2])}, R(t.scale)});
        std::vector<R> pv;
        for (double v : f.perlin_vec) pv.push_back(R(v));

        int rc;
        if ((rc = nodes.upload(f.nodes)) || (rc = spheres.upload(sp)) || (rc = sphere_mat.upload(f.sphere_mat)) ||
            (rc = sphere_seq.upload(f.sphere_seq)) || (rc = moving.upload(mv)) || (rc = rects.upload(rc_)) ||
            (rc = boxes.upload(bx)) || (rc = insts.upload(in)) || (rc = media.upload(md)) || (rc = mats.upload(mt)) ||
            (rc = texs.upload(tx)) || (rc = images.upload(f.images)) || (rc = texels.upload(f.texels)) ||
            (rc = perlin_vec.upload(pv)) || (rc = perlin_perm.upload(f.perlin_perm)))
            return rc;
        view.nodes = nodes.p; view.spheres = spheres.p; view.sphere_mat = sphere_mat.p; view.sphere_seq = sphere_seq.p;
        view.moving = moving.p; view.rects = rects.p; view.boxes = boxes.p; view.insts = insts.p; view.media = media.p;
        view.mats = mats.p; view.texs = texs.p; view.images = images.p; view.texels = texels.p;
        view.perlin_vec = perlin_vec.p; view.perlin_perm = perlin_perm.p;
        view.top_root = f.top_root;
        view.n_media = int32_t(f.media.size());
        bytes = f.nodes.size() * sizeof(BvhNode) + sp.size() * sizeof(SphereRec<R>) + mv.size() * sizeof(MovingSphereRec<R>) +
                rc_.size() * sizeof(RectRec<R>) + bx.size() * sizeof(BoxRec<R>) + in.size() * sizeof(InstanceRec<R>);
        ready = true;
        return 0;
    }
    void release() {
        nodes.release(); spheres.release(); sphere_mat.release(); sphere_seq.release(); moving.release(); rects.release();
        boxes.release(); insts.release(); media.release(); mats.release(); texs.release(); images.release();
        texels.release(); perlin_vec.release(); perlin_perm.release();
        ready = false;
    }
};

struct DeviceState {
    int device = -1;
    int num_cus = 0;
    DeviceScene<float> s32;
    DeviceScene<double> s64;
    // workspace, grown on demand and kept
    void* partial = nullptr;
    size_t partial_bytes = 0;
    unsigned long long* job_counter = nullptr; // [0] job counter, then DeviceCounters
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // scratch for the blocking host-output render()
    void* packed = nullptr; size_t packed_bytes = 0;
    void* linear = nullptr; size_t linear_bytes = 0;
    uint8_t* rgba = nullptr; size_t rgba_bytes = 0;
};

static int grow(void** p, size_t* have, size_t want) {
    if (*have >= want && *p) return 0;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *have = 0;
    HIP_TRY(hipMalloc(p, std::max<size_t>(want, 16)));
    *have = want;
    return 0;
}

void device_release(DeviceState* d) {
    if (!d) return;
    d->s32.release(); d->s64.release();
    if (d->partial) (void)hipFree(d->partial);
    if (d->job_counter) (void)hipFree(d->job_counter);
    if (d->packed) (void)hipFree(d->packed);
    if (d->linear) (void)hipFree(d->linear);
    if (d->rgba) (void)hipFree(d->rgba);
    if (d->ev0) (void)hipEventDestroy(d->ev0);
    if (d->ev1) (void)hipEventDestroy(d->ev1);
    delete d;
}

int device_commit(::rttnw_scene* s, std::string& err) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        err = "no HIP device available (this library has no CPU fallback)";
        return RTTNW_ERR_HIP;
    }
    DeviceState* d = new DeviceState();
    s->device = d;
    hipError_t e = hipGetDevice(&d->device);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, d->device);
    if (e != hipSuccess) { err = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e); return RTTNW_ERR_HIP; }
    d->num_cus = prop.multiProcessorCount;
    e = hipMalloc((void**)&d->job_counter, sizeof(unsigned long long) + sizeof(DeviceCounters));
    if (e == hipSuccess) e = hipEventCreate(&d->ev0);
    if (e == hipSuccess) e = hipEventCreate(&d->ev1);
    if (e != hipSuccess) { err = std::string("device_commit: ") + hipGetErrorString(e); return RTTNW_ERR_HIP; }
    // The scene arrays are uploaded per precision on first use (render), see ensure_scene().
    return 0;
}

template <typename R> DeviceScene<R>& scene_of(DeviceState* d);
template <> DeviceScene<float>& scene_of<float>(DeviceState* d) { return d->s32; }
template <> DeviceScene<double>& scene_of<double>(DeviceState* d) { return d->s64; }

template <typename R> CameraRec<R> narrow_camera(const CameraRec<double>& c) {
    CameraRec<R> o;
    for (int k = 0; k < 3; ++k) {
        o.origin[k] = R(c.origin[k]); o.lower_left_corner[k] = R(c.lower_left_corner[k]);
        o.horizontal[k] = R(c.horizontal[k]); o.vertical[k] = R(c.vertical[k]); o.u[k] = R(c.u[k]); o.v[k] = R(c.v[k]);
    }
    o.lens_radius = R(c.lens_radius); o.open_time = R(c.open_time); o.close_time = R(c.close_time);
    return o;
}

static void fill_layout(uint32_t w, uint32_t h, uint32_t world, rttnw_tile_layout& L) {
    L.tiles_x = (w + 7) / 8; L.tiles_y = (h + 7) / 8;
    L.n_tiles = L.tiles_x * L.tiles_y;
    L.tiles_per_rank = (L.n_tiles + world - 1) / world;
    L.pixels_per_rank = L.tiles_per_rank * 64;
}

static int validate(::rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p) {
    if (!s || !cam || !p) { set_last_error("render: NULL argument"); return RTTNW_ERR_INVALID; }
    if (!s->committed || !s->device) { set_last_error("render: scene is not committed"); return RTTNW_ERR_STATE; }
    if (!p->width || !p->height || !p->spp || !p->max_depth) { set_last_error("render: empty image, spp or depth"); return RTTNW_ERR_INVALID; }
    if (p->precision != RTTNW_F32 && p->precision != RTTNW_F64) { set_last_error("render: bad precision"); return RTTNW_ERR_INVALID; }
    if (p->tile_world == 0 || p->tile_rank >= p->tile_world) { set_last_error("render: bad tile_rank / tile_world"); return RTTNW_ERR_INVALID; }
    // moving-sphere bounds are built for the shutter interval [0,1] (BvhTree::from, hittable.rs:256)
    if (cam->open_time < 0.0 || cam->close_time > 1.0 || cam->open_time > cam->close_time) {
        set_last_error("render: shutter interval must lie inside [0,1]");
        return RTTNW_ERR_UNSUPPORTED;
    }
    return 0;
}

template <typename R>
int render_tiles_t(::rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, void* d_packed, hipStream_t stream,
                   rttnw_stats* stats) {
    DeviceState* d = s->device;
    HIP_TRY(hipSetDevice(d->device));
    DeviceScene<R>& ds = scene_of<R>(d);
    if (!ds.ready)
        if (int rc = ds.upload(s->flat)) return rc;

    rttnw_tile_layout L;
    fill_layout(p->width, p->height, p->tile_world, L);
    RenderConsts rc{};
    rc.width = p->width; rc.height = p->height; rc.spp = p->spp; rc.max_depth = p->max_depth;
    rc.spp_chunk = p->spp_chunk ? p->spp_chunk : (p->spp + 63u) / 64u; // default: <= 64 chunks per pixel (fine jobs: short tail)
    rc.n_chunks = (p->spp + rc.spp_chunk - 1) / rc.spp_chunk;
    rc.tiles_x = L.tiles_x; rc.tiles_y = L.tiles_y; rc.n_tiles = L.n_tiles;
    rc.tile_rank = p->tile_rank; rc.tile_world = p->tile_world;
    rc.my_tiles = L.n_tiles > p->tile_rank ? (L.n_tiles - p->tile_rank + p->tile_world - 1) / p->tile_world : 0;
    rc.quirks = p->quirks; rc.seed = p->seed; rc.stack_depth = s->flat.stack_depth;
    if (const char* e = getenv("RTTNW_DEBUG_STACK_EXTRA")) rc.stack_depth += uint32_t(atoi(e));
    const size_t n_jobs = size_t(rc.my_tiles) * 64 * rc.n_chunks;
    if (int g = grow(&d->partial, &d->partial_bytes, std::max<size_t>(n_jobs, 1) * 3 * sizeof(R))) return g;

    CameraRec<double> cam64;
    make_camera(cam->lookfrom, cam->lookat, cam->view_up, cam->vertical_fov, cam->aspect_ratio, cam->aperture,
                cam->focus_distance, cam->open_time, cam->close_time, cam64);
    const CameraRec<R> camr = narrow_camera<R>(cam64);

    const bool count = p->collect_counters != 0;
    auto kernel = count ? trace_kernel<R, true> : trace_kernel<R, false>;
    const size_t lds_bytes = size_t(rc.stack_depth) * TRACE_BLOCK * sizeof(int32_t);
    if (lds_bytes > 160 * 1024) { set_last_error("render: traversal stack does not fit in LDS"); return RTTNW_ERR_UNSUPPORTED; }
    HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
    int blocks_per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, (const void*)kernel, TRACE_BLOCK, lds_bytes));
    blocks_per_cu = std::max(1, std::min(blocks_per_cu, 8));
    // persistent grid: what the chip holds at once (no inter-workgroup dependency, so a little
    // over-subscription is harmless), but never more waves than jobs/64
    const size_t waves_needed = (n_jobs + 63) / 64;
    size_t grid = std::min<size_t>(size_t(d->num_cus) * blocks_per_cu, (waves_needed + 3) / 4);
    grid = std::max<size_t>(grid, 1);

    HIP_TRY(hipMemsetAsync(d->job_counter, 0, sizeof(unsigned long long) + sizeof(DeviceCounters), stream));
    DeviceCounters* dc = reinterpret_cast<DeviceCounters*>(d->job_counter + 1);
    if (stats) HIP_TRY(hipEventRecord(d->ev0, stream));
    if (n_jobs > 0) {
        hipLaunchKernelGGL(kernel, dim3(uint32_t(grid)), dim3(TRACE_BLOCK), lds_bytes, stream, ds.view, camr, rc, R(p->background[0]),
                           R(p->background[1]), R(p->background[2]), R(p->t_min), (R*)d->partial, d->job_counter, dc);
        HIP_TRY(hipGetLastError());
    }
    if (stats) HIP_TRY(hipEventRecord(d->ev1, stream));
    hipLaunchKernelGGL(resolve_kernel<R>, dim3((L.pixels_per_rank + 255) / 256), dim3(256), 0, stream, (const R*)d->partial,
                       (R*)d_packed, rc, L.pixels_per_rank);
    HIP_TRY(hipGetLastError());

    if (stats) {
        HIP_TRY(hipStreamSynchronize(stream));
        std::memset(stats, 0, sizeof(*stats));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, d->ev0, d->ev1));
        stats->kernel_ms = ms;
        // samples traced by this rank: pixels of its tiles that lie inside the image
        uint64_t px_count = 0;
        for (uint32_t t = 0; t < rc.my_tiles; ++t) {
            uint32_t tx, ty;
            tile_unpermute(rc.tile_rank + t * rc.tile_world, rc.tiles_x, tx, ty);
            uint32_t w = std::min(8u, rc.width - tx * 8), h = std::min(8u, rc.height - ty * 8);
            px_count += uint64_t(w) * h;
        }
        stats->samples = px_count * rc.spp;
        if (count) {
            DeviceCounters hc;
            HIP_TRY(hipMemcpy(&hc, dc, sizeof(hc), hipMemcpyDeviceToHost));
            stats->rays = hc.rays; stats->nodes_visited = hc.nodes; stats->prims_tested = hc.prims; stats->texel_fetches = hc.texels;
            if (getenv("RTTNW_DEBUG_SCHED")) {
                static const char* names[] = {"NODE", "SPHERE", "BOX", "MISC", "POST", "NEW"};
                unsigned long long te = 0;
                for (int k = 0; k < 6; ++k) te += hc.stage_execs[k];
                for (int k = 0; k < 6; ++k)
                    fprintf(stderr, "[sched] %-6s execs %12llu (%.1f%%)  lanes/exec %.2f  lane-steps/sample %.2f\n", names[k], hc.stage_execs[k],
                            100.0 * hc.stage_execs[k] / double(te ? te : 1), hc.stage_execs[k] ? double(hc.stage_lanes[k]) / hc.stage_execs[k] : 0.0,
                            double(hc.stage_lanes[k]) / double(stats->samples ? stats->samples : 1));
                fprintf(stderr, "[sched] waves %llu mean final thresholds:", hc.waves);
                for (int k = 0; k < 6; ++k) fprintf(stderr, " %s %.1f", names[k], double(hc.thr_sum[k]) / double(hc.waves ? hc.waves : 1));
                fprintf(stderr, "\n[sched] time-average queue population:");
                double tp = 0;
                for (int k = 0; k < 6; ++k) { fprintf(stderr, " %s %.1f", names[k], double(hc.pop_sum[k]) / double(te ? te : 1)); tp += double(hc.pop_sum[k]) / double(te ? te : 1); }
                fprintf(stderr, "  total live %.1f\n", tp);
            }
        }
        stats->n_nodes = uint32_t(s->flat.nodes.size());
        stats->n_prims = s->flat.n_prims_in_bvh;
        stats->scene_bytes = uint32_t(std::min<size_t>(ds.bytes, 0xFFFFFFFFu));
    }
    return RTTNW_OK;
}

template <typename R>
int probe_path_t(::rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, uint32_t px, uint32_t row, uint32_t sample,
                 double* out, uint32_t max_out) {
    DeviceState* d = s->device;
    HIP_TRY(hipSetDevice(d->device));
    DeviceScene<R>& ds = scene_of<R>(d);
    if (!ds.ready)
        if (int rc = ds.upload(s->flat)) return rc;
    RenderConsts rc{};
    rc.width = p->width; rc.height = p->height; rc.spp = p->spp; rc.max_depth = p->max_depth;
    rc.quirks = p->quirks; rc.seed = p->seed; rc.stack_depth = s->flat.stack_depth;
    CameraRec<double> cam64;
    make_camera(cam->lookfrom, cam->lookat, cam->view_up, cam->vertical_fov, cam->aspect_ratio, cam->aperture,
                cam->focus_distance, cam->open_time, cam->close_time, cam64);
    double* d_out = nullptr;
    int32_t* d_n = nullptr;
    HIP_TRY(hipMalloc((void**)&d_out, (size_t(max_out) * PROBE_STRIDE + 4) * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&d_n, sizeof(int32_t)));
    HIP_TRY(hipMemset(d_n, 0, sizeof(int32_t)));
    const size_t lds = size_t(rc.stack_depth) * 64 * sizeof(int32_t);
    hipLaunchKernelGGL(probe_path_kernel<R>, dim3(1), dim3(64), lds, 0, ds.view, narrow_camera<R>(cam64), rc, R(p->t_min), px, row,
                       sample, d_out, max_out, d_n);
    HIP_TRY(hipGetLastError());
    int32_t n = 0;
    HIP_TRY(hipMemcpy(&n, d_n, sizeof(n), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out, d_out, size_t(n) * PROBE_STRIDE * sizeof(double), hipMemcpyDeviceToHost));
    // radiance of the sample (path_step loop) is returned after the last possible bounce record
    HIP_TRY(hipMemcpy(out + size_t(max_out) * PROBE_STRIDE, d_out + size_t(max_out) * PROBE_STRIDE, 4 * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(d_out); (void)hipFree(d_n);
    return n;
}

} // namespace rt

// =============================================================================================
extern "C" {

int rttnw_debug_probe_path(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, uint32_t px, uint32_t row,
                           uint32_t sample, double* out, uint32_t max_out) {
    if (int rc = rt::validate(s, cam, p)) return rc;
    if (!out || px >= p->width || row >= p->height) { rt::set_last_error("debug_probe_path: bad arguments"); return RTTNW_ERR_INVALID; }
    return p->precision == RTTNW_F32 ? rt::probe_path_t<float>(s, cam, p, px, row, sample, out, max_out)
                                     : rt::probe_path_t<double>(s, cam, p, px, row, sample, out, max_out);
}


int rttnw_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rttnw_tile_layout_get(uint32_t width, uint32_t height, uint32_t world, rttnw_tile_layout* out) {
    if (!out || !width || !height || !world) { rt::set_last_error("tile_layout_get: bad arguments"); return RTTNW_ERR_INVALID; }
    rt::fill_layout(width, height, world, *out);
    return RTTNW_OK;
}

int rttnw_scene_info(rttnw_scene* s, rttnw_stats* out) {
    if (!s || !out || !s->committed) { rt::set_last_error("scene_info: scene not committed"); return RTTNW_ERR_STATE; }
    std::memset(out, 0, sizeof(*out));
    out->n_nodes = uint32_t(s->flat.nodes.size());
    out->n_prims = s->flat.n_prims_in_bvh;
    const auto& f = s->flat;
    size_t b32 = f.nodes.size() * sizeof(rt::BvhNode) + f.spheres.size() * sizeof(rt::SphereRec<float>) +
                 f.moving.size() * sizeof(rt::MovingSphereRec<float>) + f.rects.size() * sizeof(rt::RectRec<float>) +
                 f.boxes.size() * sizeof(rt::BoxRec<float>) + f.insts.size() * sizeof(rt::InstanceRec<float>);
    out->scene_bytes = uint32_t(std::min<size_t>(b32, 0xFFFFFFFFu));
    out->reserved = f.stack_depth;
    return RTTNW_OK;
}

int rttnw_render_tiles_device(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, void* d_packed, void* hip_stream,
                              rttnw_stats* stats) {
    if (int rc = rt::validate(s, cam, p)) return rc;
    if (!d_packed) { rt::set_last_error("render_tiles_device: d_packed is NULL"); return RTTNW_ERR_INVALID; }
    hipStream_t stream = (hipStream_t)hip_stream;
    return p->precision == RTTNW_F32 ? rt::render_tiles_t<float>(s, cam, p, d_packed, stream, stats)
                                     : rt::render_tiles_t<double>(s, cam, p, d_packed, stream, stats);
}

int rttnw_untile_device(uint32_t width, uint32_t height, uint32_t world, uint32_t precision, const void* d_gathered,
                        void* d_linear_rgb, uint8_t* d_rgba8, void* hip_stream) {
    if (!width || !height || !world || !d_gathered) { rt::set_last_error("untile_device: bad arguments"); return RTTNW_ERR_INVALID; }
    rttnw_tile_layout L;
    rt::fill_layout(width, height, world, L);
    dim3 block(32, 8), grid((width + 31) / 32, (height + 7) / 8);
    hipStream_t stream = (hipStream_t)hip_stream;
    if (precision == RTTNW_F32)
        hipLaunchKernelGGL(rt::untile_kernel<float>, grid, block, 0, stream, (const float*)d_gathered, (float*)d_linear_rgb, d_rgba8,
                           width, height, L.tiles_x, world, L.pixels_per_rank);
    else
        hipLaunchKernelGGL(rt::untile_kernel<double>, grid, block, 0, stream, (const double*)d_gathered, (double*)d_linear_rgb,
                           d_rgba8, width, height, L.tiles_x, world, L.pixels_per_rank);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rt::set_last_error(std::string("untile_kernel: ") + hipGetErrorString(e)); return RTTNW_ERR_HIP; }
    return RTTNW_OK;
}

int rttnw_render(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, double* out_linear_rgb, uint8_t* out_rgba8,
                 rttnw_stats* stats) {
    if (int rc = rt::validate(s, cam, p)) return rc;
    if (p->tile_world != 1) { rt::set_last_error("render: host-output form needs tile_world == 1"); return RTTNW_ERR_INVALID; }
    rt::DeviceState* d = s->device;
    rttnw_tile_layout L;
    rt::fill_layout(p->width, p->height, 1, L);
    const size_t rsz = p->precision == RTTNW_F32 ? sizeof(float) : sizeof(double);
    const size_t npx = size_t(p->width) * p->height;
    if (int g = rt::grow(&d->packed, &d->packed_bytes, size_t(L.pixels_per_rank) * 4 * rsz)) return g;
    if (int g = rt::grow(&d->linear, &d->linear_bytes, npx * 3 * rsz)) return g;
    if (int g = rt::grow((void**)&d->rgba, &d->rgba_bytes, npx * 4)) return g;
    int rc = rttnw_render_tiles_device(s, cam, p, d->packed, nullptr, stats);
    if (rc) return rc;
    rc = rttnw_untile_device(p->width, p->height, 1, p->precision, d->packed, d->linear, d->rgba, nullptr);
    if (rc) return rc;
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { rt::set_last_error(std::string("render: ") + hipGetErrorString(e)); return RTTNW_ERR_HIP; }
    if (out_rgba8) {
        e = hipMemcpy(out_rgba8, d->rgba, npx * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rt::set_last_error(std::string("render: ") + hipGetErrorString(e)); return RTTNW_ERR_HIP; }
    }
    if (out_linear_rgb) {
        if (p->precision == RTTNW_F64) {
            e = hipMemcpy(out_linear_rgb, d->linear, npx * 3 * sizeof(double), hipMemcpyDeviceToHost);
        } else {
            std::vector<float> tmp(npx * 3);
            e = hipMemcpy(tmp.data(), d->linear, npx * 3 * sizeof(float), hipMemcpyDeviceToHost);
            for (size_t i = 0; i < npx * 3; ++i) out_linear_rgb[i] = double(tmp[i]);
        }
        if (e != hipSuccess) { rt::set_last_error(std::string("render: ") + hipGetErrorString(e)); return RTTNW_ERR_HIP; }
    }
    return RTTNW_OK;
}

} // extern "C"
