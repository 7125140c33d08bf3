// render_api.cpp — the extern "C" half of the device side of include/rttnw_hip.h: device state, render entry points,
// rttnw_render_multi (per-device streams, RCCL gather).  Host code only; the kernels and their launch code live in
// render_f32.hip / render_f64.hip (render_common.hpp says why there are two).
// No CPU fallback: every entry point needs a HIP device.
#include "render_common.hpp"
#include "bvh_build.hpp"

#include <rccl/rccl.h>
#include <dlfcn.h>
#include <mutex>

namespace rt {

// One precision's launch code, by rttnw_params::precision (RTTNW_F64_STRICT: the ieee_strict build of the f64 arithmetic, rt_core.hpp).
static int render_tiles_any(::rttnw_scene* s, DeviceState* d, const rttnw_camera_desc* cam, const rttnw_params* p, void* d_packed, hipStream_t stream,
                            rttnw_stats* stats, bool sync_for_stats = true, bool prepare_only = false) {
    if (p->precision == RTTNW_F32) return render_tiles_t<float>(s, d, cam, p, d_packed, stream, stats, sync_for_stats, prepare_only);
    if (p->precision == RTTNW_F64_STRICT) return ieee_strict::render_tiles_t<double>(s, d, cam, p, d_packed, stream, stats, sync_for_stats, prepare_only);
    return render_tiles_t<double>(s, d, cam, p, d_packed, stream, stats, sync_for_stats, prepare_only);
}

int grow(void** p, size_t* have, size_t want) {
    if (*have >= want && *p) return 0;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *have = 0;
    HIP_TRY(hipMalloc(p, std::max<size_t>(want, 16)));
    *have = want;
    return 0;
}

void device_release(DeviceState* d) {
    if (!d) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (d->device >= 0) (void)hipSetDevice(d->device);
    struct Restore { int dev; ~Restore() { if (dev >= 0) (void)hipSetDevice(dev); } } restore{prev};
    d->s32.release(); d->s64.release(); d->s64_ref.release();
    if (d->partial) (void)hipFree(d->partial);
    if (d->pool_r) (void)hipFree(d->pool_r);
    if (d->pool_u) (void)hipFree(d->pool_u);
    if (d->spill) (void)hipFree(d->spill);
    if (d->job_counter) (void)hipFree(d->job_counter);
    if (d->packed) (void)hipFree(d->packed);
    if (d->linear) (void)hipFree(d->linear);
    if (d->rgba) (void)hipFree(d->rgba);
    if (d->multi_packed) (void)hipFree(d->multi_packed);
    if (d->gathered) (void)hipFree(d->gathered);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    if (d->ev0) (void)hipEventDestroy(d->ev0);
    if (d->ev1) (void)hipEventDestroy(d->ev1);
    delete d;
}

int device_bvh_builder(::rttnw_scene* s, DeviceBvhApi& out, std::string& err) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        err = "no HIP device available (the device BVH builder has no CPU fallback)";
        return RTTNW_ERR_HIP;
    }
    out.build = [s](const BuildPrim* prims, size_t n, const float* centroid_bounds, DeviceTree& tree, std::string& e) {
        return lbvh_build_device_tree(prims, n, centroid_bounds, s->bvh_builder != RTTNW_BVH_DEVICE_LBVH, tree, &s->build_kernel_ms, e);
    };
    out.rebase = [](DeviceTree& tree, uint32_t base4, uint32_t base2, std::string& e) { return device_tree_rebase(tree, base4, base2, e); };
    out.min_leaves = s->bvh_builder == RTTNW_BVH_AUTO ? RTTNW_BVH_AUTO_DEVICE_LEAVES : 2u;
    return 0;
}

int materialize_host_nodes(FlatScene& f, std::string& err) {
    if (f.device_trees.empty() || f.nodes4.size() == f.total_nodes4()) return 0;
    f.nodes4.resize(f.total_nodes4());
    f.nodes.resize(f.total_nodes2());
    for (const DeviceTree& t : f.device_trees)
        if (device_tree_download(t, f.nodes4.data() + t.base4, f.nodes.data() + t.base2, err)) {
            f.nodes4.resize(f.n_host4);
            f.nodes.resize(f.n_host2);
            return RTTNW_ERR_HIP;
        }
    return 0;
}

// State on the CURRENT device (job counter, events; the scene arrays follow on first use).
int device_state_create(DeviceState*& out, std::string& err) {
    DeviceState* d = new DeviceState();
    hipError_t e = hipGetDevice(&d->device);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, d->device);
    if (e != hipSuccess) { err = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e); device_release(d); return RTTNW_ERR_HIP; }
    d->num_cus = prop.multiProcessorCount;
    d->chunk_budget = std::min<uint64_t>(24ull << 30, std::max<uint64_t>(4ull << 30, uint64_t(prop.totalGlobalMem) / 12));
    e = hipMalloc((void**)&d->job_counter, sizeof(unsigned long long) + sizeof(DeviceCounters));
    if (e == hipSuccess) e = hipEventCreate(&d->ev0);
    if (e == hipSuccess) e = hipEventCreate(&d->ev1);
    if (e != hipSuccess) { err = std::string("device state: ") + hipGetErrorString(e); device_release(d); return RTTNW_ERR_HIP; }
    out = d;
    return 0;
}

int device_commit(::rttnw_scene* s, std::string& err) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        err = "no HIP device available (this library has no CPU fallback)";
        return RTTNW_ERR_HIP;
    }
    if (s->device) { device_release(s->device); s->device = nullptr; } // a commit that failed half-way and is retried
    // The scene arrays are uploaded per precision on first use (render), see render_tiles_t.
    return device_state_create(s->device, err);
}


// RTTNW_F64_STRICT promises the reference's operations in the reference's order.  The default lowering breaks that promise in one place: a
// sphere under Translate / YRotate wrappers is tested as a world-space copy in the top tree (scene_lower.cpp: the same quadratic written in
// another frame — t differs in the last place, which a few bounces off small spheres amplify: final_scene 800x800 spp 1000, 9 of 1536 pixels of
// the cluster crop beyond 1e-9).  So the strict build walks a second lowering of the same graph that leaves those spheres in their groups'
// trees (the ray goes through the wrappers as in hittable.rs:599-606,686-699): on that crop every pixel is within 1.5e-13 of the CPU
// restatement, at 7 % of the strict build's speed on that scene (more node steps, a second tree).  Made at the first strict render of a scene
// that has such copies, with the scene's own builder and shutter interval; scenes without them render `flat` itself.
int reference_frame_scene(::rttnw_scene* s, const FlatScene*& flat) {
    flat = &s->flat;
    if (s->flat.n_world_copies == 0) return 0;
    std::lock_guard<std::mutex> lock(s->rebuild_mutex);
    if (!s->flat_ref) {
        std::string err;
        DeviceBvhApi device_builder;
        bool on_device = s->bvh_builder != RTTNW_BVH_HOST_SAH;
        if (on_device)
            if (int brc = device_bvh_builder(s, device_builder, err)) {
                // (as rttnw_scene_commit, capi_builder.cpp: RTTNW_BVH_AUTO falls back to the host builder; an explicitly requested device builder fails)
                if (s->bvh_builder != RTTNW_BVH_AUTO) { set_last_error(err); return brc; }
                on_device = false;
            }
        std::unique_ptr<FlatScene> ref(new FlatScene());
        // RTTNW_STRICT_GROUP_TREES=1 (experiments / tests): the round-4 form — the copies stay in their groups' trees and the walk enters them
        const char* gt = getenv("RTTNW_STRICT_GROUP_TREES");
        const int ref_mode = gt && gt[0] == '1' ? 0 : 2;
        if (int rc = lower_scene(s->graph, *ref, err, on_device ? &device_builder : nullptr, s->flat.time0, s->flat.time1, ref_mode)) { set_last_error(err); return rc; }
        s->flat_ref = std::move(ref);
    }
    flat = s->flat_ref.get();
    return 0;
}

void fill_layout(uint32_t w, uint32_t h, uint32_t world, rttnw_tile_layout& L) {
    L.tiles_x = (w + 7) / 8; L.tiles_y = (h + 7) / 8;
    L.n_tiles = L.tiles_x * L.tiles_y;
    L.tiles_per_rank = (L.n_tiles + world - 1) / world;
    L.pixels_per_rank = L.tiles_per_rank * 64;
}

int validate(::rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p) {
    if (!s || !cam || !p) { set_last_error("render: NULL argument"); return RTTNW_ERR_INVALID; }
    if (!s->committed || !s->device) { set_last_error("render: scene is not committed"); return RTTNW_ERR_STATE; }
    if (!p->width || !p->height || !p->spp || !p->max_depth) { set_last_error("render: empty image, spp or depth"); return RTTNW_ERR_INVALID; }
    if (p->precision != RTTNW_F32 && p->precision != RTTNW_F64 && p->precision != RTTNW_F64_STRICT) { set_last_error("render: bad precision"); return RTTNW_ERR_INVALID; }
    // (the f32 lane-owns-path kernel's folded slab test, rt_core.hpp SLAB_FMA_FOLDED, is conservative for t_min >= 0 only; a
    // negative or NaN t_min has no meaning in main.rs:33 either)
    if (!(p->t_min >= 0.0) || !(p->t_min < 1e300)) { set_last_error("render: t_min must be finite and >= 0"); return RTTNW_ERR_INVALID; }
    if (p->tile_world == 0 || p->tile_rank >= p->tile_world) { set_last_error("render: bad tile_rank / tile_world"); return RTTNW_ERR_INVALID; }
    // the decoupled kernel packs a pixel as px | row << 16, and the free-flight draw of medium m uses RNG slot m < 16
    if (p->width > 65535u || p->height > 65535u) { set_last_error("render: width and height are limited to 65535"); return RTTNW_ERR_UNSUPPORTED; }
    if (s->flat.media.size() > SLOT_DIELECTRIC) { set_last_error("render: more than 16 constant media"); return RTTNW_ERR_UNSUPPORTED; }
    if (!(cam->open_time <= cam->close_time)) { set_last_error("render: open_time > close_time"); return RTTNW_ERR_INVALID; }
    // The boxes of moving spheres are built for the shutter interval [0, 1] (what BvhTree::from uses, hittable.rs:256).  A
    // camera whose shutter reaches outside it (BvhTree::from_time, hittable.rs:261) makes the library rebuild the trees
    // for the wider interval, once, and drop the device copies (they are uploaded again on use).  This is the ONE change a
    // committed scene can undergo (include/rttnw_hip.h says so): it happens under the scene's mutex, before anything of this
    // call is enqueued, and rttnw_scene_build_info reports the rebuilt trees afterwards.  "One render in flight per scene"
    // (the header's rule) is what keeps a concurrent render from seeing the swap.
    {
        std::lock_guard<std::mutex> lock(s->rebuild_mutex);
        if (!s->flat.moving.empty() && (cam->open_time < s->flat.time0 || cam->close_time > s->flat.time1)) {
            const double t0 = std::min(s->flat.time0, cam->open_time), t1 = std::max(s->flat.time1, cam->close_time);
            std::string err;
            DeviceBvhApi device_builder;
            const bool on_device = s->bvh_builder != RTTNW_BVH_HOST_SAH;
            if (on_device)
                if (int brc = device_bvh_builder(s, device_builder, err)) { set_last_error(err); return brc; }
            FlatScene wider;
            const auto tb = std::chrono::steady_clock::now();
            s->build_kernel_ms = 0;
            if (int rc = lower_scene(s->graph, wider, err, on_device ? &device_builder : nullptr, t0, t1)) { set_last_error(err); return rc; }
            s->lower_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb).count();
            int prev = -1;
            (void)hipGetDevice(&prev);
            std::vector<DeviceState*> all = s->more_devices;
            all.push_back(s->device);
            for (DeviceState* d : all) { // nothing of an earlier render may still read the arrays that are about to go
                (void)hipSetDevice(d->device);
                (void)hipDeviceSynchronize();
                d->s32.release(); d->s64.release(); d->s64_ref.release();
            }
            if (prev >= 0) (void)hipSetDevice(prev);
            s->flat = std::move(wider);
            s->flat_ref.reset(); // (made again, for the wider interval, by the next RTTNW_F64_STRICT render)
        }
    }
    return 0;
}

} // namespace rt

// =============================================================================================
extern "C" {

int rttnw_debug_probe_path(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p, uint32_t px, uint32_t row,
                           uint32_t sample, double* out, uint32_t max_out) {
    if (int rc = rt::validate(s, cam, p)) return rc;
    if (!out || px >= p->width || row >= p->height) { rt::set_last_error("debug_probe_path: bad arguments"); return RTTNW_ERR_INVALID; }
    if (p->precision == RTTNW_F64_STRICT) return rt::ieee_strict::probe_path_t<double>(s, cam, p, px, row, sample, out, max_out);
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
    out->n_nodes = s->flat.total_nodes4();
    out->n_prims = s->flat.n_prims_in_bvh;
    const auto& f = s->flat;
    size_t b32 = size_t(f.total_nodes4()) * sizeof(rt::Bvh4Node) + f.spheres.size() * sizeof(rt::SphereRec<float>) +
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
    return rt::render_tiles_any(s, s->device, cam, p, d_packed, stream, stats);
}

int rttnw_untile_device(uint32_t width, uint32_t height, uint32_t world, uint32_t precision, const void* d_gathered,
                        void* d_linear_rgb, uint8_t* d_rgba8, void* hip_stream) {
    if (!width || !height || !world || !d_gathered || (precision != RTTNW_F32 && precision != RTTNW_F64 && precision != RTTNW_F64_STRICT)) {
        rt::set_last_error("untile_device: bad arguments");
        return RTTNW_ERR_INVALID;
    }
    hipStream_t stream = (hipStream_t)hip_stream;
    return precision == RTTNW_F32 ? rt::untile_launch<float>(width, height, world, d_gathered, d_linear_rgb, d_rgba8, stream)
                                  : rt::untile_launch<double>(width, height, world, d_gathered, d_linear_rgb, d_rgba8, stream);
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
        if (p->precision != RTTNW_F32) {
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

// ---------------------------------------------------------------------------------------------
// rttnw_render_multi: the whole of main.rs:202-229 on the GPUs of one node, in one call.
// ---------------------------------------------------------------------------------------------
namespace rt {
// RCCL, bound at first use (a single device, or logical ranks that share one device, never touch it).
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& err) {
        if (lib) return true;
        lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) { err = std::string("cannot load RCCL: ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Send = (decltype(Send))dlsym(lib, "ncclSend");
        Recv = (decltype(Recv))dlsym(lib, "ncclRecv");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) { err = "RCCL lacks an expected entry point"; return false; }
        return true;
    }
};
static Rccl g_rccl;
struct MultiComms { // one communicator set per distinct list of devices, kept until rttnw_shutdown() / process exit
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    std::mutex in_use; // RCCL allows ONE thread at a time to issue operations on a communicator: held from GroupStart to GroupEnd
};
static std::vector<MultiComms*> g_comms;
static std::mutex g_comms_mutex; // the list itself (rttnw_render_multi may be called from several host threads, each with its own scene)
static bool g_comms_atexit = false;

// Destroy the cached communicator sets (rttnw_shutdown; registered with atexit at the first set-up, so that a process that
// never calls it still leaves RCCL in order — before the HIP runtime's own teardown, which atexit runs later: LIFO).
static void destroy_comms() {
    std::lock_guard<std::mutex> lock(g_comms_mutex);
    // (at exit, under a host that tears its own HIP context down first — Python with torch —, the runtime may already be finalising: then
    // only the host structures are released; RCCL's own teardown has nothing left to talk to)
    int n_dev = 0;
    const bool runtime_alive = hipGetDeviceCount(&n_dev) == hipSuccess && n_dev > 0;
    for (MultiComms* c : g_comms) {
        std::lock_guard<std::mutex> use(c->in_use);
        for (ncclComm_t comm : c->comms)
            if (comm && g_rccl.CommDestroy && runtime_alive) (void)g_rccl.CommDestroy(comm);
        c->comms.clear();
    }
    for (MultiComms* c : g_comms) delete c;
    g_comms.clear();
}

static DeviceState* state_on(::rttnw_scene* s, int device, std::string& err) {
    if (s->device && s->device->device == device) return s->device;
    for (DeviceState* d : s->more_devices)
        if (d->device == device) return d;
    if (hipSetDevice(device) != hipSuccess) { err = "hipSetDevice failed"; return nullptr; }
    DeviceState* d = nullptr;
    if (device_state_create(d, err)) return nullptr;
    s->more_devices.push_back(d);
    return d;
}
} // namespace rt

extern "C" int rttnw_render_multi(rttnw_scene* s, const rttnw_camera_desc* cam, const rttnw_params* p_in, uint32_t ngpu, const int32_t* device_ids,
                                  double* out_linear_rgb, uint8_t* out_rgba8, rttnw_stats* stats) {
    using namespace rt;
    if (!p_in || !ngpu || ngpu > 64 || !device_ids) { set_last_error("render_multi: bad arguments"); return RTTNW_ERR_INVALID; }
    rttnw_params p = *p_in;
    p.tile_rank = 0; p.tile_world = ngpu;
    if (int rc = validate(s, cam, &p)) return rc;
    const int n_dev = rttnw_device_count();
    for (uint32_t r = 0; r < ngpu; ++r)
        if (device_ids[r] < 0 || device_ids[r] >= n_dev) { set_last_error("render_multi: no such device"); return RTTNW_ERR_INVALID; }
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    struct Restore { int dev; ~Restore() { if (dev >= 0) (void)hipSetDevice(dev); } } restore{prev_dev};

    std::string err;
    std::vector<DeviceState*> st(ngpu);
    std::vector<int> distinct; // devices in order of first appearance; distinct[0] = the root's (rank 0's) device
    std::vector<uint32_t> slot(ngpu), per_dev;
    for (uint32_t r = 0; r < ngpu; ++r) {
        st[r] = state_on(s, device_ids[r], err);
        if (!st[r]) { set_last_error("render_multi: " + err); return RTTNW_ERR_HIP; }
        size_t k = 0;
        while (k < distinct.size() && distinct[k] != device_ids[r]) ++k;
        if (k == distinct.size()) { distinct.push_back(device_ids[r]); per_dev.push_back(0); }
        slot[r] = per_dev[k]++; // this rank's place among the ranks of its device
    }
    rttnw_tile_layout L;
    fill_layout(p.width, p.height, ngpu, L);
    const size_t rsz = p.precision == RTTNW_F32 ? sizeof(float) : sizeof(double);
    const size_t chunk = size_t(L.pixels_per_rank) * 4 * rsz, npx = size_t(p.width) * p.height;
    for (size_t k = 0; k < distinct.size(); ++k) {
        DeviceState* d = state_on(s, distinct[k], err);
        HIP_TRY(hipSetDevice(d->device));
        if (!d->stream) HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
        if (int g = grow(&d->multi_packed, &d->multi_packed_bytes, chunk * per_dev[k])) return g;
    }
    DeviceState* root = st[0];
    HIP_TRY(hipSetDevice(root->device));
    if (int g = grow(&root->gathered, &root->gathered_bytes, chunk * ngpu)) return g;
    if (int g = grow(&root->linear, &root->linear_bytes, npx * 3 * rsz)) return g;
    if (int g = grow((void**)&root->rgba, &root->rgba_bytes, npx * 4)) return g;

    // ---- first use: scene uploads and workspace growth for EVERY rank, before anything is launched (a hipMalloc or a
    // hipFree between two ranks' launches would synchronise its whole device)
    for (uint32_t r = 0; r < ngpu; ++r) {
        rttnw_params pr = p;
        pr.tile_rank = r;
        void* dst = (char*)st[r]->multi_packed + chunk * slot[r];
        int rc = render_tiles_any(s, st[r], cam, &pr, dst, st[r]->stream, nullptr, false, true);
        if (rc) return rc;
    }
    // ---- every rank traces its tiles, on its device's stream; ranks that share a device run one after the other
    std::vector<hipEvent_t> ev(size_t(ngpu) * 2, nullptr);
    struct EvFree { std::vector<hipEvent_t>& v; ~EvFree() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); } } ev_free{ev};
    for (uint32_t r = 0; r < ngpu; ++r) {
        DeviceState* d = st[r];
        HIP_TRY(hipSetDevice(d->device));
        rttnw_params pr = p;
        pr.tile_rank = r;
        void* dst = (char*)d->multi_packed + chunk * slot[r];
        HIP_TRY(hipEventCreate(&ev[2 * r]));
        HIP_TRY(hipEventCreate(&ev[2 * r + 1]));
        HIP_TRY(hipEventRecord(ev[2 * r], d->stream));
        int rc = render_tiles_any(s, d, cam, &pr, dst, d->stream, stats ? &stats[r] : nullptr, false);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(ev[2 * r + 1], d->stream));
    }

    // ---- gather to the root: a device-to-device copy for ranks on the root's device; for the others one of TWO transports over xGMI
    // (RTTNW_MULTI_GATHER=rccl|peer, default rccl):
    //   rccl  a grouped ncclSend / ncclRecv per rank buffer (north_star's "RCCL gather over xGMI"; communicators cached per device list);
    //   peer  hipMemcpyPeerAsync of each rank's packed tiles into the root's gather buffer on the RANK's stream, an event behind it that the
    //         root's stream waits for — 5 MB per rank, no library, no communicator.  Also what the call FALLS THROUGH to when RCCL cannot be
    //         loaded or ncclCommInitAll fails (one stderr line; rttnw_stats.reserved bit 9 of rank 0), so that a node whose RCCL is broken
    //         still renders.  Bit 8 of stats[0].reserved: the gather went through peer copies.
    // RTTNW_MULTI_FORCE_TRANSPORT=1 (tests; RTTNW_MULTI_FORCE_RCCL=1 is the older name): the ranks on the root's device travel through the
    // transport too — the root sending to itself — so that the dlopen'ed entry points, the communicator set-up, the peer copies, the stream
    // ordering and the error paths run on a box with ONE GPU as well.  RTTNW_MULTI_FAIL_RCCL=1 (tests): the RCCL set-up reports failure.
    const char* force_env = getenv("RTTNW_MULTI_FORCE_RCCL");
    const char* force_env2 = getenv("RTTNW_MULTI_FORCE_TRANSPORT");
    const bool force_rccl = (force_env && force_env[0] == '1') || (force_env2 && force_env2[0] == '1'); // (every rank through the transport)
    const char* gather_env = getenv("RTTNW_MULTI_GATHER");
    bool use_peer = gather_env && std::string(gather_env) == "peer";
    if (gather_env && !use_peer && std::string(gather_env) != "rccl") { set_last_error("render_multi: RTTNW_MULTI_GATHER must be rccl or peer"); return RTTNW_ERR_INVALID; }
    bool fell_back = false;
    const bool debug_multi = getenv("RTTNW_DEBUG_MULTI") != nullptr;
    if ((distinct.size() > 1 || force_rccl) && !use_peer) {
        std::unique_lock<std::mutex> comms_lock(g_comms_mutex);
        MultiComms* mc = nullptr;
        std::string why;
        const char* fail_env = getenv("RTTNW_MULTI_FAIL_RCCL");
        if (fail_env && fail_env[0] == '1') why = "RTTNW_MULTI_FAIL_RCCL=1";
        else if (!g_rccl.load(err)) why = err;
        if (why.empty()) {
            for (MultiComms* c : g_comms)
                if (c->devices == distinct) mc = c;
            if (!mc) {
                mc = new MultiComms();
                mc->devices = distinct;
                mc->comms.resize(distinct.size());
                ncclResult_t nr = g_rccl.CommInitAll(mc->comms.data(), int(distinct.size()), distinct.data());
                if (nr != ncclSuccess) {
                    why = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(nr);
                    delete mc;
                    mc = nullptr;
                } else {
                    g_comms.push_back(mc); // kept: communicator set-up costs ~100 ms; destroyed by rttnw_shutdown() or at exit
                    if (!g_comms_atexit) { g_comms_atexit = true; std::atexit(destroy_comms); }
                    if (debug_multi) fprintf(stderr, "[render_multi] RCCL communicators over %zu device(s)\n", distinct.size());
                }
            }
        }
        if (!mc) {
            // nothing has been sent yet: the peer transport does the same job
            fprintf(stderr, "[render_multi] RCCL gather unavailable (%s): gathering through peer copies\n", why.c_str());
            use_peer = fell_back = true;
        } else {
            // (the list's lock is released, the set's own is taken: two host threads rendering over the SAME devices take turns on
            // its communicators; threads over different device lists do not wait for each other)
            std::unique_lock<std::mutex> use(mc->in_use);
            comms_lock.unlock();
            ncclResult_t nr = g_rccl.GroupStart();
            uint32_t n_sent = 0;
            for (uint32_t r = 0; r < ngpu && nr == ncclSuccess; ++r) {
                size_t k = 0;
                while (distinct[k] != device_ids[r]) ++k;
                if (k == 0 && !force_rccl) continue; // on the root's device: copied below
                const void* src = (const char*)st[r]->multi_packed + chunk * slot[r];
                nr = g_rccl.Send(src, chunk, ncclChar, 0, mc->comms[k], st[r]->stream);
                if (nr == ncclSuccess) nr = g_rccl.Recv((char*)root->gathered + chunk * r, chunk, ncclChar, int(k), mc->comms[0], root->stream);
                ++n_sent;
            }
            ncclResult_t ne = g_rccl.GroupEnd();
            if (nr == ncclSuccess) nr = ne;
            if (nr != ncclSuccess) { set_last_error(std::string("RCCL gather: ") + g_rccl.GetErrorString(nr)); return RTTNW_ERR_HIP; }
            if (debug_multi) fprintf(stderr, "[render_multi] %u rank buffer(s) of %zu bytes through ncclSend / ncclRecv\n", n_sent, chunk);
        }
    }
    if ((distinct.size() > 1 || force_rccl) && use_peer) {
        // direct access root <- rank device where the fabric allows it (xGMI: every pair of a node); without it the copy is staged by the runtime
        for (size_t k = 1; k < distinct.size(); ++k) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, distinct[k], root->device) == hipSuccess && can) {
                HIP_TRY(hipSetDevice(distinct[k]));
                const hipError_t e = hipDeviceEnablePeerAccess(root->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_TRY(e);
                (void)hipGetLastError();
            }
        }
        std::vector<hipEvent_t> sent(ngpu, nullptr);
        struct SentFree { std::vector<hipEvent_t>& v; ~SentFree() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); } } sent_free{sent};
        uint32_t n_sent = 0;
        for (uint32_t r = 0; r < ngpu; ++r) {
            if (device_ids[r] == root->device && !force_rccl) continue; // on the root's device: copied below
            HIP_TRY(hipSetDevice(st[r]->device));
            const void* src = (const char*)st[r]->multi_packed + chunk * slot[r];
            HIP_TRY(hipMemcpyPeerAsync((char*)root->gathered + chunk * r, root->device, src, st[r]->device, chunk, st[r]->stream));
            HIP_TRY(hipEventCreateWithFlags(&sent[r], hipEventDisableTiming));
            HIP_TRY(hipEventRecord(sent[r], st[r]->stream));
            ++n_sent;
        }
        HIP_TRY(hipSetDevice(root->device));
        for (uint32_t r = 0; r < ngpu; ++r)
            if (sent[r]) HIP_TRY(hipStreamWaitEvent(root->stream, sent[r], 0)); // the un-tile below reads what the ranks' streams have written
        if (debug_multi) fprintf(stderr, "[render_multi] %u rank buffer(s) of %zu bytes through hipMemcpyPeerAsync\n", n_sent, chunk);
        // (the events are destroyed when this block ends: a recorded event may be destroyed while work waits on it — the wait was enqueued)
    }
    HIP_TRY(hipSetDevice(root->device));
    if (!force_rccl)
        for (uint32_t r = 0; r < ngpu; ++r)
            if (device_ids[r] == root->device)
                HIP_TRY(hipMemcpyAsync((char*)root->gathered + chunk * r, (const char*)root->multi_packed + chunk * slot[r], chunk, hipMemcpyDeviceToDevice, root->stream));
    int rc = rttnw_untile_device(p.width, p.height, ngpu, p.precision, root->gathered, root->linear, root->rgba, root->stream);
    if (rc) return rc;
    for (size_t k = 0; k < distinct.size(); ++k) {
        DeviceState* d = state_on(s, distinct[k], err);
        HIP_TRY(hipSetDevice(d->device));
        HIP_TRY(hipStreamSynchronize(d->stream));
    }
    HIP_TRY(hipSetDevice(root->device));
    if (stats)
        for (uint32_t r = 0; r < ngpu; ++r) {
            float ms = 0;
            HIP_TRY(hipSetDevice(st[r]->device));
            HIP_TRY(hipEventElapsedTime(&ms, ev[2 * r], ev[2 * r + 1]));
            stats[r].kernel_ms = ms; // trace + resolve of this rank
        }
    if (stats && use_peer && (distinct.size() > 1 || force_rccl)) stats[0].reserved |= 0x100u | (fell_back ? 0x200u : 0u);
    HIP_TRY(hipSetDevice(root->device));
    if (out_rgba8) HIP_TRY(hipMemcpy(out_rgba8, root->rgba, npx * 4, hipMemcpyDeviceToHost));
    if (out_linear_rgb) {
        if (p.precision != RTTNW_F32) {
            HIP_TRY(hipMemcpy(out_linear_rgb, root->linear, npx * 3 * sizeof(double), hipMemcpyDeviceToHost));
        } else {
            std::vector<float> tmp(npx * 3);
            HIP_TRY(hipMemcpy(tmp.data(), root->linear, npx * 3 * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < npx * 3; ++i) out_linear_rgb[i] = double(tmp[i]);
        }
    }
    return RTTNW_OK;
}

// Release what the library keeps for the life of the process (today: the RCCL communicator sets of rttnw_render_multi).  Scenes
// are the caller's (rttnw_scene_destroy).  Safe to call more than once and with renders finished; also runs at exit.
extern "C" void rttnw_shutdown(void) { rt::destroy_comms(); }
