// capi_builder.cpp — the scene-description half of include/rttnw_hip.h: each call records one node
// of the graph the reference would have built out of Box/Arc<dyn ...> objects (src/scenes.rs is
// the model caller).  Nothing is evaluated here; rttnw_scene_commit lowers and uploads.
#include "../../include/rttnw_hip.h"
#include "scene_handle.hpp"

#include <chrono>
#include <cmath>
#include <cstring>
#include <new>

namespace {
thread_local std::string g_last_error;

int fail(int code, const char* msg) {
    g_last_error = msg;
    return code;
}
using rt::GraphObj;

int check_open(rttnw_scene* s) {
    if (!s) return fail(RTTNW_ERR_INVALID, "scene is NULL");
    if (s->committed) return fail(RTTNW_ERR_STATE, "scene is committed and immutable");
    return RTTNW_OK;
}
rttnw_id push(rttnw_scene* s, GraphObj&& o) {
    s->graph.objs.push_back(std::move(o));
    return rttnw_id(s->graph.objs.size() - 1);
}
} // namespace

namespace rt {
void set_last_error(const std::string& msg) { g_last_error = msg; }
} // namespace rt

extern "C" {

int rttnw_abi_version(void) { return RTTNW_ABI_VERSION; }
const char* rttnw_last_error(void) { return g_last_error.c_str(); }

int rttnw_scene_create(uint64_t scene_seed, rttnw_scene** out) {
    if (!out) return fail(RTTNW_ERR_INVALID, "out is NULL");
    rttnw_scene* s = new (std::nothrow) rttnw_scene();
    if (!s) return fail(RTTNW_ERR_NOMEM, "out of memory");
    s->graph.seed = scene_seed;
    *out = s;
    return RTTNW_OK;
}

void rttnw_scene_destroy(rttnw_scene* s) {
    if (!s) return;
    if (s->device) rt::device_release(s->device);
    for (rt::DeviceState* d : s->more_devices) rt::device_release(d);
    delete s;
}

// ---- textures
rttnw_id rttnw_tex_solid(rttnw_scene* s, double r, double g, double b) {
    if (int rc = check_open(s)) return rc;
    GraphObj o; o.kind = GraphObj::TEX_SOLID_K; o.v[0] = r; o.v[1] = g; o.v[2] = b;
    return push(s, std::move(o));
}
rttnw_id rttnw_tex_checker(rttnw_scene* s, rttnw_id odd, rttnw_id even) {
    if (int rc = check_open(s)) return rc;
    if (!s->graph.is_texture(odd) || !s->graph.is_texture(even)) return fail(RTTNW_ERR_INVALID, "tex_checker: bad texture id");
    GraphObj o; o.kind = GraphObj::TEX_CHECKER_K; o.a = odd; o.b = even;
    return push(s, std::move(o));
}
rttnw_id rttnw_tex_noise(rttnw_scene* s, double scale) {
    if (int rc = check_open(s)) return rc;
    GraphObj o; o.kind = GraphObj::TEX_NOISE_K; o.v[0] = scale; o.a = int32_t(s->graph.n_noise++);
    return push(s, std::move(o));
}
rttnw_id rttnw_tex_image_rgba8(rttnw_scene* s, const uint8_t* rgba, uint32_t w, uint32_t h) {
    if (int rc = check_open(s)) return rc;
    GraphObj o; o.kind = GraphObj::TEX_IMAGE_K; o.a = -1;
    if (rgba && w && h) {
        s->graph.image_data.emplace_back(rgba, rgba + size_t(w) * h * 4);
        s->graph.image_w.push_back(w);
        s->graph.image_h.push_back(h);
        o.a = int32_t(s->graph.image_data.size() - 1);
    }
    return push(s, std::move(o));
}

// ---- materials
static rttnw_id push_material(rttnw_scene* s, int type, rttnw_id tex, double r, double g, double b, double param) {
    GraphObj o; o.kind = GraphObj::MAT_K; o.c = type; o.a = tex;
    o.v[0] = r; o.v[1] = g; o.v[2] = b; o.v[3] = param;
    return push(s, std::move(o));
}
rttnw_id rttnw_mat_lambertian(rttnw_scene* s, rttnw_id tex) {
    if (int rc = check_open(s)) return rc;
    if (!s->graph.is_texture(tex)) return fail(RTTNW_ERR_INVALID, "mat_lambertian: bad texture id");
    return push_material(s, rt::MAT_LAMBERTIAN, tex, 0, 0, 0, 0);
}
rttnw_id rttnw_mat_metal(rttnw_scene* s, double r, double g, double b, double fuzz) {
    if (int rc = check_open(s)) return rc;
    return push_material(s, rt::MAT_METAL, -1, r, g, b, std::fmin(fuzz, 1.0)); // material.rs:129
}
rttnw_id rttnw_mat_dielectric(rttnw_scene* s, double ri) {
    if (int rc = check_open(s)) return rc;
    return push_material(s, rt::MAT_DIELECTRIC, -1, 0, 0, 0, ri);
}
rttnw_id rttnw_mat_diffuse_light(rttnw_scene* s, rttnw_id tex) {
    if (int rc = check_open(s)) return rc;
    if (!s->graph.is_texture(tex)) return fail(RTTNW_ERR_INVALID, "mat_diffuse_light: bad texture id");
    return push_material(s, rt::MAT_DIFFUSE_LIGHT, tex, 0, 0, 0, 0);
}
rttnw_id rttnw_mat_isotropic(rttnw_scene* s, rttnw_id tex) {
    if (int rc = check_open(s)) return rc;
    if (!s->graph.is_texture(tex)) return fail(RTTNW_ERR_INVALID, "mat_isotropic: bad texture id");
    return push_material(s, rt::MAT_ISOTROPIC, tex, 0, 0, 0, 0);
}

// ---- hittables
rttnw_id rttnw_sphere(rttnw_scene* s, const double c[3], double radius, rttnw_id mat) {
    if (int rc = check_open(s)) return rc;
    if (!c || !s->graph.is_material(mat)) return fail(RTTNW_ERR_INVALID, "sphere: bad material id");
    GraphObj o; o.kind = GraphObj::SPHERE_K; o.v[0] = c[0]; o.v[1] = c[1]; o.v[2] = c[2]; o.v[3] = radius; o.a = mat;
    return push(s, std::move(o));
}
rttnw_id rttnw_moving_sphere(rttnw_scene* s, const double c0[3], const double c1[3], double t0, double t1, double radius,
                             rttnw_id mat) {
    if (int rc = check_open(s)) return rc;
    if (!c0 || !c1 || !s->graph.is_material(mat)) return fail(RTTNW_ERR_INVALID, "moving_sphere: bad material id");
    GraphObj o; o.kind = GraphObj::MOVING_K;
    for (int k = 0; k < 3; ++k) { o.v[k] = c0[k]; o.v[3 + k] = c1[k]; }
    o.v[6] = t0; o.v[7] = t1; o.v[8] = radius; o.a = mat;
    return push(s, std::move(o));
}
rttnw_id rttnw_rectangle(rttnw_scene* s, int plane, double a0, double a1, double b0, double b1, double k, rttnw_id mat) {
    if (int rc = check_open(s)) return rc;
    if (plane < 0 || plane > 2 || !s->graph.is_material(mat)) return fail(RTTNW_ERR_INVALID, "rectangle: bad plane or material id");
    GraphObj o; o.kind = GraphObj::RECT_K; o.v[0] = a0; o.v[1] = a1; o.v[2] = b0; o.v[3] = b1; o.v[4] = k; o.c = plane; o.a = mat;
    return push(s, std::move(o));
}
rttnw_id rttnw_cube(rttnw_scene* s, const double mn[3], const double mx[3], rttnw_id mat) {
    if (int rc = check_open(s)) return rc;
    if (!mn || !mx || !s->graph.is_material(mat)) return fail(RTTNW_ERR_INVALID, "cube: bad material id");
    GraphObj o; o.kind = GraphObj::CUBE_K;
    for (int k = 0; k < 3; ++k) { o.v[k] = mn[k]; o.v[3 + k] = mx[k]; }
    o.a = mat;
    return push(s, std::move(o));
}
rttnw_id rttnw_list(rttnw_scene* s) {
    if (int rc = check_open(s)) return rc;
    GraphObj o; o.kind = GraphObj::LIST_K;
    return push(s, std::move(o));
}
int rttnw_list_push(rttnw_scene* s, rttnw_id list, rttnw_id item) {
    if (int rc = check_open(s)) return rc;
    auto& g = s->graph;
    if (!g.is_hittable(list) || g.objs[list].kind != GraphObj::LIST_K) return fail(RTTNW_ERR_INVALID, "list_push: bad list id");
    if (!g.is_hittable(item) || item == list) return fail(RTTNW_ERR_INVALID, "list_push: bad item id");
    g.objs[list].items.push_back(item);
    return RTTNW_OK;
}
rttnw_id rttnw_bvh_tree(rttnw_scene* s, rttnw_id list) {
    if (int rc = check_open(s)) return rc;
    auto& g = s->graph;
    if (!g.is_hittable(list) || g.objs[list].kind != GraphObj::LIST_K) return fail(RTTNW_ERR_INVALID, "bvh_tree: bad list id");
    if (g.objs[list].items.empty()) return fail(RTTNW_ERR_INVALID, "bvh_tree: empty list");
    GraphObj o; o.kind = GraphObj::BVH_K; o.items = g.objs[list].items;
    g.objs[list].consumed = true; // BvhTree::from(list) takes the list by value — hittable.rs:254-258
    return push(s, std::move(o));
}
rttnw_id rttnw_translate(rttnw_scene* s, rttnw_id item, const double off[3]) {
    if (int rc = check_open(s)) return rc;
    if (!off || !s->graph.is_hittable(item)) return fail(RTTNW_ERR_INVALID, "translate: bad item id");
    GraphObj o; o.kind = GraphObj::TRANSLATE_K; o.a = item; o.v[0] = off[0]; o.v[1] = off[1]; o.v[2] = off[2];
    return push(s, std::move(o));
}
rttnw_id rttnw_rotate_y(rttnw_scene* s, rttnw_id item, double deg) {
    if (int rc = check_open(s)) return rc;
    if (!s->graph.is_hittable(item)) return fail(RTTNW_ERR_INVALID, "rotate_y: bad item id");
    GraphObj o; o.kind = GraphObj::ROTATE_K; o.a = item; o.v[0] = deg;
    return push(s, std::move(o));
}
rttnw_id rttnw_constant_medium(rttnw_scene* s, rttnw_id boundary, double density, rttnw_id tex) {
    if (int rc = check_open(s)) return rc;
    if (!s->graph.is_hittable(boundary) || !s->graph.is_texture(tex)) return fail(RTTNW_ERR_INVALID, "constant_medium: bad boundary or texture id");
    // phase_function: Isotropic { albedo } — hittable.rs:733
    rttnw_id iso = push_material(s, rt::MAT_ISOTROPIC, tex, 0, 0, 0, 0);
    GraphObj o; o.kind = GraphObj::MEDIUM_K; o.a = boundary; o.b = iso; o.v[0] = density; o.c = int32_t(s->n_media++);
    return push(s, std::move(o));
}
// ---- Hittable::bounding_box (hittable.rs:50) of any hittable of the graph, before or after commit.  The trait's second method: nothing on the
// render path asks for it (the lowering computes its own f32-outward boxes, scene_lower.cpp), so this is the boundary's introspection of the graph.
namespace {
struct Bound6 { double mn[3], mx[3]; };
void surround(Bound6& a, const Bound6& b) { // Bound::surrounding — bound.rs:34-46
    for (int k = 0; k < 3; ++k) { a.mn[k] = std::min(a.mn[k], b.mn[k]); a.mx[k] = std::max(a.mx[k], b.mx[k]); }
}
// true = Some(bound).  `depth`: a list that (transitively) holds itself cannot be built through this API (ids only refer backwards), the guard is belt and braces
bool graph_bounds(const rt::SceneGraph& g, int32_t id, double t0, double t1, Bound6& out, int depth) {
    if (depth > 64) return false;
    const GraphObj& o = g.objs[size_t(id)];
    switch (o.kind) {
    case GraphObj::SPHERE_K: // hittable.rs:125-130: centre -+ radius, literally (a negative radius gives min > max there too)
        for (int k = 0; k < 3; ++k) { out.mn[k] = o.v[k] - o.v[3]; out.mx[k] = o.v[k] + o.v[3]; }
        return true;
    case GraphObj::MOVING_K: { // hittable.rs:233-244 with center(time) of :187-191
        Bound6 b[2];
        const double tt[2] = {t0, t1};
        for (int e = 0; e < 2; ++e) {
            const double f = (tt[e] - o.v[6]) / (o.v[7] - o.v[6]);
            for (int k = 0; k < 3; ++k) {
                const double c = o.v[k] + f * (o.v[3 + k] - o.v[k]);
                b[e].mn[k] = c - o.v[8]; b[e].mx[k] = c + o.v[8];
            }
        }
        out = b[0];
        surround(out, b[1]);
        return true;
    }
    case GraphObj::RECT_K: { // hittable.rs:532-546: the two ranges as given, k -+ 0.0001
        const int a0 = o.c == 2 ? 1 : 0, a1 = o.c == 0 ? 1 : 2, ka = o.c == 0 ? 2 : (o.c == 1 ? 1 : 0);
        out.mn[a0] = o.v[0]; out.mx[a0] = o.v[1];
        out.mn[a1] = o.v[2]; out.mx[a1] = o.v[3];
        out.mn[ka] = o.v[4] - 0.0001; out.mx[ka] = o.v[4] + 0.0001;
        return true;
    }
    case GraphObj::CUBE_K: // hittable.rs:585-591: the two corners as given
        for (int k = 0; k < 3; ++k) { out.mn[k] = o.v[k]; out.mx[k] = o.v[3 + k]; }
        return true;
    case GraphObj::LIST_K: { // hittable.rs:165-176: None for an empty list or when any member has none
        if (o.items.empty()) return false;
        if (!graph_bounds(g, o.items[0], t0, t1, out, depth + 1)) return false;
        for (size_t i = 1; i < o.items.size(); ++i) {
            Bound6 b;
            if (!graph_bounds(g, o.items[i], t0, t1, b, depth + 1)) return false;
            surround(out, b);
        }
        return true;
    }
    case GraphObj::BVH_K: { // hittable.rs:370-372: the bound stored at construction — BvhTree::from = from_time(list, 0., 1.) (:255-257), whatever is asked;
        // the surrounding of the members' boxes, a member without one counted as Bound::default() like :306-317
        bool first = true;
        for (int32_t it : o.items) {
            Bound6 b;
            if (!graph_bounds(g, it, 0.0, 1.0, b, depth + 1)) b = Bound6{{0, 0, 0}, {0, 0, 0}};
            if (first) { out = b; first = false; } else surround(out, b);
        }
        return !first;
    }
    case GraphObj::TRANSLATE_K: { // hittable.rs:619-628
        if (!graph_bounds(g, o.a, t0, t1, out, depth + 1)) return false;
        for (int k = 0; k < 3; ++k) { out.mn[k] += o.v[k]; out.mx[k] += o.v[k]; }
        return true;
    }
    case GraphObj::ROTATE_K: { // hittable.rs:645-676,719-721: the item's box over (0., 1.), its eight corners turned about y, stored at construction and
        // returned whatever is asked.  The CORRECT rotation: the reference's z line reads the x it has just overwritten (:661-662, SURVEY quirk Q2 —
        // latent there, no YRotate is ever put into a BvhTree); a box that does not contain its object is not something to reproduce
        Bound6 b;
        if (!graph_bounds(g, o.a, 0.0, 1.0, b, depth + 1)) b = Bound6{{0, 0, 0}, {0, 0, 0}}; // (has_bound = false: Default::default(), :647-649)
        const double rad = o.v[0] * (3.14159265358979323846 / 180.0), sn = std::sin(rad), cs = std::cos(rad);
        for (int k = 0; k < 3; ++k) { out.mn[k] = INFINITY; out.mx[k] = -INFINITY; }
        for (int c = 0; c < 8; ++c) {
            const double x = (c & 1) ? b.mx[0] : b.mn[0], y = (c & 2) ? b.mx[1] : b.mn[1], z = (c & 4) ? b.mx[2] : b.mn[2];
            const double p[3] = {cs * x + sn * z, y, -sn * x + cs * z};
            for (int k = 0; k < 3; ++k) { out.mn[k] = std::min(out.mn[k], p[k]); out.mx[k] = std::max(out.mx[k], p[k]); }
        }
        return true;
    }
    case GraphObj::MEDIUM_K: // hittable.rs:798-800: the boundary's
        return graph_bounds(g, o.a, t0, t1, out, depth + 1);
    default:
        return false;
    }
}
} // namespace
int rttnw_hittable_bounds(const rttnw_scene* s, rttnw_id hittable, double initial_time, double final_time, double out_min_max[6]) {
    if (!s || !out_min_max) return fail(RTTNW_ERR_INVALID, "hittable_bounds: NULL argument");
    if (!s->graph.is_hittable(hittable)) return fail(RTTNW_ERR_INVALID, "hittable_bounds: bad hittable id");
    Bound6 b;
    if (!graph_bounds(s->graph, hittable, initial_time, final_time, b, 0)) return 0; // None
    for (int k = 0; k < 3; ++k) { out_min_max[k] = b.mn[k]; out_min_max[3 + k] = b.mx[k]; }
    return 1;
}
int rttnw_scene_set_world(rttnw_scene* s, rttnw_id world) {
    if (int rc = check_open(s)) return rc;
    auto& g = s->graph;
    if (!g.is_hittable(world) || g.objs[world].kind != GraphObj::LIST_K) return fail(RTTNW_ERR_INVALID, "set_world: bad list id");
    g.world = world;
    return RTTNW_OK;
}
int rttnw_scene_commit(rttnw_scene* s) {
    if (!s) return fail(RTTNW_ERR_INVALID, "scene is NULL");
    if (s->committed) return RTTNW_OK; // idempotent
    std::string err;
    rt::DeviceBvhApi device_builder;
    bool on_device = s->bvh_builder != RTTNW_BVH_HOST_SAH;
    if (on_device)
        if (int brc = rt::device_bvh_builder(s, device_builder, err)) {
            // (RTTNW_BVH_AUTO without a device builder — the host test build of this file — builds on the host; in the library itself a missing
            // device fails the commit a few lines down: there is no CPU render path)
            if (s->bvh_builder != RTTNW_BVH_AUTO) return fail(brc, err.c_str());
            on_device = false;
        }
    const auto t0 = std::chrono::steady_clock::now();
    s->build_kernel_ms = 0;
    int rc = rt::lower_scene(s->graph, s->flat, err, on_device ? &device_builder : nullptr);
    s->lower_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc) return fail(rc, err.c_str());
    rc = rt::device_commit(s, err);
    if (rc) return fail(rc, err.c_str());
    s->committed = true;
    return RTTNW_OK;
}

int rttnw_scene_set_bvh_builder(rttnw_scene* s, uint32_t builder) {
    if (int rc = check_open(s)) return rc;
    if (builder != RTTNW_BVH_HOST_SAH && builder != RTTNW_BVH_DEVICE_LBVH && builder != RTTNW_BVH_DEVICE_SAH && builder != RTTNW_BVH_AUTO) return fail(RTTNW_ERR_INVALID, "unknown BVH builder");
    s->bvh_builder = builder;
    return RTTNW_OK;
}

int rttnw_scene_build_info(const rttnw_scene* s, rttnw_build_info* out) {
    if (!s || !out) return fail(RTTNW_ERR_INVALID, "scene_build_info: NULL argument");
    if (!s->committed) return fail(RTTNW_ERR_STATE, "scene_build_info: scene is not committed");
    out->builder = s->bvh_builder;
    out->n_nodes = s->flat.total_nodes4();
    out->n_prims = s->flat.n_prims_in_bvh;
    out->stack_depth = s->flat.stack_depth;
    out->lower_ms = s->lower_ms;
    out->device_ms = s->build_kernel_ms;
    return RTTNW_OK;
}

int rttnw_debug_scene_nodes(const rttnw_scene* s, void* out_nodes, uint32_t max_nodes, int32_t* top_root) {
    if (!s || !s->committed) return fail(RTTNW_ERR_STATE, "debug_scene_nodes: scene is not committed");
    {   // trees the device builder left on the device are fetched when someone looks (once)
        std::string err;
        if (int rc = rt::materialize_host_nodes(const_cast<rttnw_scene*>(s)->flat, err)) return fail(rc, err.c_str());
    }
    const uint32_t n = uint32_t(std::min<size_t>(s->flat.nodes.size(), max_nodes));
    if (out_nodes && n) std::memcpy(out_nodes, s->flat.nodes.data(), size_t(n) * sizeof(rt::BvhNode));
    if (top_root) *top_root = s->flat.top_root2;
    return int(s->flat.nodes.size());
}

int rttnw_debug_scene_nodes4(const rttnw_scene* s, void* out_nodes, uint32_t max_nodes, int32_t* top_root) {
    if (!s || !s->committed) return fail(RTTNW_ERR_STATE, "debug_scene_nodes4: scene is not committed");
    {
        std::string err;
        if (int rc = rt::materialize_host_nodes(const_cast<rttnw_scene*>(s)->flat, err)) return fail(rc, err.c_str());
    }
    const uint32_t n = uint32_t(std::min<size_t>(s->flat.nodes4.size(), max_nodes));
    if (out_nodes && n) std::memcpy(out_nodes, s->flat.nodes4.data(), size_t(n) * sizeof(rt::Bvh4Node));
    if (top_root) *top_root = s->flat.top_root;
    return int(s->flat.nodes4.size());
}

const rttnw_builder_api* rttnw_builder(void) {
    static const rttnw_builder_api api = {
        rttnw_scene_create, rttnw_scene_destroy, rttnw_tex_solid, rttnw_tex_checker, rttnw_tex_noise,
        rttnw_tex_image_rgba8, rttnw_mat_lambertian, rttnw_mat_metal, rttnw_mat_dielectric,
        rttnw_mat_diffuse_light, rttnw_mat_isotropic, rttnw_sphere, rttnw_moving_sphere, rttnw_rectangle,
        rttnw_cube, rttnw_list, rttnw_list_push, rttnw_bvh_tree, rttnw_translate, rttnw_rotate_y,
        rttnw_constant_medium, rttnw_scene_set_world, rttnw_scene_commit, rttnw_last_error};
    return &api;
}

} // extern "C"
