// bvh_build.hip — linear BVH on the device: Morton keys -> radix sort -> Karras hierarchy -> bottom-up fit.
//
//   1. morton_kernel   63-bit Morton key of every leaf's centroid (21 bits per axis over the centroid bounds)
//   2. rocprim::radix_sort_pairs (key, leaf index)
//   3. hierarchy_kernel  one thread per inner node i in [0, n-2]: the range of sorted leaves it covers and its
//                      split, from common-prefix lengths (Karras, "Maximizing Parallelism in the Construction of
//                      BVHs, Octrees, and k-d Trees", HPG 2012); equal keys fall back to the index bits, so the
//                      hierarchy is well defined with duplicate centroids
//   4. fit_sweep_kernel  bottom-up fit in sweeps (one launch each, ~tree height of them): a node whose children are
//                      finished fills in BOTH children's boxes + child slots of the 64-byte node record
//                      (rt_types.hpp BvhNode) and its own box / height for its parent
//   5. preorder_kernel + relayout_kernel  nodes re-numbered depth-first (root = 0, a node next to its first child)
//   6. collapse to the 4-WIDE records the trace kernels walk (rt_types.hpp Bvh4Node), on the device, with the rule of the
//      host's collapse4 (scene_lower.cpp): a record takes a binary node's two children and, while it has fewer than four,
//      replaces its inner child of largest area by that child's two children.  Which binary nodes become records ("heads")
//      is decided top-down, one launch per 4-wide level over a frontier (collapse_mark_kernel); an exclusive scan of the head
//      flags in binary pre-order numbers the records (so records keep the binary tree's pre-order: a record next to its
//      first child, the top levels together); collapse_write_kernel writes them; collapse_need_kernel, bottom-up over the
//      same frontiers, gives the exact traversal-stack bound.  The tree STAYS on the device: the scene's node array adopts
//      the buffer (render_common.hpp DeviceScene), nothing is copied back unless someone asks to inspect it.
// HBM-bound integer/byte work: coalesced SoA arrays, no LDS needed.  The hierarchy has exactly n-1 inner nodes;
// the root is node 0 before and after the re-numbering.
#include "bvh_build.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <string.h> // rocprim/iterator/texture_cache_iterator.hpp calls memset unqualified
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>

namespace rt {
namespace {

#define LBVH_TRY(expr)                                                                           \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) { err = std::string("lbvh_build: ") + hipGetErrorString(e_); rc = -4; goto done; } \
    } while (0)

__device__ __forceinline__ uint64_t spread21(uint32_t v) { // bit i of v -> bit 3i
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void morton_kernel(const BuildPrim* __restrict__ prims, uint32_t n, float cx, float cy, float cz, float sx, float sy,
                              float sz, uint64_t* __restrict__ keys, uint32_t* __restrict__ order) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BuildPrim p = prims[i];
    const float qx = (0.5f * (p.lo[0] + p.hi[0]) - cx) * sx, qy = (0.5f * (p.lo[1] + p.hi[1]) - cy) * sy,
                qz = (0.5f * (p.lo[2] + p.hi[2]) - cz) * sz;
    const float top = 2097151.f; // 2^21 - 1
    const uint32_t ux = uint32_t(fminf(fmaxf(qx, 0.f), top)), uy = uint32_t(fminf(fmaxf(qy, 0.f), top)),
                   uz = uint32_t(fminf(fmaxf(qz, 0.f), top));
    keys[i] = (spread21(ux) << 2) | (spread21(uy) << 1) | spread21(uz);
    order[i] = i;
}

// Length of the common prefix of sorted leaves i and j (-1 outside the array); equal keys continue with the index.
__device__ __forceinline__ int common_prefix(const uint64_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(int(uint32_t(i) ^ uint32_t(j)));
}

// child slot of the hierarchy before the fit: >= 0 inner node, < 0: ~(sorted leaf position)
__global__ void hierarchy_kernel(const uint64_t* __restrict__ keys, int n, int2* __restrict__ children, int* __restrict__ node_parent) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = common_prefix(keys, n, i, i + 1) - common_prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = common_prefix(keys, n, i, i - d);
    int lmax = 2;
    while (common_prefix(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (common_prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = common_prefix(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (common_prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    int2 c;
    if (lo == gamma) c.x = ~gamma; else { c.x = gamma; node_parent[gamma] = i; }
    if (hi == gamma + 1) c.y = ~(gamma + 1); else { c.y = gamma + 1; node_parent[gamma + 1] = i; }
    children[i] = c;
    if (i == 0) node_parent[0] = -1;
}

struct NodeBox { float lo[3]; float hi[3]; };

// One sweep of the bottom-up fit: every inner node whose two children were finished by an EARLIER sweep (a leaf always
// is) fills in its 64-byte record — both children's boxes and child slots — and its own box / height for its parent.
// `done[i]` holds the number of the sweep that finished node i (0 = not yet).  Sweeps are separate launches, so a
// node only ever reads what a previous launch wrote: no fences, no atomics, nothing to keep coherent between the
// XCDs' L2s.  (The textbook form — one thread per leaf climbing with an atomic arrival counter — needs an
// agent-scope fence per level and measured 7.3 ms for 10^6 leaves on MI355X; ~height sweeps of this take 0.5 ms.)
__global__ void fit_sweep_kernel(const BuildPrim* __restrict__ prims, const uint32_t* __restrict__ order, int n_inner,
                                 const int2* __restrict__ children, int* __restrict__ done, int sweep, NodeBox* __restrict__ node_box,
                                 int* __restrict__ node_levels, BvhNode* __restrict__ out, int base) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner || done[i] != 0) return;
    const int2 c = children[i];
    if (c.x >= 0) { const int d = done[c.x]; if (d == 0 || d >= sweep) return; }
    if (c.y >= 0) { const int d = done[c.y]; if (d == 0 || d >= sweep) return; }
    BvhNode nd;
    float lo[3], hi[3];
    int levels = 0;
    for (int side = 0; side < 2; ++side) {
        const int ch = side == 0 ? c.x : c.y;
        float* blo = side == 0 ? nd.lo0 : nd.lo1;
        float* bhi = side == 0 ? nd.hi0 : nd.hi1;
        int code;
        if (ch < 0) {
            const BuildPrim q = prims[order[~ch]];
            for (int a = 0; a < 3; ++a) { blo[a] = q.lo[a]; bhi[a] = q.hi[a]; }
            code = q.leaf;
        } else {
            const NodeBox nb = node_box[ch];
            for (int a = 0; a < 3; ++a) { blo[a] = nb.lo[a]; bhi[a] = nb.hi[a]; }
            code = base + ch;
            levels = max(levels, node_levels[ch]);
        }
        if (side == 0) nd.child0 = code; else nd.child1 = code;
        for (int a = 0; a < 3; ++a) {
            lo[a] = side == 0 ? blo[a] : fminf(lo[a], blo[a]);
            hi[a] = side == 0 ? bhi[a] : fmaxf(hi[a], bhi[a]);
        }
    }
    nd.pad0 = nd.pad1 = 0;
    out[i] = nd;
    NodeBox own;
    for (int a = 0; a < 3; ++a) { own.lo[a] = lo[a]; own.hi[a] = hi[a]; }
    node_box[i] = own;
    node_levels[i] = levels + 1;
    done[i] = sweep;
}

// Depth-first (pre-order) position of every inner node: the first leaf of its range plus the number of ancestors that
// hold it in their LEFT subtree.  (Pre-order puts a node after its ancestors and after every node lying entirely to
// its left; inner nodes and the gaps between adjacent sorted leaves correspond one to one, which gives the count.)
// The hierarchy's own numbering scatters the top of the tree over the whole array; pre-order keeps a node next to
// its first child and the top levels together, as the host builder does.
__global__ void preorder_kernel(const int2* __restrict__ children, const int* __restrict__ node_parent, int n_inner, int* __restrict__ pos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner) return;
    int first = i; // leftmost leaf of node i's range: follow left children down
    for (int c = children[i].x; c >= 0; c = children[c].x) first = c;
    first = ~children[first].x;
    int lefts = 0;
    for (int c = i, p = node_parent[i]; p >= 0; c = p, p = node_parent[p]) lefts += children[p].x == c;
    pos[i] = first + lefts;
}
__global__ void relayout_kernel(const BvhNode* __restrict__ in, const int* __restrict__ pos, int n_inner, int base, BvhNode* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner) return;
    BvhNode nd = in[i];
    if (nd.child0 >= 0) nd.child0 = base + pos[nd.child0 - base];
    if (nd.child1 >= 0) nd.child1 = base + pos[nd.child1 - base];
    out[pos[i]] = nd;
}


// ---- binary tree -> 4-wide records
struct Slot { float lo[3], hi[3]; int32_t child; };
__device__ __forceinline__ double slot_area(const Slot& s) { // as scene_lower.cpp slot_area: the same doubles, the same pick
    const double d0 = double(s.hi[0]) - s.lo[0], d1 = double(s.hi[1]) - s.lo[1], d2 = double(s.hi[2]) - s.lo[2];
    if (d0 < 0 || d1 < 0 || d2 < 0) return 0.0;
    return 2.0 * (d0 * d1 + d1 * d2 + d2 * d0);
}
__device__ __forceinline__ void child_slots(const BvhNode& nd, Slot* out, int& n) {
    if (nd.child0 != CHILD_EMPTY) { Slot& s = out[n++]; for (int a = 0; a < 3; ++a) { s.lo[a] = nd.lo0[a]; s.hi[a] = nd.hi0[a]; } s.child = nd.child0; }
    if (nd.child1 != CHILD_EMPTY) { Slot& s = out[n++]; for (int a = 0; a < 3; ++a) { s.lo[a] = nd.lo1[a]; s.hi[a] = nd.hi1[a]; } s.child = nd.child1; }
}
// The (up to) four slots of the record headed by binary node b.
__device__ __forceinline__ int expand4(const BvhNode* __restrict__ nodes, int b, Slot* slots) {
    int n = 0;
    child_slots(nodes[b], slots, n);
    while (n < 4) {
        int pick = -1;
        double best = -1.0;
        for (int i = 0; i < n; ++i)
            if (slots[i].child >= 0) { const double a = slot_area(slots[i]); if (a > best) { best = a; pick = i; } }
        if (pick < 0) break;
        const BvhNode inner = nodes[slots[pick].child];
        for (int i = pick; i + 1 < n; ++i) slots[i] = slots[i + 1]; // keep the order of the others
        --n;
        child_slots(inner, slots, n);
    }
    return n;
}
// One 4-wide level: every head of the frontier marks the inner children of its record as heads of the next level.
__global__ void collapse_mark_kernel(const BvhNode* __restrict__ nodes, const int* __restrict__ frontier, int n_front, int* __restrict__ next,
                                     int* __restrict__ next_count, uint32_t* __restrict__ is_head) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_front) return;
    Slot slots[5];
    const int n = expand4(nodes, frontier[i], slots);
    for (int c = 0; c < n; ++c)
        if (slots[c].child >= 0) {
            is_head[slots[c].child] = 1u;
            next[atomicAdd(next_count, 1)] = slots[c].child;
        }
}
__global__ void collapse_write_kernel(const BvhNode* __restrict__ nodes, const int* __restrict__ heads, int n_heads, const uint32_t* __restrict__ rank,
                                      Bvh4Node* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_heads) return;
    const int b = heads[i];
    Slot slots[5];
    const int n = expand4(nodes, b, slots);
    Bvh4Node o;
    for (int c = 0; c < 4; ++c) {
        if (c < n) {
            for (int a = 0; a < 3; ++a) { o.lo[a][c] = slots[c].lo[a]; o.hi[a][c] = slots[c].hi[a]; }
            o.child[c] = slots[c].child >= 0 ? int32_t(rank[slots[c].child]) : slots[c].child;
        } else {
            for (int a = 0; a < 3; ++a) { o.lo[a][c] = INFINITY; o.hi[a][c] = -INFINITY; }
            o.child[c] = CHILD_EMPTY;
        }
        o.pad[c] = 0;
    }
    out[rank[b]] = o;
}
// need(record) = (children - 1) + the largest need among its inner children: one level per launch, deepest first.
__global__ void collapse_need_kernel(const Bvh4Node* __restrict__ out, const int* __restrict__ heads, int n_level, const uint32_t* __restrict__ rank,
                                     uint32_t* __restrict__ need) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_level) return;
    const uint32_t r = rank[heads[i]];
    const Bvh4Node nd = out[r];
    uint32_t n = 0, deepest = 0;
    for (int c = 0; c < 4; ++c) {
        if (nd.child[c] == CHILD_EMPTY) continue;
        ++n;
        if (nd.child[c] >= 0) deepest = max(deepest, need[nd.child[c]]);
    }
    need[r] = (n > 0 ? n - 1 : 0) + deepest;
}
__global__ void rebase4_kernel(Bvh4Node* __restrict__ nodes, uint32_t n, int32_t base) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 4u) return;
    int32_t* c = &nodes[i >> 2].child[i & 3u];
    if (*c >= 0) *c += base;
}

} // namespace

namespace {
struct DeviceFree { // frees on the device the buffer lives on
    int device;
    void operator()(void* p) const {
        if (!p) return;
        int prev = -1;
        (void)hipGetDevice(&prev);
        if (prev != device) (void)hipSetDevice(device);
        (void)hipFree(p);
        if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    }
};
} // namespace

int lbvh_build_device_tree(const std::vector<BuildPrim>& prims, DeviceTree& tree, double* kernel_ms, std::string& err) {
    const size_t n = prims.size();
    if (n < 2 || n >= (size_t(1) << 26)) { err = "lbvh_build: needs 2 .. 2^26-1 leaves"; return -1; }
    int rc = 0;
    const auto wall0 = std::chrono::steady_clock::now();
    auto wall_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count(); };
    double t_alloc = 0, t_up = 0, t_kernels = 0;
    // centroid bounds on the host (the leaves come from the host anyway)
    float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const BuildPrim& p : prims)
        for (int a = 0; a < 3; ++a) {
            const float c = 0.5f * (p.lo[a] + p.hi[a]);
            cmin[a] = std::min(cmin[a], c);
            cmax[a] = std::max(cmax[a], c);
        }
    float scale[3];
    for (int a = 0; a < 3; ++a) {
        const float ext = cmax[a] - cmin[a];
        scale[a] = (ext > 0.f && std::isfinite(ext)) ? 2097151.f / ext : 0.f;
    }

    int device = -1;
    BuildPrim* d_prims = nullptr;
    uint64_t *d_keys = nullptr, *d_keys2 = nullptr;
    uint32_t *d_order = nullptr, *d_order2 = nullptr;
    int2* d_children = nullptr;
    int *d_node_parent = nullptr, *d_done = nullptr, *d_levels = nullptr;
    NodeBox* d_box = nullptr;
    BvhNode *d_out = nullptr, *d_out2 = nullptr;
    int* d_pos = nullptr;
    void *d_temp = nullptr, *d_temp2 = nullptr;
    size_t temp_bytes = 0, temp2_bytes = 0;
    // collapse
    int *d_heads = nullptr, *d_count = nullptr;
    uint32_t *d_is_head = nullptr, *d_rank = nullptr, *d_need = nullptr;
    Bvh4Node* d_out4 = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const uint32_t nb = uint32_t((n + 255) / 256);
    const int n_inner = int(n - 1);
    int h_levels = 0;
    std::vector<int> level_off; // frontier k = heads[level_off[k] .. level_off[k + 1])
    uint32_t n_heads = 0, h_need = 0;
    {
        LBVH_TRY(hipGetDevice(&device));
        LBVH_TRY(hipMalloc((void**)&d_prims, n * sizeof(BuildPrim)));
        LBVH_TRY(hipMalloc((void**)&d_keys, n * 8));
        LBVH_TRY(hipMalloc((void**)&d_keys2, n * 8));
        LBVH_TRY(hipMalloc((void**)&d_order, n * 4));
        LBVH_TRY(hipMalloc((void**)&d_order2, n * 4));
        LBVH_TRY(hipMalloc((void**)&d_children, (n - 1) * sizeof(int2)));
        LBVH_TRY(hipMalloc((void**)&d_node_parent, (n - 1) * 4));
        LBVH_TRY(hipMalloc((void**)&d_done, (n - 1) * 4));
        LBVH_TRY(hipMalloc((void**)&d_levels, (n - 1) * 4));
        LBVH_TRY(hipMalloc((void**)&d_box, (n - 1) * sizeof(NodeBox)));
        LBVH_TRY(hipMalloc((void**)&d_out, (n - 1) * sizeof(BvhNode)));
        LBVH_TRY(hipMalloc((void**)&d_out2, (n - 1) * sizeof(BvhNode)));
        LBVH_TRY(hipMalloc((void**)&d_pos, (n - 1) * 4));
        LBVH_TRY(hipMalloc((void**)&d_heads, (n - 1) * 4));
        LBVH_TRY(hipMalloc((void**)&d_count, 4));
        LBVH_TRY(hipMalloc((void**)&d_is_head, (n - 1) * 4));
        LBVH_TRY(hipMalloc((void**)&d_rank, (n - 1) * 4));
        LBVH_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys, d_keys2, d_order, d_order2, n, 0, 63, hipStream_t(0)));
        LBVH_TRY(hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 16)));
        LBVH_TRY(rocprim::exclusive_scan(nullptr, temp2_bytes, d_is_head, d_rank, 0u, size_t(n_inner), rocprim::plus<uint32_t>(), hipStream_t(0)));
        LBVH_TRY(hipMalloc(&d_temp2, std::max<size_t>(temp2_bytes, 16)));
        LBVH_TRY(hipEventCreate(&e0));
        LBVH_TRY(hipEventCreate(&e1));
        t_alloc = wall_ms();
        LBVH_TRY(hipMemcpy(d_prims, prims.data(), n * sizeof(BuildPrim), hipMemcpyHostToDevice));
        t_up = wall_ms();

        LBVH_TRY(hipEventRecord(e0, 0));
        LBVH_TRY(hipMemsetAsync(d_done, 0, (n - 1) * 4, 0));
        hipLaunchKernelGGL(morton_kernel, dim3(nb), dim3(256), 0, 0, d_prims, uint32_t(n), cmin[0], cmin[1], cmin[2], scale[0], scale[1],
                           scale[2], d_keys, d_order);
        LBVH_TRY(rocprim::radix_sort_pairs(d_temp, temp_bytes, d_keys, d_keys2, d_order, d_order2, n, 0, 63, hipStream_t(0)));
        hipLaunchKernelGGL(hierarchy_kernel, dim3(nb), dim3(256), 0, 0, d_keys2, int(n), d_children, d_node_parent);
        // bottom-up fit: sweeps until the root is finished (the tree's height, which is not known beforehand: look at
        // the root's flag after every batch of sweeps)
        for (int sweep = 1, root_done = 0; !root_done;) {
            for (int k = 0; k < 16; ++k, ++sweep)
                hipLaunchKernelGGL(fit_sweep_kernel, dim3(nb), dim3(256), 0, 0, d_prims, d_order2, n_inner, d_children, d_done, sweep,
                                   d_box, d_levels, d_out, 0);
            LBVH_TRY(hipMemcpy(&root_done, d_done, 4, hipMemcpyDeviceToHost));
            if (sweep > 4096) { err = "lbvh_build: fit did not converge"; rc = -4; goto done; }
        }
        hipLaunchKernelGGL(preorder_kernel, dim3(nb), dim3(256), 0, 0, d_children, d_node_parent, n_inner, d_pos);
        hipLaunchKernelGGL(relayout_kernel, dim3(nb), dim3(256), 0, 0, d_out, d_pos, n_inner, 0, d_out2);

        // ---- collapse: heads level by level (root first), numbering, records, stack bound
        LBVH_TRY(hipMemsetAsync(d_is_head, 0, size_t(n_inner) * 4, 0));
        {
            const uint32_t one = 1u;
            const int zero = 0;
            LBVH_TRY(hipMemcpyAsync(d_is_head, &one, 4, hipMemcpyHostToDevice, 0)); // the root (binary node 0) heads record 0
            LBVH_TRY(hipMemcpyAsync(d_heads, &zero, 4, hipMemcpyHostToDevice, 0));
        }
        level_off.push_back(0);
        level_off.push_back(1);
        for (;;) {
            const int lo = level_off[level_off.size() - 2], hi = level_off.back(), n_front = hi - lo;
            LBVH_TRY(hipMemsetAsync(d_count, 0, 4, 0));
            hipLaunchKernelGGL(collapse_mark_kernel, dim3((n_front + 255) / 256), dim3(256), 0, 0, d_out2, d_heads + lo, n_front, d_heads + hi, d_count,
                               d_is_head);
            int added = 0;
            LBVH_TRY(hipMemcpy(&added, d_count, 4, hipMemcpyDeviceToHost));
            if (added == 0) break;
            if (hi + added > n_inner || level_off.size() > 4096) { err = "lbvh_build: collapse did not converge"; rc = -4; goto done; }
            level_off.push_back(hi + added);
        }
        n_heads = uint32_t(level_off.back());
        LBVH_TRY(rocprim::exclusive_scan(d_temp2, temp2_bytes, d_is_head, d_rank, 0u, size_t(n_inner), rocprim::plus<uint32_t>(), hipStream_t(0)));
        LBVH_TRY(hipMalloc((void**)&d_out4, size_t(n_heads) * sizeof(Bvh4Node)));
        LBVH_TRY(hipMalloc((void**)&d_need, size_t(n_heads) * 4));
        hipLaunchKernelGGL(collapse_write_kernel, dim3((n_heads + 255) / 256), dim3(256), 0, 0, d_out2, d_heads, int(n_heads), d_rank, d_out4);
        for (size_t k = level_off.size() - 1; k-- > 0;) {
            const int lo = level_off[k], n_level = level_off[k + 1] - lo;
            hipLaunchKernelGGL(collapse_need_kernel, dim3((n_level + 255) / 256), dim3(256), 0, 0, d_out4, d_heads + lo, n_level, d_rank, d_need);
        }
        LBVH_TRY(hipEventRecord(e1, 0));
        LBVH_TRY(hipGetLastError());
        LBVH_TRY(hipEventSynchronize(e1));
        float ms = 0;
        LBVH_TRY(hipEventElapsedTime(&ms, e0, e1));
        if (kernel_ms) *kernel_ms += ms;
        t_kernels = wall_ms();
        LBVH_TRY(hipMemcpy(&h_levels, d_levels, 4, hipMemcpyDeviceToHost));
        LBVH_TRY(hipMemcpy(&h_need, d_need, 4, hipMemcpyDeviceToHost)); // record 0 = the root
        tree = DeviceTree();
        tree.nodes4 = std::shared_ptr<void>(d_out4, DeviceFree{device});
        tree.nodes2 = std::shared_ptr<void>(d_out2, DeviceFree{device});
        d_out4 = nullptr; d_out2 = nullptr; // owned by the tree now
        tree.count4 = n_heads;
        tree.count2 = uint32_t(n_inner);
        tree.need = h_need;
        tree.levels = uint32_t(h_levels);
        tree.device = device;
        if (getenv("RTTNW_DEBUG_LOWER"))
            fprintf(stderr, "[lbvh] %zu leaves -> %u 4-wide records in %zu levels: centroid bounds + allocations %.1f ms, upload %.1f ms, kernels %.1f ms (device %.2f); nothing downloaded\n",
                    n, n_heads, level_off.size() - 1, t_alloc, t_up - t_alloc, t_kernels - t_up, ms);
    }
done:
    for (void* p : {(void*)d_prims, (void*)d_keys, (void*)d_keys2, (void*)d_order, (void*)d_order2, (void*)d_children, (void*)d_node_parent,
                    (void*)d_done, (void*)d_levels, (void*)d_box, (void*)d_out, (void*)d_out2, (void*)d_pos, d_temp, d_temp2, (void*)d_heads,
                    (void*)d_count, (void*)d_is_head, (void*)d_rank, (void*)d_need, (void*)d_out4})
        if (p) (void)hipFree(p);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}

int device_tree_rebase(DeviceTree& tree, uint32_t base4, uint32_t base2, std::string& err) {
    int rc = 0;
    int prev = -1;
    (void)hipGetDevice(&prev);
    {
        LBVH_TRY(hipSetDevice(tree.device));
        const int32_t delta = int32_t(base4) - int32_t(tree.base4);
        if (delta != 0 && tree.count4) {
            hipLaunchKernelGGL(rebase4_kernel, dim3((tree.count4 * 4u + 255u) / 256u), dim3(256), 0, 0, (Bvh4Node*)tree.nodes4.get(), tree.count4, delta);
            LBVH_TRY(hipGetLastError());
            LBVH_TRY(hipDeviceSynchronize());
        }
        tree.base4 = base4;
        tree.base2 = base2;
    }
done:
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}

int device_tree_download(const DeviceTree& tree, Bvh4Node* out4, BvhNode* out2, std::string& err) {
    int rc = 0;
    int prev = -1;
    (void)hipGetDevice(&prev);
    {
        LBVH_TRY(hipSetDevice(tree.device));
        if (out4 && tree.count4) LBVH_TRY(hipMemcpy(out4, tree.nodes4.get(), size_t(tree.count4) * sizeof(Bvh4Node), hipMemcpyDeviceToHost));
        if (out2 && tree.count2) {
            LBVH_TRY(hipMemcpy(out2, tree.nodes2.get(), size_t(tree.count2) * sizeof(BvhNode), hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < tree.count2; ++i) { // the binary records keep local indices on the device
                if (out2[i].child0 >= 0) out2[i].child0 += int32_t(tree.base2);
                if (out2[i].child1 >= 0) out2[i].child1 += int32_t(tree.base2);
            }
        }
    }
done:
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}

} // namespace rt
