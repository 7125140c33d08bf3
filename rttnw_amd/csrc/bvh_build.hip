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

} // namespace

int lbvh_build_device(const std::vector<BuildPrim>& prims, std::vector<BvhNode>& nodes, int32_t& root, uint32_t& levels,
                      double* kernel_ms, std::string& err) {
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

    BuildPrim* d_prims = nullptr;
    uint64_t *d_keys = nullptr, *d_keys2 = nullptr;
    uint32_t *d_order = nullptr, *d_order2 = nullptr;
    int2* d_children = nullptr;
    int *d_node_parent = nullptr, *d_done = nullptr, *d_levels = nullptr;
    NodeBox* d_box = nullptr;
    BvhNode *d_out = nullptr, *d_out2 = nullptr;
    int* d_pos = nullptr;
    void* d_temp = nullptr;
    size_t temp_bytes = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const int base = int(nodes.size());
    const uint32_t nb = uint32_t((n + 255) / 256);
    int h_levels = 0;
    {
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
        LBVH_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys, d_keys2, d_order, d_order2, n, 0, 63, hipStream_t(0)));
        LBVH_TRY(hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 16)));
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
                hipLaunchKernelGGL(fit_sweep_kernel, dim3(nb), dim3(256), 0, 0, d_prims, d_order2, int(n - 1), d_children, d_done, sweep,
                                   d_box, d_levels, d_out, base);
            LBVH_TRY(hipMemcpy(&root_done, d_done, 4, hipMemcpyDeviceToHost));
            if (sweep > 4096) { err = "lbvh_build: fit did not converge"; rc = -4; goto done; }
        }
        hipLaunchKernelGGL(preorder_kernel, dim3(nb), dim3(256), 0, 0, d_children, d_node_parent, int(n - 1), d_pos);
        hipLaunchKernelGGL(relayout_kernel, dim3(nb), dim3(256), 0, 0, d_out, d_pos, int(n - 1), base, d_out2);
        LBVH_TRY(hipEventRecord(e1, 0));
        LBVH_TRY(hipGetLastError());
        LBVH_TRY(hipEventSynchronize(e1));
        float ms = 0;
        LBVH_TRY(hipEventElapsedTime(&ms, e0, e1));
        if (kernel_ms) *kernel_ms += ms;
        t_kernels = wall_ms();

        nodes.resize(size_t(base) + n - 1);
        LBVH_TRY(hipMemcpy(nodes.data() + base, d_out2, (n - 1) * sizeof(BvhNode), hipMemcpyDeviceToHost));
        LBVH_TRY(hipMemcpy(&h_levels, d_levels, 4, hipMemcpyDeviceToHost));
        root = base;
        levels = uint32_t(h_levels);
        if (getenv("RTTNW_DEBUG_LOWER"))
            fprintf(stderr, "[lbvh] %zu leaves: centroid bounds + allocations %.1f ms, upload %.1f ms, kernels %.1f ms (device %.2f), download %.1f ms\n", n, t_alloc,
                    t_up - t_alloc, t_kernels - t_up, ms, wall_ms() - t_kernels);
    }
done:
    if (rc) nodes.resize(size_t(base));
    for (void* p : {(void*)d_prims, (void*)d_keys, (void*)d_keys2, (void*)d_order, (void*)d_order2, (void*)d_children, (void*)d_node_parent,
                    (void*)d_done, (void*)d_levels, (void*)d_box, (void*)d_out, (void*)d_out2, (void*)d_pos, d_temp})
        if (p) (void)hipFree(p);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}

} // namespace rt
