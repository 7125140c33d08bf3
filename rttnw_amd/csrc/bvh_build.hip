// bvh_build.hip — BVH construction on the device.  Two hierarchies over the same leaves, one 4-wide collapse after either:
//   * RTTNW_BVH_DEVICE_LBVH: linear BVH — Morton keys -> radix sort -> Karras hierarchy -> bottom-up fit (steps 1-5 below);
//   * RTTNW_BVH_DEVICE_SAH: the host builder's binned surface-area heuristic as level-synchronous kernels (the "binned SAH on
//     the device" section further down): slower to build (10.6 against 1.4 ms for 10^6 leaves), renders like the host's tree.
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
//      is decided top-down, one launch per 4-wide level over a frontier (collapse_count_kernel + a scan + collapse_mark_kernel:
//      the frontier's order is deterministic); a record's number is its head's position in the frontier arrays — LEVEL order,
//      the inner children of a record side by side (round 6: the two 64-byte quantised records of a 128-byte line are
//      siblings; RTTNW_NODE_ORDER=pre: the binary tree's pre-order of rounds 1-5, by a scan of the head flags);
//      collapse_write_kernel writes them; collapse_need_kernel, bottom-up over the
//      same frontiers, gives the exact traversal-stack bound.  The tree STAYS on the device: the scene's node array adopts
//      the buffer (render_common.hpp DeviceScene), nothing is copied back unless someone asks to inspect it.
// HBM-bound integer/byte work: coalesced SoA arrays, no LDS needed.  The hierarchy has exactly n-1 inner nodes;
// the root is node 0 before and after the re-numbering.
#include "bvh_build.hpp"
#include "bvh_quant.hpp"

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


// ---- binned SAH on the device (RTTNW_BVH_DEVICE_SAH): the host builder's algorithm (scene_lower.cpp split(): best of the binned
// surface-area-heuristic planes of the three axes over the centroid bounds, every leaf one record) as level-synchronous kernels.
// A node of the tree IS its pre-order index: a node over `count` leaves has count - 1 inner nodes below and including it, so the
// children of node s whose left side holds nl leaves are s + 1 and s + nl — numbering needs no pass of its own, and nothing
// depends on the order in which the level's segments are processed.  The leaves of a node stand contiguously in a ping-pong pair of
// leaf arrays ([begin, begin + count)); a level
//   * bins the leaves of every LARGE segment (> SAH_SMALL leaves) into 3 x SAH_BINS bins — per-block LDS bins where a block's 256
//     positions lie in one segment (the top levels), global atomics otherwise; boxes and centroid bounds as order-preserving
//     integers, so the atomic min / max do not depend on arrival order —
//   * picks every large segment's plane (one thread per segment), writes its node record, creates its children,
//   * partitions the leaves stably (a scan of the "goes left" flags over all positions), and
//   * splits every SMALL segment by an exact sweep, one wave per segment: lane i tries "everything up to leaf i's centroid goes
//     left" on each axis, the wave takes the cheapest candidate.
constexpr int SAH_BINS = 32; // (as the host builder: 16 bins cost final_scene 2.8 % more node visits)
constexpr int SAH_SMALL = 64;
__device__ __forceinline__ uint32_t enc_f(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float dec_f(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e ^ 0x80000000u) : ~e); }
struct SahBin { uint32_t count; uint32_t lo[3], hi[3]; uint32_t clo[3], chi[3]; }; // boxes / centroid bounds: enc_f
struct SahSeg { int begin, count; float clo[3], chi[3]; };                           // a node's leaves and their centroid bounds
struct SahSplit { int axis, bin, nl, median; };                                      // how a large segment's leaves divide (median: by position)
__device__ __forceinline__ float box_area(const float* lo, const float* hi) {
    const float d0 = hi[0] - lo[0], d1 = hi[1] - lo[1], d2 = hi[2] - lo[2];
    if (d0 < 0.f || d1 < 0.f || d2 < 0.f) return 0.f;
    return d0 * d1 + d1 * d2 + d2 * d0;
}
__device__ __forceinline__ int sah_bin_of(float c, float clo, float chi) {
    const float ext = chi - clo;
    if (!(ext > 0.f)) return 0;
    return min(SAH_BINS - 1, max(0, int((c - clo) * (float(SAH_BINS) / ext))));
}
__global__ void sah_bins_init_kernel(SahBin* __restrict__ bins, int n_bins) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bins) return;
    SahBin b;
    b.count = 0;
    for (int a = 0; a < 3; ++a) { b.lo[a] = 0xFFFFFFFFu; b.hi[a] = 0u; b.clo[a] = 0xFFFFFFFFu; b.chi[a] = 0u; }
    bins[i] = b;
}
// seg[i]: the large segment (node) position i belongs to, or < 0 (a small segment's, or finished).  The positions of a
// segment are contiguous, so a block of 256 positions (the top levels) or a wave of 64 (segments of a few hundred leaves) mostly
// lies in ONE segment: its leaves are binned in LDS — the block's bins, or the wave's own — and only the bins they touched
// go out as global atomics (a tenth of the atomics of binning straight into global memory for segments of 65-1000 leaves: 1.8 ->
// 0.3 ms per level of 10^6 leaves).
constexpr int SAH_BIN_WORDS = 3 * SAH_BINS * 13;
__device__ __forceinline__ bool sah_word_is_min(int w) { return (w >= 1 && w <= 3) || (w >= 7 && w <= 9); } // lo / clo: atomicMin
__device__ __forceinline__ void sah_bins_clear(uint32_t* b, int lane, int lanes) {
    for (int k = lane; k < SAH_BIN_WORDS; k += lanes) { const int w = k % 13; b[k] = w == 0 ? 0u : (sah_word_is_min(w) ? 0xFFFFFFFFu : 0u); }
}
__device__ __forceinline__ void sah_bins_flush(const uint32_t* b, uint32_t* g, int lane, int lanes) {
    for (int k = lane; k < SAH_BIN_WORDS; k += lanes) {
        const int w = k % 13;
        const uint32_t v = b[k];
        if (b[k - w] == 0u) continue; // an empty bin
        if (w == 0) atomicAdd(g + k, v);
        else if (sah_word_is_min(w)) atomicMin(g + k, v);
        else atomicMax(g + k, v);
    }
}
template <typename P> __device__ __forceinline__ void sah_bin_add(P q, const BuildPrim& p, const float* c) {
    atomicAdd(q, 1u);
    for (int k = 0; k < 3; ++k) { atomicMin(q + 1 + k, enc_f(p.lo[k])); atomicMax(q + 4 + k, enc_f(p.hi[k])); atomicMin(q + 7 + k, enc_f(c[k])); atomicMax(q + 10 + k, enc_f(c[k])); }
}
__global__ __launch_bounds__(256) void sah_bin_kernel(const BuildPrim* __restrict__ prims, const int* __restrict__ seg, int n, const SahSeg* __restrict__ segs,
                                                       const int* __restrict__ slot_of, SahBin* __restrict__ bins) {
    // a large segment has more than 64 leaves, so the 64 positions of a wave belong to at most TWO large segments (and to small or
    // finished ones, which are not binned): two sets of bins per wave serve every case without a global atomic per leaf
    __shared__ uint32_t block_bins[SAH_BIN_WORDS];
    __shared__ uint32_t wave_bins[4][2][SAH_BIN_WORDS];
    const int i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int s = i < n ? seg[i] : -1;
    const int b0 = blockIdx.x * blockDim.x, b1 = b0 + int(blockDim.x) - 1;
    const int bf = seg[min(n - 1, b0)], bl = seg[min(n - 1, b1)];
    const bool block_uniform = bf >= 0 && bf == bl && b1 < n; // (the same in every thread)
    // the wave's (up to) two large segments: that of its first binned lane, and that of its last
    const unsigned long long act = __ballot(s >= 0);
    const int sa = act ? __shfl(s, __ffsll((long long)act) - 1, 64) : -1, sb = act ? __shfl(s, 63 - __clzll((long long)act), 64) : -1;
    if (block_uniform) sah_bins_clear(block_bins, threadIdx.x, blockDim.x);
    else if (act) { sah_bins_clear(wave_bins[wib][0], lane, 64); if (sb != sa) sah_bins_clear(wave_bins[wib][1], lane, 64); }
    __syncthreads();
    if (s >= 0) {
        const BuildPrim p = prims[i];
        const SahSeg sg = segs[s];
        float c[3];
        for (int a = 0; a < 3; ++a) c[a] = 0.5f * (p.lo[a] + p.hi[a]);
        uint32_t* mine = block_uniform ? block_bins : wave_bins[wib][s == sa ? 0 : 1];
        for (int a = 0; a < 3; ++a) sah_bin_add(mine + (a * SAH_BINS + sah_bin_of(c[a], sg.clo[a], sg.chi[a])) * 13, p, c);
    }
    __syncthreads();
    if (block_uniform) sah_bins_flush(block_bins, reinterpret_cast<uint32_t*>(bins + size_t(slot_of[bf]) * 3 * SAH_BINS), threadIdx.x, blockDim.x);
    else if (act) {
        sah_bins_flush(wave_bins[wib][0], reinterpret_cast<uint32_t*>(bins + size_t(slot_of[sa]) * 3 * SAH_BINS), lane, 64);
        if (sb != sa) sah_bins_flush(wave_bins[wib][1], reinterpret_cast<uint32_t*>(bins + size_t(slot_of[sb]) * 3 * SAH_BINS), lane, 64);
    }
}
struct SahLists { int n_large, n_small; }; // the NEXT level's segment counts (appended to with atomics)
__device__ __forceinline__ void sah_make_child(int node, int begin, int count, const float* clo, const float* chi, SahSeg* __restrict__ segs,
                                               int* __restrict__ slot_of, int* __restrict__ next_large, int* __restrict__ next_small,
                                               SahLists* __restrict__ next) {
    if (count < 2) return;
    SahSeg sg;
    sg.begin = begin; sg.count = count;
    for (int a = 0; a < 3; ++a) { sg.clo[a] = clo[a]; sg.chi[a] = chi[a]; }
    segs[node] = sg;
    if (count > SAH_SMALL) { const int k = atomicAdd(&next->n_large, 1); next_large[k] = node; slot_of[node] = k; } // its bins next level
    else next_small[atomicAdd(&next->n_small, 1)] = node;
}
// One thread per large segment: the cheapest of the 3 x (SAH_BINS - 1) planes (scene_lower.cpp split()), the node record, the children.
__global__ void sah_eval_kernel(const int* __restrict__ large, int n_large, const SahBin* __restrict__ bins, SahSeg* __restrict__ segs,
                                SahSplit* __restrict__ split, BvhNode* __restrict__ out, int* __restrict__ slot_of, int* __restrict__ next_large,
                                int* __restrict__ next_small, SahLists* __restrict__ next) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_large) return;
    const int s = large[k];
    const SahSeg sg = segs[s];
    const SahBin* bs = bins + size_t(k) * 3 * SAH_BINS;
    float best = INFINITY;
    int best_axis = -1, best_bin = -1;
    for (int a = 0; a < 3; ++a) {
        if (!(sg.chi[a] - sg.clo[a] > 0.f)) continue;
        float r_area[SAH_BINS];
        uint32_t r_cnt[SAH_BINS];
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        uint32_t cnt = 0;
        for (int b = SAH_BINS - 1; b > 0; --b) {
            const SahBin q = bs[a * SAH_BINS + b];
            if (q.count) for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], dec_f(q.lo[c])); hi[c] = fmaxf(hi[c], dec_f(q.hi[c])); }
            cnt += q.count;
            r_area[b] = box_area(lo, hi); r_cnt[b] = cnt;
        }
        for (int c = 0; c < 3; ++c) { lo[c] = INFINITY; hi[c] = -INFINITY; }
        cnt = 0;
        for (int b = 0; b < SAH_BINS - 1; ++b) {
            const SahBin q = bs[a * SAH_BINS + b];
            if (q.count) for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], dec_f(q.lo[c])); hi[c] = fmaxf(hi[c], dec_f(q.hi[c])); }
            cnt += q.count;
            if (cnt == 0 || r_cnt[b + 1] == 0) continue;
            const float cost = box_area(lo, hi) * float(cnt) + r_area[b + 1] * float(r_cnt[b + 1]);
            if (cost < best) { best = cost; best_axis = a; best_bin = b; }
        }
    }
    SahSplit sp;
    float llo[3] = {INFINITY, INFINITY, INFINITY}, lhi[3] = {-INFINITY, -INFINITY, -INFINITY}, rlo[3] = {INFINITY, INFINITY, INFINITY},
          rhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float lclo[3] = {INFINITY, INFINITY, INFINITY}, lchi[3] = {-INFINITY, -INFINITY, -INFINITY}, rclo[3] = {INFINITY, INFINITY, INFINITY},
          rchi[3] = {-INFINITY, -INFINITY, -INFINITY};
    int nl = 0;
    if (best_axis >= 0) {
        for (int b = 0; b < SAH_BINS; ++b) {
            const SahBin q = bs[best_axis * SAH_BINS + b];
            if (!q.count) continue;
            const bool left = b <= best_bin;
            for (int c = 0; c < 3; ++c) {
                if (left) { llo[c] = fminf(llo[c], dec_f(q.lo[c])); lhi[c] = fmaxf(lhi[c], dec_f(q.hi[c])); lclo[c] = fminf(lclo[c], dec_f(q.clo[c])); lchi[c] = fmaxf(lchi[c], dec_f(q.chi[c])); }
                else { rlo[c] = fminf(rlo[c], dec_f(q.lo[c])); rhi[c] = fmaxf(rhi[c], dec_f(q.hi[c])); rclo[c] = fminf(rclo[c], dec_f(q.clo[c])); rchi[c] = fmaxf(rchi[c], dec_f(q.chi[c])); }
            }
            if (left) nl += int(q.count);
        }
        sp.axis = best_axis; sp.bin = best_bin; sp.nl = nl; sp.median = 0;
    } else {
        // every centroid in one bin on every axis (coincident centroids): halve by position; the children's boxes and centroid
        // bounds are the parent's (conservative: boxes only cull)
        const SahBin q0 = bs[0];
        (void)q0;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < SAH_BINS; ++b) {
                const SahBin q = bs[a * SAH_BINS + b];
                if (!q.count) continue;
                for (int c = 0; c < 3; ++c) { llo[c] = fminf(llo[c], dec_f(q.lo[c])); lhi[c] = fmaxf(lhi[c], dec_f(q.hi[c])); }
            }
        for (int c = 0; c < 3; ++c) { rlo[c] = llo[c]; rhi[c] = lhi[c]; lclo[c] = rclo[c] = sg.clo[c]; lchi[c] = rchi[c] = sg.chi[c]; }
        nl = sg.count / 2;
        sp.axis = 0; sp.bin = 0; sp.nl = nl; sp.median = 1;
    }
    split[s] = sp;
    const int nr = sg.count - nl;
    BvhNode nd;
    for (int c = 0; c < 3; ++c) { nd.lo0[c] = llo[c]; nd.hi0[c] = lhi[c]; nd.lo1[c] = rlo[c]; nd.hi1[c] = rhi[c]; }
    nd.child0 = nl > 1 ? s + 1 : CHILD_EMPTY;  // a single leaf's code is filled in by the partition kernel
    nd.child1 = nr > 1 ? s + nl : CHILD_EMPTY;
    nd.pad0 = nd.pad1 = 0;
    out[s] = nd;
    sah_make_child(s + 1, sg.begin, nl, lclo, lchi, segs, slot_of, next_large, next_small, next);
    sah_make_child(s + nl, sg.begin + nl, nr, rclo, rchi, segs, slot_of, next_large, next_small, next);
}
__global__ void sah_flag_kernel(const BuildPrim* __restrict__ prims, const int* __restrict__ seg, int n, const SahSeg* __restrict__ segs,
                                const SahSplit* __restrict__ split, uint32_t* __restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = seg[i];
    uint32_t f = 0;
    if (s >= 0) {
        const SahSplit sp = split[s];
        const SahSeg sg = segs[s];
        if (sp.median) f = (i - sg.begin) < sp.nl;
        else {
            const BuildPrim p = prims[i];
            f = sah_bin_of(0.5f * (p.lo[sp.axis] + p.hi[sp.axis]), sg.clo[sp.axis], sg.chi[sp.axis]) <= sp.bin;
        }
    }
    flag[i] = f;
}
__global__ void sah_partition_kernel(const BuildPrim* __restrict__ prims, const int* __restrict__ seg, int n, const SahSeg* __restrict__ segs,
                                     const SahSplit* __restrict__ split, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ scanned,
                                     BuildPrim* __restrict__ prims_out, int* __restrict__ seg_out, BvhNode* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = seg[i];
    if (s < 0) { seg_out[i] = -1; return; } // (the output array still holds this position's owner of two levels ago)
    const SahSplit sp = split[s];
    const SahSeg sg = segs[s];
    const bool left = flag[i] != 0u;
    const int lrank = int(scanned[i] - scanned[sg.begin]);
    const int at = left ? sg.begin + lrank : sg.begin + sp.nl + ((i - sg.begin) - lrank);
    const BuildPrim p = prims[i];
    prims_out[at] = p;
    const int side_count = left ? sp.nl : sg.count - sp.nl;
    const int child = left ? s + 1 : s + sp.nl;
    seg_out[at] = side_count > SAH_SMALL ? child : -1; // small segments go by their own list, single leaves are finished
    if (side_count == 1) { if (left) out[s].child0 = p.leaf; else out[s].child1 = p.leaf; }
}
// One wave per small segment (2 .. SAH_SMALL leaves): the WHOLE subtree below it, by exact sweeps on the three axes.  The
// segment's leaves sit in the wave's strip of LDS, one per lane; the wave splits a node (lane i tries "everything up to leaf i's
// centroid goes left" on each axis, the wave takes the cheapest candidate; ties by (cost, axis, lane)), re-orders the node's leaves
// in place and pushes the children with two or more leaves on a stack in LDS, until the stack is empty: no launch per level and
// no traffic but the node records.  (One launch per level with a wave per node measured 11 ms for 10^6 leaves; this form ~1 ms.)
constexpr int SAH_WAVES_PER_BLOCK = 4;
__global__ __launch_bounds__(64 * SAH_WAVES_PER_BLOCK) void sah_small_kernel(const int* __restrict__ small, int n_small, const BuildPrim* __restrict__ prims,
                                                                            const SahSeg* __restrict__ segs, BvhNode* __restrict__ out, int level,
                                                                            int* __restrict__ height) {
    // (level: the depth of the segments of the list, root = 1; height: the deepest inner node of the tree, gathered here)
    __shared__ BuildPrim buf_all[SAH_WAVES_PER_BLOCK][SAH_SMALL];
    __shared__ int stack_all[SAH_WAVES_PER_BLOCK][SAH_SMALL][4];
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int w = blockIdx.x * SAH_WAVES_PER_BLOCK + wib;
    if (w >= n_small) return;
    BuildPrim* buf = buf_all[wib];
    int(*stack)[4] = stack_all[wib];
    {
        const int s0 = small[w];
        const SahSeg sg = segs[s0];
        if (lane < sg.count) buf[lane] = prims[sg.begin + lane];
        if (lane == 0) { stack[0][0] = s0; stack[0][1] = 0; stack[0][2] = sg.count; stack[0][3] = level; }
    }
    int sp = 1, deepest = level; // wave-uniform
    while (sp > 0) {
        --sp;
        const int s = stack[sp][0], begin = stack[sp][1], count = stack[sp][2], depth = stack[sp][3];
        deepest = max(deepest, depth);
        const bool on = lane < count;
        BuildPrim p;
        if (on) p = buf[begin + lane];
        else { for (int a = 0; a < 3; ++a) { p.lo[a] = INFINITY; p.hi[a] = -INFINITY; } p.leaf = CHILD_EMPTY; p.pad = 0; }
        float c[3];
        for (int a = 0; a < 3; ++a) c[a] = on ? 0.5f * (p.lo[a] + p.hi[a]) : INFINITY;
        float best = INFINITY;
        int best_axis = 0;
        float bl[6], br[6]; // the best candidate's two boxes
        for (int k = 0; k < 3; ++k) { bl[k] = br[k] = INFINITY; bl[3 + k] = br[3 + k] = -INFINITY; }
        for (int a = 0; a < 3; ++a) {
            float l[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}, r[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
            int nl = 0;
            for (int j = 0; j < count; ++j) {
                const BuildPrim q = buf[begin + j]; // the same address in every lane: an LDS broadcast
                const float cj = 0.5f * (q.lo[a] + q.hi[a]);
                const bool in_left = cj < c[a] || (cj == c[a] && j <= lane);
                for (int k = 0; k < 3; ++k) {
                    if (in_left) { l[k] = fminf(l[k], q.lo[k]); l[3 + k] = fmaxf(l[3 + k], q.hi[k]); }
                    else { r[k] = fminf(r[k], q.lo[k]); r[3 + k] = fmaxf(r[3 + k], q.hi[k]); }
                }
                nl += in_left;
            }
            const int nr = count - nl;
            const float cost = (on && nr > 0) ? box_area(l, l + 3) * float(nl) + box_area(r, r + 3) * float(nr) : INFINITY;
            if (cost < best) { best = cost; best_axis = a; for (int k = 0; k < 6; ++k) { bl[k] = l[k]; br[k] = r[k]; } }
        }
        // the wave's cheapest candidate: (cost, axis, lane) lexicographically, so that ties depend on nothing but the data
        float wc = best;
        int wa = best_axis, wl = lane;
        for (int off = 32; off > 0; off >>= 1) {
            const float oc = __shfl_xor(wc, off, 64);
            const int oa = __shfl_xor(wa, off, 64), ol = __shfl_xor(wl, off, 64);
            if (oc < wc || (oc == wc && (oa < wa || (oa == wa && ol < wl)))) { wc = oc; wa = oa; wl = ol; }
        }
        const float ca = wa == 0 ? c[0] : (wa == 1 ? c[1] : c[2]);
        const float cw = __shfl(ca, wl, 64);
        const bool in_left = on && (ca < cw || (ca == cw && lane <= wl));
        const unsigned long long lm = __ballot(in_left), rm = __ballot(on && !in_left);
        int nl = __popcll(lm), nr = __popcll(rm);
        float lbox[6], rbox[6];
        for (int k = 0; k < 6; ++k) { lbox[k] = __shfl(bl[k], wl, 64); rbox[k] = __shfl(br[k], wl, 64); }
        const bool med = !(wc < INFINITY) || nl == 0 || nr == 0; // boxes that are not finite: halve by position, both children take the union
        if (med) {
            nl = count / 2; nr = count - nl;
            float u[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
            for (int j = 0; j < count; ++j) {
                const BuildPrim q = buf[begin + j];
                for (int k = 0; k < 3; ++k) { u[k] = fminf(u[k], q.lo[k]); u[3 + k] = fmaxf(u[3 + k], q.hi[k]); }
            }
            for (int k = 0; k < 6; ++k) lbox[k] = rbox[k] = u[k];
        }
        const bool go_left = med ? (on && lane < nl) : in_left;
        const unsigned long long glm = __ballot(go_left), grm = __ballot(on && !go_left), below = (1ull << lane) - 1ull;
        // (every lane holds its leaf in registers: the strip can be rewritten in place)
        if (on) buf[begin + (go_left ? __popcll(glm & below) : nl + __popcll(grm & below))] = p;
        const int first_l = __ffsll((long long)glm) - 1, first_r = __ffsll((long long)grm) - 1;
        const int leaf_l = __shfl(p.leaf, max(first_l, 0), 64), leaf_r = __shfl(p.leaf, max(first_r, 0), 64);
        if (lane == 0) {
            BvhNode nd;
            for (int k = 0; k < 3; ++k) { nd.lo0[k] = lbox[k]; nd.hi0[k] = lbox[3 + k]; nd.lo1[k] = rbox[k]; nd.hi1[k] = rbox[3 + k]; }
            nd.child0 = nl > 1 ? s + 1 : leaf_l;
            nd.child1 = nr > 1 ? s + nl : leaf_r;
            nd.pad0 = nd.pad1 = 0;
            out[s] = nd;
            int t = sp;
            if (nr > 1) { stack[t][0] = s + nl; stack[t][1] = begin + nl; stack[t][2] = nr; stack[t][3] = depth + 1; ++t; }
            if (nl > 1) { stack[t][0] = s + 1; stack[t][1] = begin; stack[t][2] = nl; stack[t][3] = depth + 1; ++t; }
        }
        sp += (nl > 1) + (nr > 1);
    }
    if (lane == 0) atomicMax(height, deepest);
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
// (two passes and a scan instead of an atomic counter: the frontier's ORDER numbers the records in level order, and a build must repeat bit for bit)
__global__ void collapse_count_kernel(const BvhNode* __restrict__ nodes, const int* __restrict__ frontier, int n_front, uint32_t* __restrict__ n_inner_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_front) return;
    Slot slots[5];
    const int n = expand4(nodes, frontier[i], slots);
    uint32_t n_inner = 0;
    for (int c = 0; c < n; ++c) n_inner += slots[c].child >= 0;
    n_inner_out[i] = n_inner;
}
__global__ void collapse_mark_kernel(const BvhNode* __restrict__ nodes, const int* __restrict__ frontier, int n_front, int* __restrict__ next,
                                     const uint32_t* __restrict__ offset, uint32_t* __restrict__ is_head) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_front) return;
    Slot slots[5];
    const int n = expand4(nodes, frontier[i], slots);
    uint32_t at = offset[i]; // the record's inner children side by side in the next frontier, in slot order
    for (int c = 0; c < n; ++c)
        if (slots[c].child >= 0) {
            is_head[slots[c].child] = 1u;
            next[at++] = slots[c].child;
        }
}
// Level order: record number = position of its head in the frontier arrays (root first, the inner children of a record side by side).
__global__ void rank_by_position_kernel(const int* __restrict__ heads, int n_heads, uint32_t* __restrict__ rank) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_heads) rank[heads[i]] = uint32_t(i);
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

int lbvh_build_device_tree(const BuildPrim* prims, size_t n, const float* centroid_bounds, bool sah, DeviceTree& tree, double* kernel_ms, std::string& err) {
    if (n < 2 || n >= (size_t(1) << 26)) { err = "lbvh_build: needs 2 .. 2^26-1 leaves"; return -1; }
    int rc = 0;
    const auto wall0 = std::chrono::steady_clock::now();
    auto wall_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count(); };
    double t_alloc = 0, t_up = 0, t_kernels = 0;
    // centroid bounds on the host (the leaves come from the host anyway)
    float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (centroid_bounds) {
        for (int a = 0; a < 3; ++a) { cmin[a] = centroid_bounds[a]; cmax[a] = centroid_bounds[3 + a]; }
    } else {
        for (size_t i = 0; i < n; ++i)
            for (int a = 0; a < 3; ++a) {
                const float c = 0.5f * (prims[i].lo[a] + prims[i].hi[a]);
                cmin[a] = std::min(cmin[a], c);
                cmax[a] = std::max(cmax[a], c);
            }
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
    void *d_temp = nullptr, *d_temp2 = nullptr, *d_temp3 = nullptr;
    size_t temp_bytes = 0, temp2_bytes = 0, temp3_bytes = 0;
    // binned SAH
    BuildPrim *d_pa = nullptr, *d_pb = nullptr;
    int *d_sa = nullptr, *d_sb = nullptr, *d_slot = nullptr, *d_list[4] = {nullptr, nullptr, nullptr, nullptr}; // lists: large a/b, small a/b
    SahSeg* d_segs = nullptr;
    SahSplit* d_split = nullptr;
    SahBin* d_bins = nullptr;
    SahLists* d_next = nullptr;
    uint32_t *d_flag = nullptr, *d_scan = nullptr;
    int sah_levels = 0;
    const size_t max_large = n / size_t(SAH_SMALL + 1) + 2;
    // collapse
    int* d_heads = nullptr;
    uint32_t *d_is_head = nullptr, *d_rank = nullptr, *d_need = nullptr, *d_fcount = nullptr, *d_foff = nullptr;
    Bvh4Node* d_out4 = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const uint32_t nb = uint32_t((n + 255) / 256);
    const int n_inner = int(n - 1);
    int h_levels = 0;
    std::vector<int> level_off; // frontier k = heads[level_off[k] .. level_off[k + 1])
    uint32_t n_heads = 0, h_need = 0;
    {
        LBVH_TRY(hipGetDevice(&device));
        // both hierarchies: the leaves, the pre-ordered binary records, the collapse's arrays
        LBVH_TRY(hipMalloc((void**)&d_prims, n * sizeof(BuildPrim)));
        LBVH_TRY(hipMalloc((void**)&d_levels, (n - 1) * 4));
        LBVH_TRY(hipMalloc((void**)&d_out2, (n - 1) * sizeof(BvhNode)));
        LBVH_TRY(hipMalloc((void**)&d_heads, (n - 1) * 4));
        LBVH_TRY(hipMalloc((void**)&d_is_head, (n - 1) * 4));
        LBVH_TRY(hipMalloc((void**)&d_rank, (n - 1) * 4));
        LBVH_TRY(hipMalloc((void**)&d_fcount, (n - 1) * 4));
        LBVH_TRY(hipMalloc((void**)&d_foff, (n - 1) * 4));
        LBVH_TRY(rocprim::exclusive_scan(nullptr, temp2_bytes, d_is_head, d_rank, 0u, size_t(n_inner), rocprim::plus<uint32_t>(), hipStream_t(0)));
        LBVH_TRY(hipMalloc(&d_temp2, std::max<size_t>(temp2_bytes, 16)));
        if (!sah) {
            LBVH_TRY(hipMalloc((void**)&d_keys, n * 8));
            LBVH_TRY(hipMalloc((void**)&d_keys2, n * 8));
            LBVH_TRY(hipMalloc((void**)&d_order, n * 4));
            LBVH_TRY(hipMalloc((void**)&d_order2, n * 4));
            LBVH_TRY(hipMalloc((void**)&d_children, (n - 1) * sizeof(int2)));
            LBVH_TRY(hipMalloc((void**)&d_node_parent, (n - 1) * 4));
            LBVH_TRY(hipMalloc((void**)&d_done, (n - 1) * 4));
            LBVH_TRY(hipMalloc((void**)&d_box, (n - 1) * sizeof(NodeBox)));
            LBVH_TRY(hipMalloc((void**)&d_out, (n - 1) * sizeof(BvhNode)));
            LBVH_TRY(hipMalloc((void**)&d_pos, (n - 1) * 4));
            LBVH_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys, d_keys2, d_order, d_order2, n, 0, 63, hipStream_t(0)));
            LBVH_TRY(hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 16)));
        } else {
            LBVH_TRY(hipMalloc((void**)&d_pa, n * sizeof(BuildPrim)));
            LBVH_TRY(hipMalloc((void**)&d_pb, n * sizeof(BuildPrim)));
            LBVH_TRY(hipMalloc((void**)&d_sa, n * 4));
            LBVH_TRY(hipMalloc((void**)&d_sb, n * 4));
            LBVH_TRY(hipMalloc((void**)&d_slot, n * 4));
            LBVH_TRY(hipMalloc((void**)&d_list[0], max_large * 4));
            LBVH_TRY(hipMalloc((void**)&d_list[1], max_large * 4));
            LBVH_TRY(hipMalloc((void**)&d_list[2], (n / 2 + 2) * 4));
            LBVH_TRY(hipMalloc((void**)&d_list[3], (n / 2 + 2) * 4));
            LBVH_TRY(hipMalloc((void**)&d_segs, n * sizeof(SahSeg)));
            LBVH_TRY(hipMalloc((void**)&d_split, n * sizeof(SahSplit)));
            LBVH_TRY(hipMalloc((void**)&d_bins, max_large * 3 * SAH_BINS * sizeof(SahBin)));
            LBVH_TRY(hipMalloc((void**)&d_next, sizeof(SahLists)));
            LBVH_TRY(hipMalloc((void**)&d_flag, n * 4));
            LBVH_TRY(hipMalloc((void**)&d_scan, n * 4));
            LBVH_TRY(rocprim::exclusive_scan(nullptr, temp3_bytes, d_flag, d_scan, 0u, n, rocprim::plus<uint32_t>(), hipStream_t(0)));
            LBVH_TRY(hipMalloc(&d_temp3, std::max<size_t>(temp3_bytes, 16)));
        }
        LBVH_TRY(hipEventCreate(&e0));
        LBVH_TRY(hipEventCreate(&e1));
        t_alloc = wall_ms();
        LBVH_TRY(hipMemcpy(d_prims, prims, n * sizeof(BuildPrim), hipMemcpyHostToDevice));
        t_up = wall_ms();

        LBVH_TRY(hipEventRecord(e0, 0));
        if (sah) {
            // level 0: the root segment over all leaves, with the centroid bounds the host has already
            SahSeg root;
            root.begin = 0; root.count = int(n);
            for (int a = 0; a < 3; ++a) { root.clo[a] = cmin[a]; root.chi[a] = cmax[a]; }
            const int zero = 0;
            LBVH_TRY(hipMemcpyAsync(d_segs, &root, sizeof(root), hipMemcpyHostToDevice, 0));
            LBVH_TRY(hipMemcpyAsync(d_slot, &zero, 4, hipMemcpyHostToDevice, 0));
            int n_large = n > size_t(SAH_SMALL) ? 1 : 0, n_small = 1 - n_large;
            LBVH_TRY(hipMemcpyAsync(d_list[n_large ? 0 : 2], &zero, 4, hipMemcpyHostToDevice, 0));
            LBVH_TRY(hipMemsetAsync(d_sa, n_large ? 0 : 0xFF, n * 4, 0)); // every position in segment 0 (large) or none (small)
            LBVH_TRY(hipMemcpyAsync(d_pa, d_prims, n * sizeof(BuildPrim), hipMemcpyDeviceToDevice, 0));
            LBVH_TRY(hipMemsetAsync(d_levels, 0, 4, 0)); // the height reached inside the small subtrees
            int cur = 0; // which of the ping-pong arrays / lists is the level's input
            while (n_large > 0 || n_small > 0) {
                if (++sah_levels > 4096) { err = "lbvh_build: the SAH build did not converge"; rc = -4; goto done; }
                BuildPrim *pin = cur ? d_pb : d_pa, *pout = cur ? d_pa : d_pb;
                int *sin = cur ? d_sb : d_sa, *sout = cur ? d_sa : d_sb;
                int *large_in = d_list[cur], *large_out = d_list[cur ^ 1], *small_in = d_list[2 + cur], *small_out = d_list[2 + (cur ^ 1)];
                LBVH_TRY(hipMemsetAsync(d_next, 0, sizeof(SahLists), 0));
                if (n_large > 0) {
                    const int n_bins = n_large * 3 * SAH_BINS;
                    hipLaunchKernelGGL(sah_bins_init_kernel, dim3((n_bins + 255) / 256), dim3(256), 0, 0, d_bins, n_bins);
                    hipLaunchKernelGGL(sah_bin_kernel, dim3(nb), dim3(256), 0, 0, pin, sin, int(n), d_segs, d_slot, d_bins);
                    hipLaunchKernelGGL(sah_eval_kernel, dim3((n_large + 63) / 64), dim3(64), 0, 0, large_in, n_large, d_bins, d_segs, d_split, d_out2, d_slot,
                                       large_out, small_out, d_next);
                    hipLaunchKernelGGL(sah_flag_kernel, dim3(nb), dim3(256), 0, 0, pin, sin, int(n), d_segs, d_split, d_flag);
                    LBVH_TRY(rocprim::exclusive_scan(d_temp3, temp3_bytes, d_flag, d_scan, 0u, n, rocprim::plus<uint32_t>(), hipStream_t(0)));
                    hipLaunchKernelGGL(sah_partition_kernel, dim3(nb), dim3(256), 0, 0, pin, sin, int(n), d_segs, d_split, d_flag, d_scan, pout, sout, d_out2);
                }
                if (n_small > 0) // (the segments of 64 leaves or fewer that the large splits of the last level made: each a whole subtree)
                    hipLaunchKernelGGL(sah_small_kernel, dim3((n_small + SAH_WAVES_PER_BLOCK - 1) / SAH_WAVES_PER_BLOCK), dim3(64 * SAH_WAVES_PER_BLOCK), 0, 0,
                                       small_in, n_small, pin, d_segs, d_out2, sah_levels, d_levels);
                SahLists h_next;
                LBVH_TRY(hipMemcpy(&h_next, d_next, sizeof(h_next), hipMemcpyDeviceToHost));
                if (size_t(h_next.n_large) > max_large || size_t(h_next.n_small) > n / 2 + 2) { err = "lbvh_build: SAH segment lists overflowed"; rc = -4; goto done; }
                n_large = h_next.n_large; n_small = h_next.n_small;
                cur ^= 1;
            }
        } else {
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
        }

        // ---- collapse: heads level by level (root first), numbering, records, stack bound
        LBVH_TRY(hipMemsetAsync(d_is_head, 0, size_t(n_inner) * 4, 0));
        {
            static const uint32_t one = 1u; // (static: the source of an asynchronous copy must outlive the call)
            static const int zero = 0;
            LBVH_TRY(hipMemcpyAsync(d_is_head, &one, 4, hipMemcpyHostToDevice, 0)); // the root (binary node 0) heads record 0
            LBVH_TRY(hipMemcpyAsync(d_heads, &zero, 4, hipMemcpyHostToDevice, 0));
        }
        level_off.push_back(0);
        level_off.push_back(1);
        for (;;) {
            const int lo = level_off[level_off.size() - 2], hi = level_off.back(), n_front = hi - lo;
            hipLaunchKernelGGL(collapse_count_kernel, dim3((n_front + 255) / 256), dim3(256), 0, 0, d_out2, d_heads + lo, n_front, d_fcount);
            LBVH_TRY(rocprim::exclusive_scan(d_temp2, temp2_bytes, d_fcount, d_foff, 0u, size_t(n_front), rocprim::plus<uint32_t>(), hipStream_t(0)));
            hipLaunchKernelGGL(collapse_mark_kernel, dim3((n_front + 255) / 256), dim3(256), 0, 0, d_out2, d_heads + lo, n_front, d_heads + hi, d_foff,
                               d_is_head);
            uint32_t last[2] = {0, 0};
            LBVH_TRY(hipMemcpy(&last[0], d_fcount + (n_front - 1), 4, hipMemcpyDeviceToHost));
            LBVH_TRY(hipMemcpy(&last[1], d_foff + (n_front - 1), 4, hipMemcpyDeviceToHost));
            const int added = int(last[0] + last[1]);
            if (added == 0) break;
            if (hi + added > n_inner || level_off.size() > 4096) { err = "lbvh_build: collapse did not converge"; rc = -4; goto done; }
            level_off.push_back(hi + added);
        }
        n_heads = uint32_t(level_off.back());
        // Numbering of the records.  Level order (default since round 6): a record's number is its head's position in the frontier arrays, so the
        // inner children of a record lie side by side — two 64-byte quantised records to a 128-byte line, and a walk that visits a record visits
        // 1.6 of its children on average (tests/hostsim/cache_model.hpp: 50.1 -> 45.1 node-record miss lines per sample on spheres_1m against
        // the binary tree's pre-order; treelets of 4 .. 64 records: 47 - 48).  RTTNW_NODE_ORDER=pre: the binary pre-order of rounds 1-5.
        const char* order_env = getenv("RTTNW_NODE_ORDER");
        if (order_env && std::string(order_env) == "pre")
            LBVH_TRY(rocprim::exclusive_scan(d_temp2, temp2_bytes, d_is_head, d_rank, 0u, size_t(n_inner), rocprim::plus<uint32_t>(), hipStream_t(0)));
        else
            hipLaunchKernelGGL(rank_by_position_kernel, dim3((n_heads + 255) / 256), dim3(256), 0, 0, d_heads, int(n_heads), d_rank);
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
        if (sah) h_levels = std::max(h_levels, sah_levels); // the rounds of large segments are a level each; the small subtrees report theirs
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
            fprintf(stderr, "[lbvh] %s, %zu leaves -> %u 4-wide records in %zu levels: centroid bounds + allocations %.1f ms, upload %.1f ms, kernels %.1f ms (device %.2f); nothing downloaded\n",
                    sah ? ("binned SAH, " + std::to_string(sah_levels) + " rounds").c_str() : "Karras hierarchy", n, n_heads, level_off.size() - 1, t_alloc, t_up - t_alloc, t_kernels - t_up, ms);
    }
done:
    for (void* p : {(void*)d_prims, (void*)d_keys, (void*)d_keys2, (void*)d_order, (void*)d_order2, (void*)d_children, (void*)d_node_parent,
                    (void*)d_done, (void*)d_levels, (void*)d_box, (void*)d_out, (void*)d_out2, (void*)d_pos, d_temp, d_temp2, (void*)d_heads,
                    (void*)d_is_head, (void*)d_rank, (void*)d_fcount, (void*)d_foff, (void*)d_need, (void*)d_out4, d_temp3, (void*)d_pa, (void*)d_pb, (void*)d_sa, (void*)d_sb,
                    (void*)d_slot, (void*)d_list[0], (void*)d_list[1], (void*)d_list[2], (void*)d_list[3], (void*)d_segs, (void*)d_split, (void*)d_bins,
                    (void*)d_next, (void*)d_flag, (void*)d_scan})
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


// ---------------------------------------------------------------------------------------------------------------------------------
// f32 4-wide records -> the quantised records of the f64 decoupled kernel (bvh_quant.hpp), on the current device: one thread per
// record, no order between them.
namespace {
__global__ void quant4_make_kernel(const Bvh4Node* __restrict__ nodes4, uint32_t n, Bvh4QNode* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Bvh4QNode o;
    quant4_make(nodes4, int32_t(i), o);
    out[i] = o;
}
} // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// The interleaved node + sphere buffer (bvh_build.hpp): one thread per record.
namespace {
__global__ void interleave_count_kernel(const Bvh4Node* __restrict__ nodes4, uint32_t n, uint8_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k = 0;
    for (int c = 0; c < 4; ++c) {
        const int32_t ch = nodes4[i].child[c];
        if (ch < 0 && ch != CHILD_EMPTY && leaf_kind(ch) == PRIM_SPHERE) k += leaf_count(ch);
    }
    out[i] = uint8_t(k);
}
__device__ __forceinline__ void copy16(void* dst, const void* src, uint32_t bytes) {
    for (uint32_t b = 0; b < bytes; b += 16) *reinterpret_cast<uint4*>((char*)dst + b) = *reinterpret_cast<const uint4*>((const char*)src + b);
}
__global__ void interleave_scatter_kernel(InterleaveArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n4) return;
    Bvh4QNode q;
    quant4_make(a.nodes4, int32_t(i), q);
    const uint32_t su = a.sphere_bytes / 16u;
    uint32_t at = a.noff[i] + 4u; // the record's spheres follow it
    for (int c = 0; c < 4; ++c) {
        const int32_t ch = a.nodes4[i].child[c];
        if (ch == CHILD_EMPTY) continue;
        if (ch >= 0) { q.child[c] = int32_t(a.noff[ch] >> 2); continue; }
        if (leaf_kind(ch) != PRIM_SPHERE) continue; // other kinds keep their records where they are
        const uint32_t cnt = leaf_count(ch), first = leaf_first(ch), at_sphere = at / su;
        q.child[c] = make_leaf(PRIM_SPHERE, cnt, at_sphere);
        for (uint32_t k = 0; k < cnt; ++k) {
            copy16((char*)a.buffer + size_t(at + k * su) * 16u, (const char*)a.spheres + size_t(first + k) * a.sphere_bytes, a.sphere_bytes);
            a.seq_out[at_sphere + k] = a.sphere_seq[first + k];
            copy16((char*)a.mats_out + size_t(at_sphere + k) * a.mat_bytes, (const char*)a.mats + size_t(first + k) * a.mat_bytes, a.mat_bytes);
        }
        at += cnt * su;
    }
    *reinterpret_cast<Bvh4QNode*>((char*)a.buffer + size_t(a.noff[i]) * 16u) = q;
}
} // namespace

int interleave_count_device(const Bvh4Node* d_nodes4, uint32_t n, uint8_t* d_sphere_count, std::string& err) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(interleave_count_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d_nodes4, n, d_sphere_count);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { err = std::string("interleave_count: ") + hipGetErrorString(e); return -4; }
    return 0;
}
int interleave_build_device(const InterleaveArgs& a, std::string& err) {
    if (a.n4 == 0) return 0;
    hipLaunchKernelGGL(interleave_scatter_kernel, dim3((a.n4 + 127) / 128), dim3(128), 0, 0, a);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { err = std::string("interleave_build: ") + hipGetErrorString(e); return -4; }
    return 0;
}

int quant4_build_device(const Bvh4Node* d_nodes4, uint32_t n, Bvh4QNode* d_out, std::string& err) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(quant4_make_kernel, dim3((n + 127) / 128), dim3(128), 0, 0, d_nodes4, n, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { err = std::string("quant4_build: ") + hipGetErrorString(e); return -4; }
    return 0;
}

} // namespace rt
