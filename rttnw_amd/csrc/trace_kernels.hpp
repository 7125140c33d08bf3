// trace_kernels.hpp — the kernels of the device half of include/rttnw_hip.h for gfx950 (MI355X), templates over the arithmetic type;
// instantiated per precision in render_f32.hip / render_f64.hip.
//
// One kernel does the whole per-pixel sample loop of main.rs:202-229:
//   * persistent workgroups pull JOBS from one global counter.  A job is (pixel, chunk of samples) — rt_types.hpp
//     plan_chunks, rt_core.hpp job_decode: 64 consecutive jobs are a 2x2 pixel block x 16 chunks, so the lanes of a
//     wave start nearly the same ray.  Idle lanes are counted with __ballot and pick distinct jobs by popcount rank
//     out of a batch of 256 job indices the wave reserved with ONE atomic — lanes never wait for the longest path in
//     the wave (path lengths run 1..50, the Cornell blocks trap rays).
//   * a lane folds its job's samples sequentially (main.rs:211) with one path per lane: regenerate a
//     camera ray when the path dies, otherwise do one world.hit + scatter (rt_core.hpp).
//   * BVH traversal keeps its stack in LDS, interleaved by lane (entry e of lane l at e*blockDim+l:
//     conflict-free ds_read/ds_write_b32); small scenes keep the node records there too.
//   * job sums go to a partial buffer; a resolve kernel adds a pixel's chunks in chunk order, so the
//     image is bit-identical whatever the scheduling, the grid size or the number of GPUs.
// Two forms of the loop (KERNELS.md): trace_kernel_plain (a lane owns a path; shade phases asynchronous to the walks) and
// trace_kernel (paths decoupled from lanes through wave-private queues, for trees that live in HBM).
// No CPU fallback: every entry point needs a HIP device.
#pragma once
#include "render_common.hpp"
#ifndef RT_ASYNC_SHADE
#define RT_ASYNC_SHADE 1 // the lane-owns-path kernel leaves its walk loop for a shade phase once at most RT_ASYNC_SLACK walks are unfinished; those are
                         // SUSPENDED — their lanes skip the phase and walk on in the next one (0: every round waits for its longest walk, rounds 1-4)
#endif
#ifndef RT_ASYNC_WHOLE_LEAF
#define RT_ASYNC_WHOLE_LEAF 0 // 1: a leaf step tests all (<= 4) records of the leaf instead of one
#endif
#ifndef RT_ASYNC_SLACK
#define RT_ASYNC_SLACK 8
#endif
#ifndef RT_WAVE_WHOLE_LEAF
#define RT_WAVE_WHOLE_LEAF 0 // decoupled kernels: a leaf step tests all (<= 4) records of the leaf instead of one
#endif
#include "trace_tally.hpp"

#ifndef RT_WAVE_QUANT
#define RT_WAVE_QUANT 2 // which decoupled kernels walk the quantised records (rt_types.hpp Bvh4QNode): 1 the f64 ones, 2 all, 0 none.  (Mid-round-4: f64 +4 / +7 %,
                        // f32 -2 % against the f32 records — and +4 % for half-precision node-local records, 80 of 128 bytes a visit, which the f32 kernel walked
                        // for a while.  At the round's end — no slot tests, no instance code, 13-real path slots — the f32 kernel moves 6.2 TB/s and little else,
                        // and half the node bytes are worth +10 %: spheres_1m f32 433 -> 476 Msamples/s on these records; the half-precision ones are gone.)
#endif
#ifndef RT_WAVE_RETIRE
#define RT_WAVE_RETIRE 8 // finished rays that end a burst while rays are queued (the hand-over of their lanes).  Round 4's end, spheres_1m f64 / f32 Msamples/s:
                         // 4 / 6 / 8 / 12 / 16 / 24 / 32 -> 342 / 340 / 342 / 339 / 336 / 323 / 308 and 490 / 484 / 489 / 489 / 483 / 464 / 449
#endif
#ifndef RT_WAVE_STEPS
#define RT_WAVE_STEPS 5 // node steps per trip of the decoupled kernel's bursts (2 / 3 / 4 / 6: 302 / 321 / 325 / 330 Msamples/s in round 1; round 4's end, spheres_1m
                        // f64 / strict / f32 with 4 / 5 / 6: 334 / 340 / 332, 331 / 332 / 331, 481 / 481 / 476; 8: f32 442)
#endif
#ifndef RT_F64_BLOCK
#define RT_F64_BLOCK 1024 // threads per block of the LDS-resident f64 kernel (4 waves/SIMD at 128 VGPRs; see the Makefile's f64 flags and profiles/r03/README.md)
#endif

namespace rt {
inline namespace RT_ARITH_NS {

// ---------------------------------------------------------------------------------------------
// device-side helpers
// ---------------------------------------------------------------------------------------------
// Traversal memory of a lane: its BVH stack — the first LDS_STACK_ENTRIES entries in LDS (entry e of lane l at
// e*stride + l: conflict-free b32 accesses), deeper ones in a strip of global memory (entry e of thread g at
// e*spill_stride + g; a 4-wide walk can have three pending children per level but rarely has more than a dozen) — and
// the way it reads node records.
typedef __attribute__((address_space(3))) int32_t* LdsIntPtr;    // explicit address spaces: the compiler otherwise merges
typedef __attribute__((address_space(1))) int32_t* GlobalIntPtr; // the two halves of get() into one FLAT load
template <uint32_t STRIDE, uint32_t ENTRIES = LDS_STACK_ENTRIES> struct LdsStack { // STRIDE = lanes sharing the LDS stack area: entry e of a lane at base[e * STRIDE]; ENTRIES kept in LDS
    static constexpr int WIDE = NODES_F32X4;    // walks Bvh4Node records
    static constexpr int SLAB_F32 = SLAB_EXACT; // node records in global memory: the f32 kernels' exact slab test (rt_core.hpp)
    static constexpr int SPARE = int(ENTRIES); // a lane's extra LDS slot: target of the node step's masked-off stores
    LdsIntPtr base;        // &lds[threadIdx.x]
    GlobalIntPtr spill;    // &spill_buffer[global thread]  (a wave-uniform base with the thread index added at the rare access saves the two registers
                           // and cost final_scene f64 3 %: 1653 -> 1599, the pushes' code in the node step grows)
    uint32_t spill_stride; // threads of the launch
    __device__ __forceinline__ void set(int i, int32_t v) {
        if (uint32_t(i) < ENTRIES) base[uint32_t(i) * STRIDE] = v;
        else spill[size_t(uint32_t(i) - ENTRIES) * spill_stride] = v;
    }
    __device__ __forceinline__ int32_t get(int i) const {
        if (uint32_t(i) < ENTRIES) return base[uint32_t(i) * STRIDE];
        return spill[size_t(uint32_t(i) - ENTRIES) * spill_stride];
    }
    // a node step that finds entries i, i+1, i+2 inside the LDS part writes them without looking at the spill strip
    __device__ __forceinline__ bool room_for_three(int i) const { return uint32_t(i) + 3u <= ENTRIES; }
    __device__ __forceinline__ void set_fast(int i, int32_t v) { base[uint32_t(i) * STRIDE] = v; }
    // node records in global memory: a plane piece is addressed by its index inside the 128-byte record
    __device__ __forceinline__ uint32_t plane_off(uint32_t q) const { return q; }
    template <typename R> __device__ __forceinline__ void fetch(const SceneView<R>& sc, int32_t i, const uint32_t* near_off, Planes4& out) const {
        const int4* rec = reinterpret_cast<const int4*>(sc.nodes + i);
        union { int4 q[7]; struct { float nr[3][4], fr[3][4]; int32_t child[4]; } p; } u;
#pragma unroll
        for (uint32_t a = 0; a < 3; ++a) {
            u.q[a] = rec[near_off[a]];
            u.q[3 + a] = rec[2u * a + 3u - near_off[a]]; // the other one of (a, a + 3)
        }
        u.q[6] = rec[6];
        __builtin_memcpy(&out, &u, sizeof(out));
    }
};
// The f64 decoupled kernel's: walks the quantised records (rt_types.hpp Bvh4QNode) — four 16-byte reads of one line per visit.
template <uint32_t STRIDE, uint32_t ENTRIES> struct LdsStackQuant4 : LdsStack<STRIDE, ENTRIES> {
    static constexpr int WIDE = NODES_Q8X4;
    template <typename R> __device__ __forceinline__ void fetch4q(const SceneView<R>& sc, int32_t i, uint32_t* w) const {
        const int4* rec = reinterpret_cast<const int4*>(sc.nodes4q + i);
        int4 q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = rec[k];
        __builtin_memcpy(w, q, 64);
    }
};
// Same, with the whole node array resident in LDS in PIECE-MAJOR order: the q-th 16 bytes of node i at
// piece[q*n_nodes + i] (q < 7: the pad is left out).  64 lanes fetching the same piece of 64 unrelated nodes then spread
// over all the 16-byte bank slots (i mod 16); in node-major order the 128-byte records would all start at the same two
// — measured on the 64-byte binary records: 31 % of the LDS cycles were bank conflicts that way.
template <uint32_t STRIDE> struct LdsStackNodes : LdsStack<STRIDE> {
    static constexpr int SLAB_F32 = SLAB_FMA_FOLDED; // the issue-bound form of the lane-owns-path kernel: the cheapest box test
    const int4* piece; // LDS
    uint32_t n_nodes;
    __device__ __forceinline__ uint32_t plane_off(uint32_t q) const { return q * n_nodes; }
    template <typename R> __device__ __forceinline__ void fetch(const SceneView<R>&, int32_t i, const uint32_t* near_off, Planes4& out) const {
        union { int4 q[7]; struct { float nr[3][4], fr[3][4]; int32_t child[4]; } p; } u;
#pragma unroll
        for (uint32_t a = 0; a < 3; ++a) {
            u.q[a] = piece[near_off[a] + uint32_t(i)];
            u.q[3 + a] = piece[(2u * a + 3u) * n_nodes - near_off[a] + uint32_t(i)];
        }
        u.q[6] = piece[6u * n_nodes + uint32_t(i)];
        __builtin_memcpy(&out, &u, sizeof(out));
    }
};

template <bool COUNT, int SHAPES> struct CounterSel { using type = NoCountersT<SHAPES>; }; // SHAPES: rt_core.hpp SHAPES_FAST / SHAPES_GENERAL / SHAPES_NONE
template <int SHAPES> struct CounterSel<true, SHAPES> { using type = LaneCountersT<SHAPES>; };

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Hands a distinct job index to every lane of `mask` (wave-uniform).  Jobs come from the batch [next, end) the wave
// has reserved; when it runs short the wave leader reserves JOB_BATCH more with ONE atomic on the global counter.
// (One atomic per refill event saturated the single counter address at ~10^8 small jobs per second.)
constexpr unsigned long long JOB_BATCH = 256;
__device__ __forceinline__ unsigned long long wave_take_jobs(unsigned long long mask, uint32_t lane, unsigned long long& next,
                                                              unsigned long long& end, unsigned long long* __restrict__ job_counter) {
    const uint32_t want = uint32_t(__popcll(mask)), rank = uint32_t(__popcll(mask & ((1ull << lane) - 1ull)));
    const unsigned long long avail = end - next;
    if (avail >= want) {
        const unsigned long long job = next + rank;
        next += want;
        return job;
    }
    const int leader = __ffsll((long long)mask) - 1;
    unsigned long long base = 0;
    if (int(lane) == leader) base = atomicAdd(job_counter, JOB_BATCH);
    const uint32_t blo = __shfl(uint32_t(base), leader, 64), bhi = __shfl(uint32_t(base >> 32), leader, 64);
    base = (unsigned long long)blo | ((unsigned long long)bhi << 32);
    const unsigned long long job = rank < avail ? next + rank : base + (rank - avail);
    next = base + (want - avail);
    end = base + JOB_BATCH;
    return job;
}

#ifndef RT_POOL_NT
#define RT_POOL_NT 0
#endif
template <typename T> __device__ __forceinline__ T pool_ld(const T* p) {
#if RT_POOL_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <typename T, typename V> __device__ __forceinline__ void pool_st(T* p, V v) {
#if RT_POOL_NT
    __builtin_nontemporal_store(T(v), p);
#else
    *p = T(v);
#endif
}
constexpr int TRACE_BLOCK = 256;
#ifndef RT_SLOTS
#define RT_SLOTS 128
#endif
constexpr uint32_t SLOTS_PER_WAVE = RT_SLOTS; // paths owned by one wave64: 64 being traversed + up to 64 queued
constexpr uint32_t QCAP = RT_SLOTS;           // capacity of a wave's ray queue and hit queue (entries)
static_assert(QCAP == SLOTS_PER_WAVE, "a slot has at most one ray or hit in flight: the queues never hold more entries than the wave has slots");

// Path state of a slot, in global memory (L2-resident), struct-of-arrays over all slots of the launch.
// (no radiance: a path's value is the term of its last bounce, rt_core.hpp path_shade — round 4; 13 reals a slot instead of 16)
enum : uint32_t { PR_TX = 0, PR_TY, PR_TZ, PR_AX, PR_AY, PR_AZ, PR_COUNT }; // (the slot's RAY stays in LDS from its emit to its shade: the wave's ray arena, below)
enum : uint32_t { PU_KEY_LO = 0, PU_KEY_HI, PU_BOUNCE, PU_PXROW, PU_S, PU_SEND, PU_JOB_LO, PU_JOB_HI, PU_COUNT };
// bytes of LDS one wave needs: ray queue (7 reals + slot), hit queue (t + prim + inst + meta), traversal stacks
// LDS stack entries of the decoupled kernel: 16 for f32; 13 for f64, whose queues are twice as wide: THREE 256-thread blocks must fit a CU's 160 KB,
// or the kernel runs at 2 waves per SIMD however few registers it is held to.  gfx950 hands LDS out in granules of 1280 BYTES (128 to a CU): a block
// of 42 granules (53 760 B) fits three times, one of 54 016 B does not — measured in round 5 (profiles/r05/README.md: 445 against 344 Msamples/s
// on spheres_1m; hipOccupancyMaxActiveBlocksPerMultiprocessor says 3 for both).  Rounds 1-4 asked for 54 272 B with 12 entries — the runtime's
// answer was 3, SQ_WAVE_CYCLES said 2 of 3 waves were ever resident — so the f64 decoupled kernels ran a third short of their waves: the ray slot
// queue as bytes (slots are < 128) and 13 entries make it 53 760 B exactly (the spill strip in global memory takes the rare deeper entries).
#ifndef RT_F64_WAVE_STACK
#define RT_F64_WAVE_STACK 13
#endif
#ifndef RT_WAVE_LDS_PAD
#define RT_WAVE_LDS_PAD 0 // experiments: unused bytes per wave (where the LDS stops holding three blocks per CU)
#endif
template <typename R> __host__ __device__ constexpr uint32_t wave_stack_entries() { return sizeof(R) == 8 ? uint32_t(RT_F64_WAVE_STACK) : LDS_STACK_ENTRIES; }
template <typename R> __host__ __device__ constexpr uint32_t wave_lds_bytes(uint32_t stack_depth, bool no_time = false) {
    // ray queue (7 reals) + hit t | hit prim, inst, meta (words) | ray slot (bytes) | stack: + the spare slot
    return (no_time ? 7u : 8u) * QCAP * uint32_t(sizeof(R)) + 3u * QCAP * 4u + QCAP + (wave_stack_entries<R>() + 1u) * 64u * 4u + RT_WAVE_LDS_PAD;
}
constexpr uint32_t LDS_GRANULE_BYTES = 1280u, LDS_BYTES_PER_CU = 160u * 1024u; // gfx950: 128 granules per CU
__host__ __device__ constexpr uint32_t lds_blocks_per_cu(uint32_t block_bytes) {
    return block_bytes == 0u ? 1024u : LDS_BYTES_PER_CU / ((block_bytes + LDS_GRANULE_BYTES - 1u) / LDS_GRANULE_BYTES * LDS_GRANULE_BYTES);
}
// waves of ONE block that fills a CU (the LEAN flavour of the decoupled kernel): as many as the CU's LDS granules hold, at most 16 (4 per SIMD)
__host__ __device__ constexpr uint32_t wave_block_waves(uint32_t wave_bytes) {
    uint32_t n = 16u;
    while (n > 4u && (n * wave_bytes + LDS_GRANULE_BYTES - 1u) / LDS_GRANULE_BYTES > LDS_BYTES_PER_CU / LDS_GRANULE_BYTES) --n;
    return n;
}
static_assert(RT_WAVE_LDS_PAD != 0 || lds_blocks_per_cu(wave_lds_bytes<double>(0) * 4u) >= 3u, "the f64 decoupled kernel's block must fit a CU's LDS three times");
static_assert(lds_blocks_per_cu(wave_lds_bytes<float>(0) * 4u) >= 3u, "the f32 decoupled kernel's block must fit a CU's LDS three times");
template <typename R> __host__ __device__ constexpr bool wave_walks_quantised() { return RT_WAVE_QUANT == 2 || (RT_WAVE_QUANT == 1 && sizeof(R) == 8); }
constexpr uint32_t HIT_FRESH = 0x100u; // hit-queue meta: slot (8 bits) | FRESH | box face << 9
static_assert(SLOTS_PER_WAVE <= 256u, "slot numbers travel as bytes");

// The per-pixel sample loop of main.rs:202-229 as ONE persistent kernel in which PATHS ARE DECOUPLED FROM LANES.
//
// A wave64 owns 128 path slots whose state (ray, throughput, radiance, RNG key, pixel/sample bookkeeping) lives in
// global memory; a lane only ever holds a RAY BEING TRAVERSED (origin, direction, closest hit, BVH cursor), so the
// traversal loop is tight and nothing else is loop-carried.  Two wave-private LDS queues connect the two halves:
//   * TRAVERSE iteration: lanes without a ray pop one from the ray queue (ranks by __ballot/popcount — the queues are
//     private to the wave, no atomics), every lane advances its ray by one walk trip (a few node steps and a leaf
//     step), lanes whose ray is finished push (slot, t, primitive) onto the hit queue and are free for the next ray:
//     no lane waits for the longest traversal in the wave.
//   * SHADE: as soon as 64 hits are queued the whole wave processes them at full occupancy — media, hit record,
//     emitted + scatter (rt_core.hpp path_shade) — and pushes the 64 continuation rays.  A path that ended adds its
//     radiance to its job's sequential sum (main.rs:211-216) and starts the job's next sample; a slot whose job is
//     finished writes the job's partial sum and takes the next job ((pixel, sample chunk), see job_decode) from the
//     wave's batch of job indices (one atomic on the global counter per 256 jobs).
// Results do not depend on any of this scheduling: every draw is keyed by (pixel, sample, bounce), every job is a
// sequential fold, and the resolve kernel adds a pixel's jobs in chunk order.
// Re-read a by-value kernel argument from the kernarg segment at its (cold) point of use, so that it does not hold
// SGPRs for the whole kernel: the trace kernels are at the 102-SGPR limit and spill to VGPR lanes otherwise.  The
// pointer is passed through an empty asm so that the loads stay where they are written.
template <typename T> __device__ __forceinline__ T kernarg_reload(uint32_t offset) {
    typedef const char __attribute__((address_space(4)))* KPtr;
    KPtr p = (KPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    T v;
    __builtin_memcpy(&v, p + offset, sizeof(T));
    return v;
}
// Layout of the first three kernel arguments: the kernarg segment places by-value arguments in order at their natural
// alignment, which is what this struct does with its members (the GPU parity tests would not survive a mismatch).
template <typename R> struct TraceArgsHead { SceneView<R> sc; CameraRec<R> cam; RenderConsts rc; };
static_assert(alignof(SceneView<float>) <= 8 && alignof(CameraRec<double>) <= 8 && alignof(RenderConsts) <= 8, "kernarg_reload assumes naturally aligned arguments");

template <typename R, bool COUNT, int GENERAL> // GENERAL: SHAPES_FAST (0) / SHAPES_GENERAL (1) / SHAPES_NONE (2: the scene has no instance record, rt_core.hpp)
// (at least 3 waves/SIMD: 170 VGPRs — the f32 code needs 164; the f64 code, allowed 256, ran at 2 waves/SIMD and waited on
// the fabric: spheres_1m f64 167 -> 264 Msamples/s with 140 registers spilled; 4 waves/SIMD: 205)
// (the LEAN flavour — 127-129 registers in f64, ~110 in f32 — may be launched as ONE block per CU of as many waves as the CU's LDS holds, wave_block_waves():
// its waves never synchronise with each other, so a block is only a unit of LDS allocation, and 13 f64 waves fit where three 4-wave blocks hold 12)
__global__ __launch_bounds__(GENERAL == SHAPES_NONE_NT ? 1024 : TRACE_BLOCK, GENERAL == SHAPES_NONE_NT ? 1 : 3) void trace_kernel(SceneView<R> sc, CameraRec<R> cam, RenderConsts rc, R bg_r, R bg_g,
                                                            R bg_b, R t_min, R* __restrict__ partial,
                                                            unsigned long long* __restrict__ job_counter,
                                                            DeviceCounters* __restrict__ counters, R* __restrict__ pool_r,
                                                            uint32_t* __restrict__ pool_u, uint32_t n_slots, int32_t* __restrict__ spill) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    typename CounterSel<COUNT, GENERAL>::type cnt;
    const uint32_t lane = threadIdx.x & 63u, wave_in_block = threadIdx.x >> 6;
    unsigned char* wbase = lds_raw + wave_in_block * wave_lds_bytes<R>(rc.stack_depth, decltype(cnt)::NO_TIME);
    // RAY ARENA [7][SLOTS_PER_WAVE]: o.xyz, d.xyz, time of every slot's current ray, INDEXED BY SLOT — written when the ray is emitted, read by the lane
    // that walks it and again by the shade of its hit; the queues carry slot numbers only.  (Rounds 2-4 queued the ray by queue position and kept a copy
    // in the slot's global state for the shade to read back: 2 x 7 of a slot's 13 reals per bounce, and this kernel waits on L2-miss lines.)
    R* const ra = reinterpret_cast<R*>(wbase);
    R* const hq_t = ra + (decltype(cnt)::NO_TIME ? 6u : 7u) * SLOTS_PER_WAVE; // hit queue: t
    int32_t* const hq_prim = reinterpret_cast<int32_t*>(hq_t + QCAP);
    int32_t* const hq_inst = hq_prim + QCAP;
    uint32_t* const hq_meta = reinterpret_cast<uint32_t*>(hq_inst + QCAP);
    uint8_t* const rq_slot = reinterpret_cast<uint8_t*>(hq_meta + QCAP); // (slots are 0 .. 127: a byte each)
    typename std::conditional<wave_walks_quantised<R>(), LdsStackQuant4<64, wave_stack_entries<R>()>, LdsStack<64, wave_stack_entries<R>()>>::type stack;
    stack.base = (LdsIntPtr)(reinterpret_cast<int32_t*>(rq_slot + QCAP) + lane);
    stack.spill = (GlobalIntPtr)(spill + (blockIdx.x * blockDim.x + threadIdx.x));
    stack.spill_stride = gridDim.x * blockDim.x;

    const uint32_t wave_global = blockIdx.x * (blockDim.x / 64u) + wave_in_block;
    const size_t gbase = size_t(wave_global) * SLOTS_PER_WAVE;
    const unsigned long long n_jobs = rc.n_jobs;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    const V3<R> background(bg_r, bg_g, bg_b);

    // every slot starts out needing its first job
    hq_meta[lane] = lane | HIT_FRESH;
    for (uint32_t i = lane + 64u; i < SLOTS_PER_WAVE; i += 64u) hq_meta[i] = i | HIT_FRESH;
    uint32_t ray_n = 0, hit_n = SLOTS_PER_WAVE; // wave-uniform queue fill levels
    unsigned long long batch_next = 0, batch_end = 0; // the wave's reserved batch of job indices

    uint32_t dbg[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    bool has_ray = false;
    uint32_t slot = 0;
    Ray<R> wray; // the ray this lane is traversing, in world space
    Trav<R> tr;

    for (;;) {
        __builtin_amdgcn_wave_barrier();
        const bool any_ray = __ballot(has_ray) != 0ull;
        if (hit_n >= 64u || (!any_ray && ray_n == 0u)) {
            if (hit_n == 0u) break; // nothing traversing, nothing queued: this wave is done
            // ================================================================== SHADE (up to 64 queued hits)
            const uint32_t m = hit_n < 64u ? hit_n : 64u;
            if constexpr (COUNT) { dbg[9] += 1; dbg[10] += m; }
            const bool on = lane < m;
            const uint32_t e = hit_n - 1u - (on ? lane : 0u);
            hit_n -= m;
            const uint32_t meta = on ? hq_meta[e] : HIT_FRESH;
            const uint32_t hslot = meta & 0xFFu;
            const bool fresh = (meta & HIT_FRESH) != 0u;
            const size_t g = gbase + hslot;
            R* const pr = pool_r + g;
            uint32_t* const pu = pool_u + g;

            PathState<R> ps;
            bool emit = false, need_sample = false, slot_done = false;
            uint32_t pxrow = 0, smp = 0, smp_end = 0;
            unsigned long long job = ~0ull;
            V3<R> acc;
            if (on && !fresh) {
                ps.ray.o = V3<R>(ra[0u * SLOTS_PER_WAVE + hslot], ra[1u * SLOTS_PER_WAVE + hslot], ra[2u * SLOTS_PER_WAVE + hslot]);
                ps.ray.d = V3<R>(ra[3u * SLOTS_PER_WAVE + hslot], ra[4u * SLOTS_PER_WAVE + hslot], ra[5u * SLOTS_PER_WAVE + hslot]);
                if constexpr (!decltype(cnt)::NO_TIME) ps.ray.time = ra[6u * SLOTS_PER_WAVE + hslot]; // (a scene in which nothing reads the time does not carry it)
                else ps.ray.time = R(0);
                ps.throughput = V3<R>(pool_ld(pr + size_t(PR_TX) * n_slots), pool_ld(pr + size_t(PR_TY) * n_slots), pool_ld(pr + size_t(PR_TZ) * n_slots));
                ps.key = (unsigned long long)pool_ld(pu + size_t(PU_KEY_LO) * n_slots) | ((unsigned long long)pool_ld(pu + size_t(PU_KEY_HI) * n_slots) << 32);
                ps.bounce = pool_ld(pu + size_t(PU_BOUNCE) * n_slots);
                HitRef best;
                best.prim = hq_prim[e];
                best.inst = hq_inst[e];
                best.aux = int32_t((meta >> 9) & 7u);
                const bool found = ref_kind(best.prim) != PRIM_NONE;
                // (scene view and constants re-read from the kernarg segment: the traversal loop keeps only the pointers it uses)
                if (path_shade(ps, kernarg_reload<SceneView<R>>(offsetof(TraceArgsHead<R>, sc)), kernarg_reload<RenderConsts>(offsetof(TraceArgsHead<R>, rc)), background, t_min,
                               found, hq_t[e], best, cnt)) {
                    emit = true; // next world.hit of the same path
                } else {         // main.rs:216: acc + color(...)
                    pxrow = pool_ld(pu + size_t(PU_PXROW) * n_slots);
                    smp = pool_ld(pu + size_t(PU_S) * n_slots);
                    smp_end = pool_ld(pu + size_t(PU_SEND) * n_slots);
                    job = (unsigned long long)pool_ld(pu + size_t(PU_JOB_LO) * n_slots) | ((unsigned long long)pool_ld(pu + size_t(PU_JOB_HI) * n_slots) << 32);
                    acc = V3<R>(pool_ld(pr + size_t(PR_AX) * n_slots), pool_ld(pr + size_t(PR_AY) * n_slots), pool_ld(pr + size_t(PR_AZ) * n_slots)) + ps.radiance;
                    ++smp;
                    need_sample = true;
                }
            } else if (on) {
                need_sample = true; // fresh slot: smp == smp_end == 0, no job yet
            }
            // job hand-out for the slots whose job is finished: wave-aggregated, one atomic per round.  An empty job
            // (a tile pixel outside the image) is finished at once, hence the loop.
            for (;;) {
                const bool need_job = need_sample && !slot_done && smp >= smp_end;
                const unsigned long long jm = __ballot(need_job);
                if (jm == 0ull) break;
                if (need_job && job != ~0ull) { // retire the finished job: its sequential sum
                    R* dst = partial + job * 3ull;
                    dst[0] = acc.x; dst[1] = acc.y; dst[2] = acc.z;
                }
                const unsigned long long mine = wave_take_jobs(jm, lane, batch_next, batch_end, job_counter);
                if (need_job) {
                    job = mine;
                    if (job >= n_jobs) {
                        slot_done = true; // no jobs left: this slot retires
                    } else {
                        const RenderConsts rj = kernarg_reload<RenderConsts>(offsetof(TraceArgsHead<R>, rc)); // cold: keep it out of the SGPRs
                        const JobInfo ji = job_decode(rj, uint32_t(job));
                        pxrow = ji.px | (ji.row << 16);
                        smp = ji.s; smp_end = ji.s_end;
                        job = ji.real ? (unsigned long long)ji.sum_index : ~0ull; // from here on: where the job's sum goes (none for padding)
                        acc = V3<R>();
                    }
                }
            }
            if (need_sample && !slot_done) { // main.rs:212-215: the job's next sample
                path_begin(ps, kernarg_reload<CameraRec<R>>(offsetof(TraceArgsHead<R>, cam)), kernarg_reload<RenderConsts>(offsetof(TraceArgsHead<R>, rc)), pxrow & 0xFFFFu, pxrow >> 16, smp);
                pool_st(pu + size_t(PU_KEY_LO) * n_slots, uint32_t(ps.key));
                pool_st(pu + size_t(PU_KEY_HI) * n_slots, uint32_t(ps.key >> 32));
                pool_st(pu + size_t(PU_PXROW) * n_slots, pxrow);
                pool_st(pu + size_t(PU_S) * n_slots, smp);
                pool_st(pu + size_t(PU_SEND) * n_slots, smp_end);
                pool_st(pu + size_t(PU_JOB_LO) * n_slots, uint32_t(job));
                pool_st(pu + size_t(PU_JOB_HI) * n_slots, uint32_t(job >> 32));
                pool_st(pr + size_t(PR_AX) * n_slots, acc.x); pool_st(pr + size_t(PR_AY) * n_slots, acc.y); pool_st(pr + size_t(PR_AZ) * n_slots, acc.z);
                emit = true;
            }
            const unsigned long long em = __ballot(emit);
            if (emit) { // the slot's next ray: path state back to memory, ray onto the queue
                pool_st(pr + size_t(PR_TX) * n_slots, ps.throughput.x); pool_st(pr + size_t(PR_TY) * n_slots, ps.throughput.y); pool_st(pr + size_t(PR_TZ) * n_slots, ps.throughput.z);
                pool_st(pu + size_t(PU_BOUNCE) * n_slots, ps.bounce);
                const uint32_t idx = ray_n + uint32_t(__popcll(em & lanes_below));
                ra[0u * SLOTS_PER_WAVE + hslot] = ps.ray.o.x; ra[1u * SLOTS_PER_WAVE + hslot] = ps.ray.o.y; ra[2u * SLOTS_PER_WAVE + hslot] = ps.ray.o.z;
                ra[3u * SLOTS_PER_WAVE + hslot] = ps.ray.d.x; ra[4u * SLOTS_PER_WAVE + hslot] = ps.ray.d.y; ra[5u * SLOTS_PER_WAVE + hslot] = ps.ray.d.z;
                if constexpr (!decltype(cnt)::NO_TIME) ra[6u * SLOTS_PER_WAVE + hslot] = ps.ray.time;
                rq_slot[idx] = uint8_t(hslot);
            }
            ray_n += uint32_t(__popcll(em));
            continue;
        }

        // ====================================================================== TRAVERSE (one step for every ray)
        const unsigned long long nm = __ballot(!has_ray);
        if (nm != 0ull && ray_n != 0u) { // hand queued rays to the idle lanes
            const uint32_t want = uint32_t(__popcll(nm)), take = want < ray_n ? want : ray_n;
            if constexpr (COUNT) { dbg[11] += 1; dbg[12] += take; }
            const uint32_t rank = uint32_t(__popcll(nm & lanes_below));
            if (!has_ray && rank < take) {
                const uint32_t e = ray_n - 1u - rank;
                slot = rq_slot[e];
                wray.o = V3<R>(ra[0u * SLOTS_PER_WAVE + slot], ra[1u * SLOTS_PER_WAVE + slot], ra[2u * SLOTS_PER_WAVE + slot]);
                wray.d = V3<R>(ra[3u * SLOTS_PER_WAVE + slot], ra[4u * SLOTS_PER_WAVE + slot], ra[5u * SLOTS_PER_WAVE + slot]);
                if constexpr (!decltype(cnt)::NO_TIME) wray.time = ra[6u * SLOTS_PER_WAVE + slot];
                else wray.time = R(0);
                cnt.ray();
                trav_begin(tr, sc, wray, stack);
                has_ray = true;
            }
            ray_n -= take;
        }
        // A burst of walk trips (the loop body of closest_solid(), with more node steps per trip: these trees are deep): it
        // ends once enough lanes have finished their ray to make the hand-over below worth its cost.  (An earlier form
        // voted, per step, for ONE kind of step — inner node / sphere / box / other — to run for all lanes waiting on it;
        // the vote cost about as much as a node step, and plain trips beat it: spheres_1m 306 -> 330 Msamples/s,
        // final_scene through this kernel 916 -> 1084.  Node steps per trip 2 / 3 / 4 / 6: 302 / 321 / 325 / 330.)
        {
            const uint32_t retire_batch = ray_n != 0u ? uint32_t(RT_WAVE_RETIRE) : 64u;
            if constexpr (COUNT) dbg[8] += 1;
            for (;;) {
                const bool walking = has_ray && tr.node != TRAV_DONE;
                if (__ballot(walking) == 0ull) break; // every ray of the wave is finished
                if (walking) {
#pragma unroll
                    for (int k = 0; k < RT_WAVE_STEPS; ++k)
                        if (tr.node >= 0) trav_node_step(tr, sc, wray, t_min, stack, cnt);
                    if (tr.node < 0 && tr.node != TRAV_DONE) trav_leaf_step<RT_WAVE_WHOLE_LEAF != 0>(tr, sc, wray, t_min, stack, cnt);
                }
                if (uint32_t(__popcll(__ballot(has_ray && tr.node == TRAV_DONE))) >= retire_batch) break;
            }
        }
        const bool fin = has_ray && tr.node == TRAV_DONE;
        const unsigned long long fm = __ballot(fin);
        if (fm != 0ull) { // finished rays: hit onto the queue, lane free again
            if (fin) {
                const uint32_t idx = hit_n + uint32_t(__popcll(fm & lanes_below));
                hq_t[idx] = tr.closest;
                hq_prim[idx] = tr.found ? tr.best.prim : make_ref(PRIM_NONE, 0);
                hq_inst[idx] = tr.best.inst;
                hq_meta[idx] = slot | (uint32_t(tr.best.aux) << 9);
                has_ray = false;
            }
            hit_n += uint32_t(__popcll(fm));
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
            for (int k = 0; k < 16; ++k) atomicAdd(&counters->dbg[k], (unsigned long long)dbg[k]);
        }
    }
}

// The f64 translation unit that holds the launch code is built with the pre-RA machine scheduler off (Makefile: the lane-owns-path kernel's
// 128 registers), which costs THIS kernel 10 % (spheres_1m f64 280 -> 308 Msamples/s without the flag): its contracted f64 instantiations are
// compiled in a unit of their own (render_f64_wave.hip) and only declared where they are launched.
#if defined(RT_F64_WAVE_KERNELS_ELSEWHERE)
#define RT_WAVE_DECL(COUNT, GENERAL)                                                                                                                      \
    extern template __global__ void trace_kernel<double, COUNT, GENERAL>(SceneView<double>, CameraRec<double>, RenderConsts, double, double, double, double, \
                                                                         double*, unsigned long long*, DeviceCounters*, double*, uint32_t*, uint32_t, int32_t*);
RT_WAVE_DECL(false, SHAPES_FAST) RT_WAVE_DECL(false, SHAPES_GENERAL) RT_WAVE_DECL(false, SHAPES_NONE) RT_WAVE_DECL(false, SHAPES_NONE_NT) RT_WAVE_DECL(true, SHAPES_FAST) RT_WAVE_DECL(true, SHAPES_GENERAL)
#undef RT_WAVE_DECL
#endif

// The plain form of the same loop: a lane OWNS a path (and its job): path state stays in registers, no queues.  Rounds 1-4 alternated
// "regenerate or advance by one bounce" (rt_core.hpp path_step = whole BVH walk + shade) with the wave-aggregated job fetch, and every lane
// waited for the longest BVH walk of the wave at every bounce.  Since round 5 (RT_ASYNC_SHADE) the walk loop is left for a SHADE PHASE once
// at most RT_ASYNC_SLACK walks are unfinished: the finished lanes shade, regenerate and start their next walk, the unfinished ones keep
// their cursor and walk on beside them.  Kept beside the decoupled kernel because which of the two is faster depends on the scene
// (render_tiles.hpp: the crossover is at ~13 000 four-wide records; KERNELS.md).
// NSTEPS: node steps per trip round the walk loop (rt_core.hpp closest_solid): RT_NODE_STEPS, or 3 for tiny top trees (render_tiles.hpp).
template <typename R, bool COUNT, int BLOCK, bool LDSN, int GENERAL, int NSTEPS = RT_NODE_STEPS> // GENERAL: SHAPES_FAST / SHAPES_GENERAL / SHAPES_NONE, as above
// (the 256-thread form — nodes in global memory — asks for at least 3 waves/SIMD like the decoupled kernel: its f64 code,
// allowed 256 VGPRs, ran at 2: a 20 000-sphere scene 29.9 -> 13.7 ms per 67 Msamples)
__global__ __launch_bounds__(BLOCK, BLOCK == 256 ? 3 : 1) void trace_kernel_plain(SceneView<R> sc_arg, CameraRec<R> cam, RenderConsts rc, R bg_r, R bg_g,
                                                            R bg_b, R t_min, R* __restrict__ partial,
                                                            unsigned long long* __restrict__ job_counter,
                                                            DeviceCounters* __restrict__ counters, int32_t* __restrict__ spill) {
    // LDSN: the whole node array is copied into LDS (piece-major, see LdsStackNodes) in front of the stacks — small
    // scenes: one dependent ~100-cycle LDS read per node visit instead of an L1/L2 round trip
    extern __shared__ __align__(16) int32_t lds_stack[];
    typename std::conditional<LDSN, LdsStackNodes<BLOCK>, LdsStack<BLOCK>>::type stack;
    stack.spill = (GlobalIntPtr)(spill + (blockIdx.x * blockDim.x + threadIdx.x));
    stack.spill_stride = gridDim.x * blockDim.x;
    SceneView<R> sc = sc_arg; // (the LDS form redirects the Perlin tables below)
    if constexpr (LDSN) {
        const uint32_t n = rc.lds_nodes & LDS_NODES_MASK, n_perlin = rc.lds_nodes >> LDS_PERLIN_SHIFT;
        if (n_perlin) {
            // The Perlin tables (noise.rs:40-47: 256 vectors + three permutations, 6.75 KB in f64) behind the stacks: a turbulence
            // value is 7 octaves x 8 corners of dependent table reads, made for the one or two lanes of a wave that hit a noise
            // texture while the others wait — from LDS instead of L2: final_scene f64 1284 -> 1315 Msamples/s, f32 1756 -> 1774.
            R* pv = reinterpret_cast<R*>(lds_stack + n * (4u * BVH4_USED_SIXTEENTHS) + (LDS_STACK_ENTRIES + 1u) * BLOCK);
            uint32_t* pp = reinterpret_cast<uint32_t*>(pv + 768u * n_perlin);
            for (uint32_t i = threadIdx.x; i < 768u * n_perlin; i += blockDim.x) pv[i] = sc_arg.perlin_vec[i];
            for (uint32_t i = threadIdx.x; i < 192u * n_perlin; i += blockDim.x) pp[i] = reinterpret_cast<const uint32_t*>(sc_arg.perlin_perm)[i];
            sc.perlin_vec = pv;
            sc.perlin_perm = reinterpret_cast<const uint8_t*>(pp);
        }
        {
            // The record arrays a LEAF step reads, where they are small enough to follow (the host decides, render_tiles.hpp):
            // transform chains, rectangles, moving spheres, cubes.  An instance leaf is two dependent record reads (the chain, then
            // the wrapped record) in the middle of the walk loop: cornell_box f64 1557 -> 1617 Msamples/s with all of them in LDS.
            uint32_t* at = reinterpret_cast<uint32_t*>(lds_stack + n * (4u * BVH4_USED_SIXTEENTHS) + (LDS_STACK_ENTRIES + 1u) * BLOCK) +
                           uint32_t(lds_pad32(lds_perlin_bytes(n_perlin, sizeof(R)))) / 4u;
            auto stage = [&](const void* src, uint32_t bytes) -> const void* {
                uint32_t* dst = at;
                const uint32_t words = uint32_t(lds_pad32(bytes)) / 4u;
                for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) dst[i] = reinterpret_cast<const uint32_t*>(src)[i];
                at += words;
                return dst;
            };
            if (rc.lds_recs[0]) sc.insts = static_cast<const InstanceRec<R>*>(stage(sc_arg.insts, rc.lds_recs[0] * uint32_t(sizeof(InstanceRec<R>))));
            if (rc.lds_recs[1]) sc.rects = static_cast<const RectRec<R>*>(stage(sc_arg.rects, rc.lds_recs[1] * uint32_t(sizeof(RectRec<R>))));
            if (rc.lds_recs[2]) sc.moving = static_cast<const MovingSphereRec<R>*>(stage(sc_arg.moving, rc.lds_recs[2] * uint32_t(sizeof(MovingSphereRec<R>))));
            if (rc.lds_recs[3]) sc.boxes = static_cast<const BoxRec<R>*>(stage(sc_arg.boxes, rc.lds_recs[3] * uint32_t(sizeof(BoxRec<R>))));
            // (and the spheres' material slots: the first link of the chain material -> texture -> table at every sphere hit: +0.4 %)
            if (rc.lds_recs[4]) sc.sphere_mat = static_cast<const int32_t*>(stage(sc_arg.sphere_mat, rc.lds_recs[4] * 4u));
        }
        const int4* src = reinterpret_cast<const int4*>(sc.nodes);
        int4* dst = reinterpret_cast<int4*>(lds_stack);
        for (uint32_t i = threadIdx.x; i < n * 8u; i += blockDim.x)
            if ((i & 7u) < BVH4_USED_SIXTEENTHS) dst[(i & 7u) * n + (i >> 3)] = src[i];
        __syncthreads();
        stack.base = (LdsIntPtr)(lds_stack + n * (4u * BVH4_USED_SIXTEENTHS) + threadIdx.x);
        stack.piece = dst;
        stack.n_nodes = n;
    } else {
        stack.base = (LdsIntPtr)(lds_stack + threadIdx.x);
    }
    typename CounterSel<COUNT, GENERAL>::type cnt;

    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long n_jobs = rc.n_jobs;
    const V3<R> background(bg_r, bg_g, bg_b);

    bool has_job = false, alive = false, done = false;
    unsigned long long batch_next = 0, batch_end = 0; // the wave's reserved batch of job indices
    uint32_t job = 0;    // where the lane's job writes its sum (JobInfo::sum_index: below 2^32, plan_jobs)
    uint32_t pxrow = 0;  // the job's pixel, px | row << 16 (the image is at most 65535 wide and high: render_api.cpp validate) — one register, not two:
                         // the f64 instantiations run at their register cap and every spilled one is a scratch access per phase (cornell_box f64: three
                         // registers fewer = +6 %, profiles/r05/README.md)
    uint32_t s = 0, s_end = 0;
    V3<R> acc;
    PathState<R> ps;
    bool walking = false; // RT_ASYNC_SHADE: the lane has a walk in progress (begun, or suspended by a shade phase)
    Trav<R> tr;
    uint32_t tally_trips = 0; // (counting variant: trips of the lane's walk so far)

    // counting variant only: where a wave's time and lanes go (RTTNW_DEBUG_SCHED prints it) — wave clock per phase
    // [0..3], lockstep iterations of the BVH walk [4] (with a node lane [7], with a leaf lane [8]) against the lane
    // steps they served [5] node / [6] leaf, bounce rounds [9] and the lanes alive in them [10], regenerations [11,12]
    unsigned long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (;;) {
        long long tk0 = 0;
        if constexpr (COUNT) tk0 = clock64();
        // ---- job hand-out: wave-aggregated, one atomic per refill event
        const bool need = !done && !alive && s >= s_end;
        const unsigned long long mask = __ballot(need);
        if (mask != 0ull) {
            if (need && has_job) { // retire the finished job: its sequential sum
                R* dst = partial + (unsigned long long)job * 3ull;
                dst[0] = acc.x; dst[1] = acc.y; dst[2] = acc.z;
                has_job = false;
            }
            const unsigned long long mine = wave_take_jobs(mask, lane, batch_next, batch_end, job_counter);
            if (need) {
                if (mine >= n_jobs) {
                    done = true;
                } else {
                    const JobInfo ji = job_decode(rc, uint32_t(mine));
                    pxrow = ji.px | (ji.row << 16); s = ji.s; s_end = ji.s_end;
                    job = ji.sum_index; // from here on: where the job's sum goes
                    acc = V3<R>();
                    has_job = ji.real; // padding jobs have no sum to write
                }
            }
        }
        if (__ballot(!done) == 0ull) break;

        // ---- one path per lane: regenerate or advance by one bounce
        if constexpr (!COUNT && RT_ASYNC_SHADE != 0) {
            // ASYNCHRONOUS SHADE PHASES (round 5).  Rounds 1-4 ran `path_step` here: every lane's whole walk, then the shade — a round lasted as long as
            // its LONGEST walk (final_scene: 9.5 trips where the mean walk has 4.2; a node step served 20 of 64 lanes).  Now the walk loop is left once
            // at most RT_ASYNC_SLACK walks are unfinished: the finished lanes shade, regenerate and start their next walk, the unfinished ones keep
            // their cursor (closest hit, node, stack) and walk on beside them.  Host model of the wave (tests/hostsim policy 4, profiles/experiments/wave_async.py):
            // node-step executions -31 %, leaf-step executions -17 %, shade phases +9 %.  A lane's own sequence of steps is untouched: same image.
            const bool fresh_path = !done && !walking && !alive && s < s_end;
            if (fresh_path) {
                path_begin(ps, kernarg_reload<CameraRec<R>>(offsetof(TraceArgsHead<R>, cam)), rc, pxrow & 0xFFFFu, pxrow >> 16, s);
                alive = true;
            }
            const bool fresh_walk = !done && !walking && alive;
            if (fresh_walk) { cnt.ray(); trav_init(tr, sc); walking = true; }
            // (the slab constants of a suspended walk are made again, in the instruction stream the fresh walks need anyway — for EVERY lane, so that
            // the compiler sees them dead across the shade phase: nothing but the cursor lives through it.  A walk suspended inside an instance's
            // tree keeps its own ray, and its constants with it.)
            if constexpr (CounterSel<COUNT, GENERAL>::type::NO_INST) trav_set_ray(tr, ps.ray, stack);
            else if (fresh_walk) trav_set_ray(tr, ps.ray, stack);
            if (fresh_walk) trav_reject_unwalkable(tr, ps.ray);
            for (;;) { // trips (rt_core.hpp closest_solid's loop body) until few enough walks are unfinished
                const bool unfinished = walking && tr.node != TRAV_DONE;
                if (unfinished) {
#pragma unroll
                    for (int k = 0; k < NSTEPS; ++k)
                        if (tr.node >= 0) trav_node_step(tr, sc, ps.ray, t_min, stack, cnt);
                }
                // (serving the lanes at a leaf by CLASS of record kind — the cube's ~200 instructions run for 2 lanes of 64 on final_scene — only once enough
                // lanes wait at a class was modelled at -8.8 % instructions and measured at -1.4 % / -4.5 % (f64 / f32): profiles/r05/README.md)
                if (unfinished && tr.node < 0 && tr.node != TRAV_DONE) trav_leaf_step<RT_ASYNC_WHOLE_LEAF != 0>(tr, sc, ps.ray, t_min, stack, cnt);
                const unsigned long long um = __ballot(walking && tr.node != TRAV_DONE);
                if (um == 0ull) break;
                if (uint32_t(__popcll(um)) <= uint32_t(RT_ASYNC_SLACK) && __ballot(walking && tr.node == TRAV_DONE) != 0ull) break;
            }
            if (walking && tr.node == TRAV_DONE) {
                walking = false;
                alive = path_shade(ps, sc, rc, background, t_min, tr.found, tr.closest, tr.best, cnt);
                if (!alive) { // main.rs:216: acc + color(...)
                    acc = acc + ps.radiance;
                    ++s;
                }
            }
        } else if constexpr (!COUNT) {
            if (!done) {
                if (!alive && s < s_end) {
                    path_begin(ps, kernarg_reload<CameraRec<R>>(offsetof(TraceArgsHead<R>, cam)), rc, pxrow & 0xFFFFu, pxrow >> 16, s);
                    alive = true;
                }
                if (alive) {
                    alive = path_step<NSTEPS>(ps, sc, rc, background, t_min, stack, cnt);
                    if (!alive) { // main.rs:216: acc + color(...)
                        acc = acc + ps.radiance;
                        ++s;
                    }
                }
            }
        } else { // the same steps, with the wave clock read between the phases and the lockstep loop tallied (trace_tally.hpp)
#if RT_ASYNC_SHADE
            plain_phase_tallied<NSTEPS>(done, alive, walking, tr, tally_trips, pxrow & 0xFFFFu, pxrow >> 16, s, s_end, acc, ps, cam, rc, sc, background, t_min, stack, cnt, prof, counters, lane, tk0);
#else
            plain_round_tallied(done, alive, pxrow & 0xFFFFu, pxrow >> 16, s, s_end, acc, ps, cam, rc, sc, background, t_min, stack, cnt, prof, counters, lane, tk0);
#endif
        }
    }

    if constexpr (COUNT) {
        uint32_t r = wave_sum(cnt.rays), n = wave_sum(cnt.nodes), p = wave_sum(cnt.prims), t = wave_sum(cnt.texels);
        if (lane == 0) {
            atomicAdd(&counters->rays, (unsigned long long)r);
            atomicAdd(&counters->nodes, (unsigned long long)n);
            atomicAdd(&counters->prims, (unsigned long long)p);
            atomicAdd(&counters->texels, (unsigned long long)t);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (prof[k]) atomicAdd(&counters->dbg[k], prof[k]);
    }
}

// Add a launch's chunk sums, in chunk order, onto the pixel's running sum — ONE chain per pixel over the whole render,
// sum = (((c0 + c1) + c2) + ...), continued from launch to launch (the first launch starts it), so the image does not
// depend on how the render was split into launches; the last launch divides by spp (main.rs:217): packed pixel records
// (r, g, b, 1).  Pad tiles (>= my_tiles) are zero-filled.  The running sum lives in `packed` itself.
template <typename R>
__global__ void resolve_kernel(const R* __restrict__ partial, R* __restrict__ packed, RenderConsts rc, uint32_t pixels_per_rank,
                               uint32_t first_launch, uint32_t last_launch, uint32_t total_spp) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pixels_per_rank) return;
    R* dst = packed + (unsigned long long)p * 4ull;
    R r = 0, g = 0, b = 0, a = 0;
    const unsigned long long jobs_per_chunk = (unsigned long long)rc.my_tiles * 64ull;
    if (p < jobs_per_chunk) {
        if (!first_launch) { r = dst[0]; g = dst[1]; b = dst[2]; }
        for (uint32_t c = 0; c < rc.n_chunks; ++c) {
            const R* src = partial + ((unsigned long long)c * jobs_per_chunk + p) * 3ull;
            r = r + src[0]; g = g + src[1]; b = b + src[2];
        }
        if (last_launch) {
            const R spp = R(total_spp);
            r = r / spp; g = g / spp; b = b / spp;
            a = R(1);
        }
    }
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
                                  uint32_t sample, double* __restrict__ out, uint32_t max_out, int32_t* __restrict__ n_out,
                                  int32_t* __restrict__ spill) {
    extern __shared__ int32_t lds_stack[];
    if (threadIdx.x != 0) return;
    LdsStack<64> stack{(LdsIntPtr)lds_stack, (GlobalIntPtr)spill, 1u};
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

} // namespace RT_ARITH_NS
} // namespace rt
