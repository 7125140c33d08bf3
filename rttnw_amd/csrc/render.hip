// render.hip — the device half of include/rttnw_hip.h for gfx950 (MI355X).
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
// Two forms of the loop: trace_kernel_plain (a lane owns a path) and trace_kernel (paths decoupled
// from lanes through wave-private queues, for trees that live in HBM).
// No CPU fallback: every entry point needs a HIP device.
#include "../../include/rttnw_hip.h"
#include "rt_core.hpp"
#include "scene_handle.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#ifndef RT_F32_STREAM_BLOCK
#define RT_F32_STREAM_BLOCK 1024
#endif
#ifndef RT_F64_BLOCK
#define RT_F64_BLOCK 768 // threads per block of the LDS-resident f64 kernel: 3 waves/SIMD at 168 VGPRs (512 / 768 / 1024: 718 / 885 / 874 Msamples/s on final_scene, 1004 / 1344 / 1166 on cornell_box; the spills at 768 are kernel-invariant values reloaded in shade)
#endif

namespace rt {

// ---------------------------------------------------------------------------------------------
// device-side helpers
// ---------------------------------------------------------------------------------------------
// Traversal memory of a lane: its BVH stack — the first LDS_STACK_ENTRIES entries in LDS (entry e of lane l at
// e*stride + l: conflict-free b32 accesses), deeper ones in a strip of global memory (entry e of thread g at
// e*spill_stride + g; a 4-wide walk can have three pending children per level but rarely has more than a dozen) — and
// the way it reads node records.
typedef __attribute__((address_space(3))) int32_t* LdsIntPtr;    // explicit address spaces: the compiler otherwise merges
typedef __attribute__((address_space(1))) int32_t* GlobalIntPtr; // the two halves of get() into one FLAT load
template <uint32_t STRIDE> struct LdsStack { // STRIDE = lanes sharing the LDS stack area: entry e of a lane at base[e * STRIDE]
    static constexpr int SPARE = int(LDS_STACK_ENTRIES); // a lane's extra LDS slot: target of the node step's masked-off stores
    LdsIntPtr base;        // &lds[threadIdx.x]
    GlobalIntPtr spill;    // &spill_buffer[global thread]
    uint32_t spill_stride; // threads of the launch
    __device__ __forceinline__ void set(int i, int32_t v) {
        if (uint32_t(i) < LDS_STACK_ENTRIES) base[uint32_t(i) * STRIDE] = v;
        else spill[size_t(uint32_t(i) - LDS_STACK_ENTRIES) * spill_stride] = v;
    }
    __device__ __forceinline__ int32_t get(int i) const {
        if (uint32_t(i) < LDS_STACK_ENTRIES) return base[uint32_t(i) * STRIDE];
        return spill[size_t(uint32_t(i) - LDS_STACK_ENTRIES) * spill_stride];
    }
    // a node step that finds entries i, i+1, i+2 inside the LDS part writes them without looking at the spill strip
    __device__ __forceinline__ bool room_for_three(int i) const { return uint32_t(i) + 3u <= LDS_STACK_ENTRIES; }
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
// Same, with the whole node array resident in LDS in PIECE-MAJOR order: the q-th 16 bytes of node i at
// piece[q*n_nodes + i] (q < 7: the pad is left out).  64 lanes fetching the same piece of 64 unrelated nodes then spread
// over all the 16-byte bank slots (i mod 16); in node-major order the 128-byte records would all start at the same two
// — measured on the 64-byte binary records: 31 % of the LDS cycles were bank conflicts that way.
template <uint32_t STRIDE> struct LdsStackNodes : LdsStack<STRIDE> {
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

template <bool COUNT, bool GENERAL> struct CounterSel { using type = NoCountersT<GENERAL>; };
template <bool GENERAL> struct CounterSel<true, GENERAL> { using type = LaneCountersT<GENERAL>; };

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

constexpr int TRACE_BLOCK = 256;
constexpr uint32_t SLOTS_PER_WAVE = 128; // paths owned by one wave64: 64 being traversed + up to 64 queued
constexpr uint32_t QCAP = 128;           // capacity of a wave's ray queue and hit queue (entries)

// Path state of a slot, in global memory (L2-resident), struct-of-arrays over all slots of the launch.
// (PR_HT, PU_HPRIM, PU_HINST: the slot's finished walk — only trace_kernel_stream keeps it here, when its LDS has no room for it)
enum : uint32_t { PR_OX = 0, PR_OY, PR_OZ, PR_DX, PR_DY, PR_DZ, PR_TIME, PR_TX, PR_TY, PR_TZ, PR_LX, PR_LY, PR_LZ, PR_AX, PR_AY, PR_AZ, PR_HT, PR_COUNT };
enum : uint32_t { PU_KEY_LO = 0, PU_KEY_HI, PU_BOUNCE, PU_PXROW, PU_S, PU_SEND, PU_JOB_LO, PU_JOB_HI, PU_HPRIM, PU_HINST, PU_COUNT };
// bytes of LDS one wave needs: ray queue (7 reals + slot), hit queue (t + prim + inst + meta), traversal stacks
template <typename R> __host__ __device__ constexpr uint32_t wave_lds_bytes(uint32_t stack_depth) {
    return 8u * QCAP * uint32_t(sizeof(R)) + 4u * QCAP * 4u + (LDS_STACK_ENTRIES + 1u) * 64u * 4u; // stack: + the spare slot
}
constexpr uint32_t HIT_FRESH = 0x80u; // hit-queue meta: slot (7 bits) | FRESH | box face << 8

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

template <typename R, bool COUNT, bool GENERAL>
// (at least 3 waves/SIMD: 170 VGPRs — the f32 code needs 164; the f64 code, allowed 256, ran at 2 waves/SIMD and waited on
// the fabric: spheres_1m f64 167 -> 264 Msamples/s with 140 registers spilled; 4 waves/SIMD: 205)
__global__ __launch_bounds__(TRACE_BLOCK, 3) void trace_kernel(SceneView<R> sc, CameraRec<R> cam, RenderConsts rc, R bg_r, R bg_g,
                                                            R bg_b, R t_min, R* __restrict__ partial,
                                                            unsigned long long* __restrict__ job_counter,
                                                            DeviceCounters* __restrict__ counters, R* __restrict__ pool_r,
                                                            uint32_t* __restrict__ pool_u, uint32_t n_slots, int32_t* __restrict__ spill) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    typename CounterSel<COUNT, GENERAL>::type cnt;
    const uint32_t lane = threadIdx.x & 63u, wave_in_block = threadIdx.x >> 6;
    unsigned char* wbase = lds_raw + wave_in_block * wave_lds_bytes<R>(rc.stack_depth);
    R* const rq_f = reinterpret_cast<R*>(wbase);            // ray queue [7][QCAP]: o.xyz, d.xyz, time
    R* const hq_t = rq_f + 7u * QCAP;                       // hit queue: t
    uint32_t* const rq_slot = reinterpret_cast<uint32_t*>(hq_t + QCAP);
    int32_t* const hq_prim = reinterpret_cast<int32_t*>(rq_slot + QCAP);
    int32_t* const hq_inst = hq_prim + QCAP;
    uint32_t* const hq_meta = reinterpret_cast<uint32_t*>(hq_inst + QCAP);
    LdsStack<64> stack{(LdsIntPtr)(reinterpret_cast<int32_t*>(hq_meta + QCAP) + lane), (GlobalIntPtr)(spill + (blockIdx.x * TRACE_BLOCK + threadIdx.x)), gridDim.x * TRACE_BLOCK};

    const uint32_t wave_global = blockIdx.x * (TRACE_BLOCK / 64) + wave_in_block;
    const size_t gbase = size_t(wave_global) * SLOTS_PER_WAVE;
    const unsigned long long n_jobs = rc.n_jobs;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    const V3<R> background(bg_r, bg_g, bg_b);

    // every slot starts out needing its first job
    hq_meta[lane] = lane | HIT_FRESH;
    hq_meta[lane + 64u] = (lane + 64u) | HIT_FRESH;
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
            const uint32_t hslot = meta & 0x7Fu;
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
                ps.ray.o = V3<R>(pr[size_t(PR_OX) * n_slots], pr[size_t(PR_OY) * n_slots], pr[size_t(PR_OZ) * n_slots]);
                ps.ray.d = V3<R>(pr[size_t(PR_DX) * n_slots], pr[size_t(PR_DY) * n_slots], pr[size_t(PR_DZ) * n_slots]);
                ps.ray.time = pr[size_t(PR_TIME) * n_slots];
                ps.throughput = V3<R>(pr[size_t(PR_TX) * n_slots], pr[size_t(PR_TY) * n_slots], pr[size_t(PR_TZ) * n_slots]);
                ps.radiance = V3<R>(pr[size_t(PR_LX) * n_slots], pr[size_t(PR_LY) * n_slots], pr[size_t(PR_LZ) * n_slots]);
                ps.key = (unsigned long long)pu[size_t(PU_KEY_LO) * n_slots] | ((unsigned long long)pu[size_t(PU_KEY_HI) * n_slots] << 32);
                ps.bounce = pu[size_t(PU_BOUNCE) * n_slots];
                HitRef best;
                best.prim = hq_prim[e];
                best.inst = hq_inst[e];
                best.aux = int32_t((meta >> 8) & 7u);
                const bool found = ref_kind(best.prim) != PRIM_NONE;
                // (scene view and constants re-read from the kernarg segment: the traversal loop keeps only the pointers it uses)
                if (path_shade(ps, kernarg_reload<SceneView<R>>(offsetof(TraceArgsHead<R>, sc)), kernarg_reload<RenderConsts>(offsetof(TraceArgsHead<R>, rc)), background, t_min,
                               found, hq_t[e], best, cnt)) {
                    emit = true; // next world.hit of the same path
                } else {         // main.rs:216: acc + color(...)
                    pxrow = pu[size_t(PU_PXROW) * n_slots];
                    smp = pu[size_t(PU_S) * n_slots];
                    smp_end = pu[size_t(PU_SEND) * n_slots];
                    job = (unsigned long long)pu[size_t(PU_JOB_LO) * n_slots] | ((unsigned long long)pu[size_t(PU_JOB_HI) * n_slots] << 32);
                    acc = V3<R>(pr[size_t(PR_AX) * n_slots], pr[size_t(PR_AY) * n_slots], pr[size_t(PR_AZ) * n_slots]) + ps.radiance;
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
                pu[size_t(PU_KEY_LO) * n_slots] = uint32_t(ps.key);
                pu[size_t(PU_KEY_HI) * n_slots] = uint32_t(ps.key >> 32);
                pu[size_t(PU_PXROW) * n_slots] = pxrow;
                pu[size_t(PU_S) * n_slots] = smp;
                pu[size_t(PU_SEND) * n_slots] = smp_end;
                pu[size_t(PU_JOB_LO) * n_slots] = uint32_t(job);
                pu[size_t(PU_JOB_HI) * n_slots] = uint32_t(job >> 32);
                pr[size_t(PR_AX) * n_slots] = acc.x; pr[size_t(PR_AY) * n_slots] = acc.y; pr[size_t(PR_AZ) * n_slots] = acc.z;
                emit = true;
            }
            const unsigned long long em = __ballot(emit);
            if (emit) { // the slot's next ray: path state back to memory, ray onto the queue
                pr[size_t(PR_OX) * n_slots] = ps.ray.o.x; pr[size_t(PR_OY) * n_slots] = ps.ray.o.y; pr[size_t(PR_OZ) * n_slots] = ps.ray.o.z;
                pr[size_t(PR_DX) * n_slots] = ps.ray.d.x; pr[size_t(PR_DY) * n_slots] = ps.ray.d.y; pr[size_t(PR_DZ) * n_slots] = ps.ray.d.z;
                pr[size_t(PR_TIME) * n_slots] = ps.ray.time;
                pr[size_t(PR_TX) * n_slots] = ps.throughput.x; pr[size_t(PR_TY) * n_slots] = ps.throughput.y; pr[size_t(PR_TZ) * n_slots] = ps.throughput.z;
                pr[size_t(PR_LX) * n_slots] = ps.radiance.x; pr[size_t(PR_LY) * n_slots] = ps.radiance.y; pr[size_t(PR_LZ) * n_slots] = ps.radiance.z;
                pu[size_t(PU_BOUNCE) * n_slots] = ps.bounce;
                const uint32_t idx = ray_n + uint32_t(__popcll(em & lanes_below));
                rq_f[0u * QCAP + idx] = ps.ray.o.x; rq_f[1u * QCAP + idx] = ps.ray.o.y; rq_f[2u * QCAP + idx] = ps.ray.o.z;
                rq_f[3u * QCAP + idx] = ps.ray.d.x; rq_f[4u * QCAP + idx] = ps.ray.d.y; rq_f[5u * QCAP + idx] = ps.ray.d.z;
                rq_f[6u * QCAP + idx] = ps.ray.time;
                rq_slot[idx] = hslot;
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
                wray.o = V3<R>(rq_f[0u * QCAP + e], rq_f[1u * QCAP + e], rq_f[2u * QCAP + e]);
                wray.d = V3<R>(rq_f[3u * QCAP + e], rq_f[4u * QCAP + e], rq_f[5u * QCAP + e]);
                wray.time = rq_f[6u * QCAP + e];
                slot = rq_slot[e];
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
            const uint32_t retire_batch = ray_n != 0u ? 16u : 64u;
            if constexpr (COUNT) dbg[8] += 1;
            for (;;) {
                const bool walking = has_ray && tr.node != TRAV_DONE;
                if (__ballot(walking) == 0ull) break; // every ray of the wave is finished
                if (walking) {
#pragma unroll
                    for (int k = 0; k < 6; ++k)
                        if (tr.node >= 0) trav_node_step(tr, sc, wray, t_min, stack, cnt);
                    if (tr.node < 0 && tr.node != TRAV_DONE) trav_leaf_step(tr, sc, wray, t_min, stack, cnt);
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
                hq_meta[idx] = slot | (uint32_t(tr.best.aux) << 8);
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

// ---------------------------------------------------------------------------------------------
// The STREAM form: the decoupled loop for scenes whose node records fit in LDS (one large block per CU, like
// trace_kernel_plain), with the rays kept in REGISTERS.  A wave owns 128 path slots (state in global memory, as above);
// a lane holds the ray it is walking and, most of the time, the NEXT one: the moment its walk ends it writes the hit to
// the slot (three stores, nothing waits for them), puts the slot number on the wave's hit queue and goes on with its
// next ray in the same trip — no lane waits for the longest walk of the wave, and no ray is read back from memory.
// (HITS_LDS: the hits wait in LDS instead of the slot when the block's LDS has the room.)
// When 64 hits are queued the whole wave shades them (one wait for the 64 slots' state) and hands the 64 continuation
// rays to the lanes by ds_bpermute: first to the lanes that have none, then as "next" rays.  128 slots = 64 + 64 register
// places + what is queued, so every ray finds a place.  Same per-path steps, same keyed draws, same job sums as the
// other forms: the images are bit-identical (tests/test_gpu_parity.py::test_kernel_forms_agree).
// per wave: [hit queue's t, primitive, instance when HITS_LDS] + its slot | FRESH | face << 8 entries + the hand-over's lane table
inline constexpr uint32_t stream_wave_bytes(uint32_t real_bytes, bool hits_lds) { return (hits_lds ? QCAP * (real_bytes + 8u) : 0u) + QCAP * 2u + 64u * 2u; }
inline size_t stream_form_bytes(uint32_t n_nodes4, uint32_t stack_depth, uint32_t block, uint32_t real_bytes, bool hits_lds) {
    return lds_form_bytes(n_nodes4, stack_depth, block) + size_t(block / 64u) * stream_wave_bytes(real_bytes, hits_lds);
}
// (Path state is read and written with ordinary accesses: non-temporal ones, and agent-scope ones that bypass the vector
// cache, were both slower: final_scene f32 1187 and 1271 against 1329 Msamples/s.)
__device__ __forceinline__ float lane_pull(float v, uint32_t src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(int(src_lane << 2), __float_as_int(v)));
}
__device__ __forceinline__ uint32_t lane_pull(uint32_t v, uint32_t src_lane) {
    return uint32_t(__builtin_amdgcn_ds_bpermute(int(src_lane << 2), int(v)));
}
__device__ __forceinline__ double lane_pull(double v, uint32_t src_lane) {
    const int lo = __builtin_amdgcn_ds_bpermute(int(src_lane << 2), __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(int(src_lane << 2), __double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <typename R> __device__ __forceinline__ Ray<R> lane_pull(const Ray<R>& r, uint32_t src_lane) {
    Ray<R> g;
    g.o = V3<R>(lane_pull(r.o.x, src_lane), lane_pull(r.o.y, src_lane), lane_pull(r.o.z, src_lane));
    g.d = V3<R>(lane_pull(r.d.x, src_lane), lane_pull(r.d.y, src_lane), lane_pull(r.d.z, src_lane));
    g.time = lane_pull(r.time, src_lane);
    return g;
}

template <typename R, bool COUNT, int BLOCK, bool GENERAL, bool HITS_LDS>
__global__ __launch_bounds__(BLOCK, 1) void trace_kernel_stream(SceneView<R> sc, CameraRec<R> cam, RenderConsts rc, R bg_r, R bg_g, R bg_b, R t_min,
                                                                R* __restrict__ partial, unsigned long long* __restrict__ job_counter,
                                                                DeviceCounters* __restrict__ counters, R* __restrict__ pool_r,
                                                                uint32_t* __restrict__ pool_u, uint32_t n_slots, int32_t* __restrict__ spill) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    typename CounterSel<COUNT, GENERAL>::type cnt;
    const uint32_t lane = threadIdx.x & 63u, wave_in_block = threadIdx.x >> 6;
    constexpr int NODE_STEPS = RT_NODE_STEPS; // the trips of closest_solid()
    LdsStackNodes<BLOCK> stack;
    stack.spill = (GlobalIntPtr)(spill + (blockIdx.x * BLOCK + threadIdx.x));
    stack.spill_stride = gridDim.x * BLOCK;
    uint16_t *hq, *tbl;
    R* hq_t = nullptr;
    int32_t *hq_prim = nullptr, *hq_inst = nullptr;
    { // [node pieces][stacks of the block][per wave: hit queue, lane table]
        const uint32_t n = rc.lds_nodes;
        const int4* src = reinterpret_cast<const int4*>(sc.nodes);
        int4* dst = reinterpret_cast<int4*>(lds_raw);
        for (uint32_t i = threadIdx.x; i < n * 8u; i += BLOCK)
            if ((i & 7u) < BVH4_USED_SIXTEENTHS) dst[(i & 7u) * n + (i >> 3)] = src[i];
        int32_t* const stacks = reinterpret_cast<int32_t*>(lds_raw) + n * (4u * BVH4_USED_SIXTEENTHS);
        stack.base = (LdsIntPtr)(stacks + threadIdx.x);
        stack.piece = dst;
        stack.n_nodes = n;
        unsigned char* wbase = reinterpret_cast<unsigned char*>(stacks + (LDS_STACK_ENTRIES + 1u) * BLOCK) + wave_in_block * stream_wave_bytes(sizeof(R), HITS_LDS);
        if constexpr (HITS_LDS) {
            hq_t = reinterpret_cast<R*>(wbase);
            hq_prim = reinterpret_cast<int32_t*>(hq_t + QCAP);
            hq_inst = hq_prim + QCAP;
            wbase = reinterpret_cast<unsigned char*>(hq_inst + QCAP);
        }
        hq = reinterpret_cast<uint16_t*>(wbase);
        tbl = hq + QCAP;
    }
    const uint32_t wave_global = blockIdx.x * (BLOCK / 64) + wave_in_block;
    const size_t gbase = size_t(wave_global) * SLOTS_PER_WAVE;
    const unsigned long long n_jobs = rc.n_jobs;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    const V3<R> background(bg_r, bg_g, bg_b);

    hq[lane] = uint16_t(lane | HIT_FRESH); // every slot starts out needing its first job
    hq[lane + 64u] = uint16_t((lane + 64u) | HIT_FRESH);
    __syncthreads(); // the node records are in place
    uint32_t hit_n = SLOTS_PER_WAVE;                  // wave-uniform fill level of the hit queue
    unsigned long long batch_next = 0, batch_end = 0; // the wave's reserved batch of job indices
    uint32_t dbg[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    bool has_ray = false, has_next = false;
    uint32_t slot = 0, nslot = 0;
    Ray<R> wray, nray; // the ray being walked and the one after it, in world space
    Trav<R> tr;

    for (;;) {
        __builtin_amdgcn_wave_barrier();
        if (hit_n >= 64u || __ballot(has_ray) == 0ull) {
            if (hit_n == 0u) break; // nothing walking, nothing queued: this wave is done
            // ================================================================== SHADE (up to 64 queued hits)
            const uint32_t m = hit_n < 64u ? hit_n : 64u;
            if constexpr (COUNT) { dbg[9] += 1; dbg[10] += m; }
            long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
            if constexpr (COUNT) c1 = c0 = clock64();
            const bool on = lane < m;
            const uint32_t e = hit_n - 1u - (on ? lane : 0u);
            hit_n -= m;
            const uint32_t meta = on ? uint32_t(hq[e]) : HIT_FRESH;
            const uint32_t hslot = meta & 0x7Fu;
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
                ps.ray.o = V3<R>(pr[size_t(PR_OX) * n_slots], pr[size_t(PR_OY) * n_slots], pr[size_t(PR_OZ) * n_slots]);
                ps.ray.d = V3<R>(pr[size_t(PR_DX) * n_slots], pr[size_t(PR_DY) * n_slots], pr[size_t(PR_DZ) * n_slots]);
                ps.ray.time = pr[size_t(PR_TIME) * n_slots];
                ps.throughput = V3<R>(pr[size_t(PR_TX) * n_slots], pr[size_t(PR_TY) * n_slots], pr[size_t(PR_TZ) * n_slots]);
                ps.radiance = V3<R>(pr[size_t(PR_LX) * n_slots], pr[size_t(PR_LY) * n_slots], pr[size_t(PR_LZ) * n_slots]);
                ps.key = (unsigned long long)pu[size_t(PU_KEY_LO) * n_slots] | ((unsigned long long)pu[size_t(PU_KEY_HI) * n_slots] << 32);
                ps.bounce = pu[size_t(PU_BOUNCE) * n_slots];
                HitRef best;
                R hit_t;
                if constexpr (HITS_LDS) {
                    hit_t = hq_t[e]; best.prim = hq_prim[e]; best.inst = hq_inst[e];
                } else {
                    hit_t = pr[size_t(PR_HT) * n_slots];
                    best.prim = int32_t(pu[size_t(PU_HPRIM) * n_slots]);
                    best.inst = int32_t(pu[size_t(PU_HINST) * n_slots]);
                }
                best.aux = int32_t((meta >> 8) & 7u);
                const bool found = ref_kind(best.prim) != PRIM_NONE;
                if constexpr (COUNT) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); c1 = clock64(); }
                if (path_shade(ps, kernarg_reload<SceneView<R>>(offsetof(TraceArgsHead<R>, sc)), kernarg_reload<RenderConsts>(offsetof(TraceArgsHead<R>, rc)), background, t_min,
                               found, hit_t, best, cnt)) {
                    emit = true; // next world.hit of the same path
                } else {         // main.rs:216: acc + color(...)
                    pxrow = pu[size_t(PU_PXROW) * n_slots];
                    smp = pu[size_t(PU_S) * n_slots];
                    smp_end = pu[size_t(PU_SEND) * n_slots];
                    job = (unsigned long long)pu[size_t(PU_JOB_LO) * n_slots] | ((unsigned long long)pu[size_t(PU_JOB_HI) * n_slots] << 32);
                    acc = V3<R>(pr[size_t(PR_AX) * n_slots], pr[size_t(PR_AY) * n_slots], pr[size_t(PR_AZ) * n_slots]) + ps.radiance;
                    ++smp;
                    need_sample = true;
                }
            } else if (on) {
                need_sample = true; // fresh slot: smp == smp_end == 0, no job yet
            }
            if constexpr (COUNT) c2 = clock64();
            // job hand-out for the slots whose job is finished: wave-aggregated (an empty job — a tile pixel outside the
            // image — is finished at once, hence the loop)
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
                pu[size_t(PU_KEY_LO) * n_slots] = uint32_t(ps.key);
                pu[size_t(PU_KEY_HI) * n_slots] = uint32_t(ps.key >> 32);
                pu[size_t(PU_PXROW) * n_slots] = pxrow;
                pu[size_t(PU_S) * n_slots] = smp;
                pu[size_t(PU_SEND) * n_slots] = smp_end;
                pu[size_t(PU_JOB_LO) * n_slots] = uint32_t(job);
                pu[size_t(PU_JOB_HI) * n_slots] = uint32_t(job >> 32);
                pr[size_t(PR_AX) * n_slots] = acc.x; pr[size_t(PR_AY) * n_slots] = acc.y; pr[size_t(PR_AZ) * n_slots] = acc.z;
                emit = true;
            }
            if constexpr (COUNT) c3 = clock64();
            const unsigned long long em = __ballot(emit);
            if (emit) { // the slot's next ray: path state back to memory, this lane into the hand-over's table
                pr[size_t(PR_OX) * n_slots] = ps.ray.o.x; pr[size_t(PR_OY) * n_slots] = ps.ray.o.y; pr[size_t(PR_OZ) * n_slots] = ps.ray.o.z;
                pr[size_t(PR_DX) * n_slots] = ps.ray.d.x; pr[size_t(PR_DY) * n_slots] = ps.ray.d.y; pr[size_t(PR_DZ) * n_slots] = ps.ray.d.z;
                pr[size_t(PR_TIME) * n_slots] = ps.ray.time;
                pr[size_t(PR_TX) * n_slots] = ps.throughput.x; pr[size_t(PR_TY) * n_slots] = ps.throughput.y; pr[size_t(PR_TZ) * n_slots] = ps.throughput.z;
                pr[size_t(PR_LX) * n_slots] = ps.radiance.x; pr[size_t(PR_LY) * n_slots] = ps.radiance.y; pr[size_t(PR_LZ) * n_slots] = ps.radiance.z;
                pu[size_t(PU_BOUNCE) * n_slots] = ps.bounce;
                tbl[__popcll(em & lanes_below)] = uint16_t(lane);
            }
            const uint32_t n_emit = uint32_t(__popcll(em));
            if (n_emit != 0u) {
                __builtin_amdgcn_wave_barrier();
                uint32_t given = 0;
                { // first the lanes that have no ray: they start its walk here
                    const bool want = !has_ray;
                    const unsigned long long wm = __ballot(want);
                    const uint32_t r = uint32_t(__popcll(wm & lanes_below));
                    const bool take = want && r < n_emit;
                    const uint32_t src = take ? uint32_t(tbl[r]) : lane;
                    const Ray<R> got = lane_pull(ps.ray, src);
                    const uint32_t got_slot = lane_pull(hslot, src);
                    if (take) {
                        wray = got; slot = got_slot; has_ray = true;
                        cnt.ray();
                        trav_begin(tr, sc, wray, stack);
                    }
                    const uint32_t wn = uint32_t(__popcll(wm));
                    given = wn < n_emit ? wn : n_emit;
                    if constexpr (COUNT) { dbg[11] += 1; dbg[12] += given; }
                }
                if (given < n_emit) { // the others become "next" rays
                    const bool want = !has_next;
                    const unsigned long long wm = __ballot(want);
                    const uint32_t r = given + uint32_t(__popcll(wm & lanes_below));
                    const bool take = want && r < n_emit;
                    const uint32_t src = take ? uint32_t(tbl[r]) : lane;
                    const Ray<R> got = lane_pull(ps.ray, src);
                    const uint32_t got_slot = lane_pull(hslot, src);
                    if (take) { nray = got; nslot = got_slot; has_next = true; }
                }
            }
            if constexpr (COUNT) {
                c4 = clock64();
                if (lane == 0) { dbg[0] += uint32_t((c1 - c0) >> 4); dbg[1] += uint32_t((c2 - c1) >> 4); dbg[2] += uint32_t((c3 - c2) >> 4); dbg[3] += uint32_t((c4 - c3) >> 4); }
            }
            continue;
        }

        // ====================================================================== WALK until 64 hits are queued
        if constexpr (COUNT) dbg[8] += 1;
        long long w0 = 0;
        if constexpr (COUNT) w0 = clock64();
        for (;;) {
            if (__ballot(has_ray) == 0ull) break;
            if constexpr (COUNT) { dbg[13] += 1; dbg[14] += uint32_t(__popcll(__ballot(has_ray))); }
            if (has_ray) {
#pragma unroll
                for (int k = 0; k < NODE_STEPS; ++k)
                    if (tr.node >= 0) trav_node_step(tr, sc, wray, t_min, stack, cnt);
                if (tr.node < 0 && tr.node != TRAV_DONE) trav_leaf_step(tr, sc, wray, t_min, stack, cnt);
            }
            const bool fin = has_ray && tr.node == TRAV_DONE;
            const unsigned long long fm = __ballot(fin);
            if (fm != 0ull) {
                if (fin) { // the hit goes to the slot, the slot onto the queue, the lane on to its next ray
                    const uint32_t idx = hit_n + uint32_t(__popcll(fm & lanes_below));
                    const int32_t hprim = tr.found ? tr.best.prim : make_ref(PRIM_NONE, 0);
                    if constexpr (HITS_LDS) {
                        hq_t[idx] = tr.closest; hq_prim[idx] = hprim; hq_inst[idx] = tr.best.inst;
                    } else {
                        const size_t g = gbase + slot;
                        pool_r[size_t(PR_HT) * n_slots + g] = tr.closest;
                        pool_u[size_t(PU_HPRIM) * n_slots + g] = uint32_t(hprim);
                        pool_u[size_t(PU_HINST) * n_slots + g] = uint32_t(tr.best.inst);
                    }
                    hq[idx] = uint16_t(slot | (uint32_t(tr.best.aux) << 8));
                    has_ray = has_next;
                    if (has_next) {
                        wray = nray; slot = nslot; has_next = false;
                        cnt.ray();
                        trav_begin(tr, sc, wray, stack);
                    }
                }
                hit_n += uint32_t(__popcll(fm));
                if (hit_n >= 64u) break;
            }
        }
        if constexpr (COUNT) {
            if (lane == 0) dbg[4] += uint32_t((clock64() - w0) >> 4);
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

// The plain form of the same loop: a lane OWNS a path (and its job) and alternates "regenerate or advance by one
// bounce" (rt_core.hpp path_step = whole BVH walk + shade) with the wave-aggregated job fetch.  Simpler, less
// bookkeeping per ray, but every lane waits for the longest BVH walk of the wave at every bounce.  Kept beside the
// decoupled kernel because which of the two is faster depends on the scene (DESIGN.md "Kernels").
template <typename R, bool COUNT, int BLOCK, bool LDSN, bool GENERAL>
// (the 256-thread form — nodes in global memory — asks for at least 3 waves/SIMD like the decoupled kernel: its f64 code,
// allowed 256 VGPRs, ran at 2: a 20 000-sphere scene 29.9 -> 13.7 ms per 67 Msamples)
__global__ __launch_bounds__(BLOCK, BLOCK == 256 ? 3 : 1) void trace_kernel_plain(SceneView<R> sc, CameraRec<R> cam, RenderConsts rc, R bg_r, R bg_g,
                                                            R bg_b, R t_min, R* __restrict__ partial,
                                                            unsigned long long* __restrict__ job_counter,
                                                            DeviceCounters* __restrict__ counters, int32_t* __restrict__ spill) {
    // LDSN: the whole node array is copied into LDS (piece-major, see LdsStackNodes) in front of the stacks — small
    // scenes: one dependent ~100-cycle LDS read per node visit instead of an L1/L2 round trip
    extern __shared__ __align__(16) int32_t lds_stack[];
    typename std::conditional<LDSN, LdsStackNodes<BLOCK>, LdsStack<BLOCK>>::type stack;
    stack.spill = (GlobalIntPtr)(spill + (blockIdx.x * blockDim.x + threadIdx.x));
    stack.spill_stride = gridDim.x * blockDim.x;
    if constexpr (LDSN) {
        const uint32_t n = rc.lds_nodes;
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
    unsigned long long job = 0;
    uint32_t px = 0, row = 0, s = 0, s_end = 0;
    V3<R> acc;
    PathState<R> ps;

    // counting variant only: where a wave's time and lanes go (RTTNW_DEBUG_SCHED prints it) — wave clock per phase
    // [0..3], lockstep iterations of the BVH walk [4] (with a node lane [7], with a leaf lane [8]) against the lane
    // steps they served [5] node / [6] leaf, bounce rounds [9] and the lanes alive in them [10], regenerations [11,12]
    unsigned long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (;;) {
        long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0;
        if constexpr (COUNT) tk0 = clock64();
        // ---- job hand-out: wave-aggregated, one atomic per refill event
        const bool need = !done && !alive && s >= s_end;
        const unsigned long long mask = __ballot(need);
        if (mask != 0ull) {
            if (need && has_job) { // retire the finished job: its sequential sum
                R* dst = partial + job * 3ull;
                dst[0] = acc.x; dst[1] = acc.y; dst[2] = acc.z;
                has_job = false;
            }
            const unsigned long long mine = wave_take_jobs(mask, lane, batch_next, batch_end, job_counter);
            if (need) {
                job = mine;
                if (job >= n_jobs) {
                    done = true;
                } else {
                    const JobInfo ji = job_decode(rc, uint32_t(job));
                    px = ji.px; row = ji.row; s = ji.s; s_end = ji.s_end;
                    job = ji.sum_index; // from here on: where the job's sum goes
                    acc = V3<R>();
                    has_job = ji.real; // padding jobs have no sum to write
                }
            }
        }
        if (__ballot(!done) == 0ull) break;

        // ---- one path per lane: regenerate or advance by one bounce
        if constexpr (!COUNT) {
            if (!done) {
                if (!alive && s < s_end) {
                    path_begin(ps, kernarg_reload<CameraRec<R>>(offsetof(TraceArgsHead<R>, cam)), rc, px, row, s);
                    alive = true;
                }
                if (alive) {
                    alive = path_step(ps, sc, rc, background, t_min, stack, cnt);
                    if (!alive) { // main.rs:216: acc + color(...)
                        acc = acc + ps.radiance;
                        ++s;
                    }
                }
            }
        } else { // the same steps, with the wave clock read between the phases and the lockstep loop tallied
            tk1 = clock64();
            const bool begin = !done && !alive && s < s_end;
            const unsigned long long bm = __ballot(begin);
            if (begin) {
                path_begin(ps, cam, rc, px, row, s);
                alive = true;
            }
            tk2 = clock64();
            const unsigned long long am = __ballot(alive);
            bool found = false;
            R closest = R(0);
            HitRef best;
            best.prim = 0; best.inst = -1; best.aux = 0;
            uint32_t my_trips = 0;
            if (alive) {
                cnt.ray();
                Trav<R> tr;
                trav_begin(tr, sc, ps.ray, stack);
                while (tr.node != TRAV_DONE) {
                    ++my_trips;
                    // the loop body of closest_solid() (two node steps, then a leaf step for the lanes at a leaf by then), tallied
                    const unsigned long long act = __ballot(true);
                    const bool is_node = tr.node >= 0;
                    const unsigned long long nm = __ballot(is_node);
                    const long long q0 = clock64();
                    if (is_node) { prof[5] += 1; trav_node_step(tr, sc, ps.ray, t_min, stack, cnt); }
                    if (tr.node >= 0) { prof[5] += 1; trav_node_step(tr, sc, ps.ray, t_min, stack, cnt); }
                    const long long q1 = clock64();
                    const bool is_leaf = tr.node < 0 && tr.node != TRAV_DONE;
                    const unsigned long long lm = __ballot(is_leaf);
                    uint32_t kmask = 0; // kinds among the leaf lanes: bit k = record kind k, bit 5 = empty slot
                    {
                        const uint32_t kd = tr.node == CHILD_EMPTY ? 5u : leaf_kind(tr.node);
#pragma unroll
                        for (uint32_t k = 0; k < 6; ++k) kmask |= __ballot(is_leaf && kd == k) != 0ull ? (1u << k) : 0u;
                    }
                    const long long q1b = clock64();
                    if (is_leaf) { prof[6] += 1; trav_leaf_step(tr, sc, ps.ray, t_min, stack, cnt); }
                    const long long q2 = clock64();
                    if (lane == uint32_t(__ffsll((long long)act) - 1)) {
                        prof[4] += 1;
                        prof[7] += nm != 0ull;
                        prof[8] += lm != 0ull;
                        prof[13] += (unsigned long long)(q1 - q0);
                        prof[14] += (unsigned long long)(q2 - q1b);
                        if (kmask && rc.profile == 2u) { // leaf time by the set of record kinds the iteration served: dbg[16+set], count dbg[80+set]
                            atomicAdd(&counters->dbg[16 + kmask], (unsigned long long)(q2 - q1b));
                            atomicAdd(&counters->dbg[80 + kmask], 1ull);
                        }
                    }
                }
                found = tr.found; closest = tr.closest; best = tr.best;
                if (rc.profile == 3u) {
                    atomicAdd(&counters->dbg[16 + min(my_trips, 63u)], 1ull); // histogram of trips per walk
                    // ... and trips by what the walk found: 0 miss, 1 + kind (sphere, moving, rect, box), 6 anything inside an instance
                    const uint32_t cls = !tr.found ? 0u : (tr.best.inst >= 0 ? 6u : 1u + ref_kind(tr.best.prim));
                    atomicAdd(&counters->dbg[144 + cls], (unsigned long long)my_trips);
                    atomicAdd(&counters->dbg[152 + cls], 1ull);
                }
            }
            if (rc.profile == 3u) { // ... and of the trips of the wave's longest walk, per round
                uint32_t mx = my_trips;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) mx = max(mx, uint32_t(__shfl_xor(int(mx), off, 64)));
                if (lane == 0) atomicAdd(&counters->dbg[80 + min(mx, 63u)], 1ull);
            }
            tk3 = clock64();
            long long tk3b = tk3;
            if (alive) { // path_shade(), with the wave clock read between its two halves
                HitRecord<R> rec;
                const bool hit = world_hit_finish(sc, ps.ray, t_min, ps.key, ps.bounce, rc.quirks, found, closest, best, rec, cnt);
                tk3b = clock64();
                if (!hit) {
                    ps.radiance = ps.radiance + ps.throughput * background;
                    alive = false;
                } else {
                    V3<R> att, emitted;
                    const bool cont = shade(sc, rec, ps.key, ps.bounce, ps.ray, att, emitted, cnt);
                    ps.radiance = ps.radiance + ps.throughput * emitted;
                    if (cont) { ps.throughput = ps.throughput * att; ps.bounce += 1; }
                    alive = cont && ps.bounce < rc.max_depth;
                }
                if (!alive) {
                    acc = acc + ps.radiance;
                    ++s;
                }
            }
            const long long tk4 = clock64();
            if (lane == 0) {
                prof[0] += (unsigned long long)(tk1 - tk0);
                prof[1] += (unsigned long long)(tk2 - tk1);
                prof[2] += (unsigned long long)(tk3 - tk2);
                prof[3] += (unsigned long long)(tk4 - tk3);
                prof[15] += (unsigned long long)(tk3b - tk3);
                prof[9] += 1;
                prof[10] += (unsigned long long)__popcll(am);
                prof[11] += bm != 0ull;
                prof[12] += (unsigned long long)__popcll(bm);
            }
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

// Sum a pixel's chunk partials of ONE PASS in chunk order and add them to the pixel's running sum (pass 0 starts it);
// the last pass divides by the render's spp (main.rs:217): packed pixel records (r, g, b, 1).  Pad tiles
// (>= my_tiles) are zero-filled.  The running sum lives in `packed` itself.
template <typename R>
__global__ void resolve_kernel(const R* __restrict__ partial, R* __restrict__ packed, RenderConsts rc, uint32_t pixels_per_rank,
                               uint32_t first_pass, uint32_t last_pass, uint32_t total_spp) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pixels_per_rank) return;
    R* dst = packed + (unsigned long long)p * 4ull;
    R r = 0, g = 0, b = 0, a = 0;
    const unsigned long long jobs_per_chunk = (unsigned long long)rc.my_tiles * 64ull;
    if (p < jobs_per_chunk) {
        for (uint32_t c = 0; c < rc.n_chunks; ++c) {
            const R* src = partial + ((unsigned long long)c * jobs_per_chunk + p) * 3ull;
            r = r + src[0]; g = g + src[1]; b = b + src[2];
        }
        if (!first_pass) { r = dst[0] + r; g = dst[1] + g; b = dst[2] + b; }
        if (last_pass) {
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
    ProbeCounters cnt; // every graph shape; (u, v) evaluated at every hit that reads them (the trace kernels defer them, rt_core.hpp)
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
    DevBuf<Bvh4Node> nodes;
    DevBuf<SphereRec<R>> spheres;
    DevBuf<int32_t> sphere_mat, sphere_seq;
    DevBuf<MovingSphereRec<R>> moving;
    DevBuf<RectRec<R>> rects;
    DevBuf<BoxRec<R>> boxes;
    DevBuf<InstanceRec<R>> insts;
    DevBuf<MediumRec<R>> media;
    DevBuf<int32_t> medium_refs;
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
        for (auto& r : f.rects) rc_.push_back({R(r.a0), R(r.a1), R(r.b0), R(r.b1), R(r.k), r.plane, r.mat, r.seq});
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
            o.n_ops = i.n_ops; o.root = i.root; o.single_leaf = i.single_leaf;
            for (int k = 0; k < MAX_INSTANCE_OPS; ++k) {
                o.ops[k].type = i.ops[k].type;
                for (int c = 0; c < 3; ++c) o.ops[k].v[c] = R(i.ops[k].v[c]);
            }
            in.push_back(o);
        }
        std::vector<MediumRec<R>> md;
        for (auto& m : f.media) md.push_back({m.b_first, m.b_count, m.inst, m.n_outer, m.mat, m.ref0, R(m.neg_inv_density)});
        std::vector<MaterialRec<R>> mt;
        for (auto& m : f.mats) mt.push_back({m.type, m.tex, {R(m.albedo[0]), R(m.albedo[1]), R(m.albedo[2])}, R(m.param)});
        std::vector<TextureRec<R>> tx;
        for (auto& t : f.texs) tx.push_back({t.type, t.a, t.b, 0, {R(t.color[0]), R(t.color[1]), R(t.color[2])}, R(t.scale)});
        std::vector<R> pv;
        for (double v : f.perlin_vec) pv.push_back(R(v));

        int rc;
        if ((rc = nodes.upload(f.nodes4)) || (rc = spheres.upload(sp)) || (rc = sphere_mat.upload(f.sphere_mat)) ||
            (rc = sphere_seq.upload(f.sphere_seq)) || (rc = moving.upload(mv)) || (rc = rects.upload(rc_)) ||
            (rc = boxes.upload(bx)) || (rc = insts.upload(in)) || (rc = media.upload(md)) || (rc = medium_refs.upload(f.medium_refs)) || (rc = mats.upload(mt)) ||
            (rc = texs.upload(tx)) || (rc = images.upload(f.images)) || (rc = texels.upload(f.texels)) ||
            (rc = perlin_vec.upload(pv)) || (rc = perlin_perm.upload(f.perlin_perm)))
            return rc;
        view.nodes = nodes.p; view.spheres = spheres.p; view.sphere_mat = sphere_mat.p; view.sphere_seq = sphere_seq.p;
        view.moving = moving.p; view.rects = rects.p; view.boxes = boxes.p; view.insts = insts.p; view.media = media.p; view.medium_refs = medium_refs.p;
        view.mats = mats.p; view.texs = texs.p; view.images = images.p; view.texels = texels.p;
        view.perlin_vec = perlin_vec.p; view.perlin_perm = perlin_perm.p;
        view.top_root = f.top_root;
        view.n_media = int32_t(f.media.size());
        bytes = f.nodes4.size() * sizeof(Bvh4Node) + sp.size() * sizeof(SphereRec<R>) + mv.size() * sizeof(MovingSphereRec<R>) +
                rc_.size() * sizeof(RectRec<R>) + bx.size() * sizeof(BoxRec<R>) + in.size() * sizeof(InstanceRec<R>);
        ready = true;
        return 0;
    }
    void release() {
        nodes.release(); spheres.release(); sphere_mat.release(); sphere_seq.release(); moving.release(); rects.release();
        boxes.release(); insts.release(); media.release(); medium_refs.release(); mats.release(); texs.release(); images.release();
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
    void* pool_r = nullptr; size_t pool_r_bytes = 0; // path-slot state (reals / words), SoA over all slots
    void* pool_u = nullptr; size_t pool_u_bytes = 0;
    void* spill = nullptr; size_t spill_bytes = 0;   // traversal-stack entries beyond LDS_STACK_ENTRIES, per thread of the launch
    unsigned long long* job_counter = nullptr; // [0] job counter, then DeviceCounters
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t stream = nullptr;              // rttnw_render_multi: this device's launch stream
    void* multi_packed = nullptr; size_t multi_packed_bytes = 0; // packed tiles of the logical ranks living on this device
    void* gathered = nullptr; size_t gathered_bytes = 0;         // root device: every rank's packed tiles
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
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (d->device >= 0) (void)hipSetDevice(d->device);
    struct Restore { int dev; ~Restore() { if (dev >= 0) (void)hipSetDevice(dev); } } restore{prev};
    d->s32.release(); d->s64.release();
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

int device_bvh_builder(::rttnw_scene* s, BvhBuilder& out, std::string& err) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        err = "no HIP device available (the device BVH builder has no CPU fallback)";
        return RTTNW_ERR_HIP;
    }
    out = [s](const std::vector<BuildPrim>& prims, std::vector<BvhNode>& nodes, int32_t& root, uint32_t& levels, std::string& e) {
        return lbvh_build_device(prims, nodes, root, levels, &s->build_kernel_ms, e);
    };
    return 0;
}

// State on the CURRENT device (job counter, events; the scene arrays follow on first use).
static int device_state_create(DeviceState*& out, std::string& err) {
    DeviceState* d = new DeviceState();
    hipError_t e = hipGetDevice(&d->device);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, d->device);
    if (e != hipSuccess) { err = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e); device_release(d); return RTTNW_ERR_HIP; }
    d->num_cus = prop.multiProcessorCount;
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
            BvhBuilder device_builder;
            const bool on_device = s->bvh_builder == RTTNW_BVH_DEVICE_LBVH;
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
                d->s32.release(); d->s64.release();
            }
            if (prev >= 0) (void)hipSetDevice(prev);
            s->flat = std::move(wider);
        }
    }
    return 0;
}

template <typename R>
// prepare_only: upload the scene on first use and grow every workspace buffer this render will need (blocking hipMalloc /
// hipMemcpy / hipFree calls), launch nothing — rttnw_render_multi does that for ALL its ranks before the first launch, so
// that no allocation (a device-wide synchronisation) sits between two ranks' kernels.
int render_tiles_t(::rttnw_scene* s, DeviceState* d, const rttnw_camera_desc* cam, const rttnw_params* p, void* d_packed, hipStream_t stream,
                   rttnw_stats* stats, bool sync_for_stats = true, bool prepare_only = false) {
    HIP_TRY(hipSetDevice(d->device));
    DeviceScene<R>& ds = scene_of<R>(d);
    if (!ds.ready)
        if (int rc = ds.upload(s->flat)) return rc;

    rttnw_tile_layout L;
    fill_layout(p->width, p->height, p->tile_world, L);
    RenderConsts rc{};
    rc.width = p->width; rc.height = p->height; rc.spp = p->spp; rc.max_depth = p->max_depth;
    rc.tiles_x = L.tiles_x; rc.tiles_y = L.tiles_y; rc.n_tiles = L.n_tiles;
    rc.tile_rank = p->tile_rank; rc.tile_world = p->tile_world;
    rc.my_tiles = L.n_tiles > p->tile_rank ? (L.n_tiles - p->tile_rank + p->tile_world - 1) / p->tile_world : 0;
    rc.quirks = p->quirks; rc.seed = p->seed; rc.stack_depth = s->flat.stack_depth;
    rc.profile = p->collect_counters;
    rc.sample_begin = p->sample_begin;
    rc.div_tiles_x = make_fastdiv(std::max<uint32_t>(1u, rc.tiles_x));
    // Passes over consecutive sample ranges (rt_types.hpp plan_passes): one for ordinary renders; more when the chunk
    // sums of the whole render would not fit the workspace budget.  Decided by the image size and spp alone.
    const uint32_t n_pass = plan_passes(p->spp, p->spp_chunk, uint64_t(L.n_tiles) * 64, 3 * sizeof(R));

    CameraRec<double> cam64;
    make_camera(cam->lookfrom, cam->lookat, cam->view_up, cam->vertical_fov, cam->aspect_ratio, cam->aperture,
                cam->focus_distance, cam->open_time, cam->close_time, cam64);
    const CameraRec<R> camr = narrow_camera<R>(cam64);

    const bool count = p->collect_counters != 0;
    // Two forms of the same loop (DESIGN.md "Kernels"): measured on MI355X the lane-owns-a-path form wins on shallow
    // scenes (cornell_box, final_scene: <= ~1k nodes), the decoupled form on deep BVHs where traversal lengths vary
    // most (1M spheres).  RTTNW_KERNEL=plain|wave overrides the choice (experiments only).
    const char* kv = getenv("RTTNW_KERNEL");
    // (crossover measured on spheres_1m-like scenes of 2e4 - 2.5e5 spheres, 512x512 spp 256: f32 at ~24 k 4-wide nodes — 19.6 k:
    // 3105 against 2972 Msamples/s, 28.3 k: 2258 against 2452 — f64 at ~50 k — 28.3 k: 2022 against 1740, 50.9 k: 1284 against 1318)
    bool plain = s->flat.nodes4.size() < (sizeof(R) == 4 ? 24576u : 49152u);
    if (kv && (std::strcmp(kv, "plain") == 0 || std::strcmp(kv, "plainglobal") == 0)) plain = true;
    if (kv && std::strcmp(kv, "wave") == 0) plain = false;
    // RTTNW_KERNEL=stream: the LDS-node form of the decoupled loop (trace_kernel_stream; experiments: it reaches 0.41-0.44
    // lane utilisation but is slower than the lane-owns-path form on every scene measured, profiles/r02/README.md)
    constexpr int STREAM_BLOCK = sizeof(R) == 4 ? RT_F32_STREAM_BLOCK : RT_F64_BLOCK;
    const uint32_t n4_all = uint32_t(s->flat.nodes4.size());
    const bool stream_hits_lds = stream_form_bytes(n4_all, s->flat.stack_depth, STREAM_BLOCK, sizeof(R), true) <= 160 * 1024;
    const bool streamf = kv && std::strcmp(kv, "stream") == 0 && stream_form_bytes(n4_all, s->flat.stack_depth, STREAM_BLOCK, sizeof(R), stream_hits_lds) <= 160 * 1024;
    if (streamf) plain = false;
    if (!prepare_only) HIP_TRY(hipMemsetAsync(d->job_counter, 0, sizeof(unsigned long long) + sizeof(DeviceCounters), stream));
    DeviceCounters* dc = reinterpret_cast<DeviceCounters*>(d->job_counter + 1);
    auto persistent_grid = [&](const void* kernel, size_t lds_bytes, size_t waves_needed, size_t& grid) -> int {
        if (lds_bytes > 160 * 1024) { set_last_error("render: queues + traversal stacks do not fit in LDS"); return RTTNW_ERR_UNSUPPORTED; }
        HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
        int blocks_per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kernel, TRACE_BLOCK, lds_bytes));
        blocks_per_cu = std::max(1, std::min(blocks_per_cu, 8));
        // what the chip holds at once (no inter-workgroup dependency, so a little over-subscription is harmless),
        // but never more waves than there is work for
        grid = std::max<size_t>(1, std::min<size_t>(size_t(d->num_cus) * blocks_per_cu, (waves_needed + 3) / 4));
        return 0;
    };
    // stack entries beyond the LDS-resident ones, for every thread of a launch
    auto grow_spill = [&](size_t threads) -> int {
        const size_t extra = rc.stack_depth > LDS_STACK_ENTRIES ? rc.stack_depth - LDS_STACK_ENTRIES : 0;
        return grow(&d->spill, &d->spill_bytes, std::max<size_t>(threads * extra, 1) * sizeof(int32_t));
    };
    // One pass: trace kernel over the pass's jobs, then the resolve step.
    auto trace_pass = [&]() -> int {
        const size_t n_jobs = rc.n_jobs;
        if (plain) {
            // Small scenes: node array in LDS, in ONE large block per CU so that nodes + all the lanes' stacks fit in 160 KB:
            // 1024 threads (4 waves/SIMD at <= 128 VGPRs) for f32, 512 threads (2 waves/SIMD, all the 256-VGPR f64 code allows)
            constexpr int LDS_BLOCK = sizeof(R) == 4 ? 1024 : RT_F64_BLOCK;
            const uint32_t n4 = uint32_t(s->flat.nodes4.size());
            const bool want_lds = !(kv && std::strcmp(kv, "plainglobal") == 0) && lds_form_bytes(n4, rc.stack_depth, LDS_BLOCK) <= 160 * 1024;
            rc.lds_nodes = want_lds ? n4 : 0u;
            const int block = want_lds ? LDS_BLOCK : TRACE_BLOCK;
            const bool gen = s->flat.needs_general; // rare graph shapes: the instantiation that carries their code
            const void* kernel =
                want_lds ? (count ? (gen ? (const void*)trace_kernel_plain<R, true, LDS_BLOCK, true, true> : (const void*)trace_kernel_plain<R, true, LDS_BLOCK, true, false>)
                                  : (gen ? (const void*)trace_kernel_plain<R, false, LDS_BLOCK, true, true> : (const void*)trace_kernel_plain<R, false, LDS_BLOCK, true, false>))
                         : (count ? (gen ? (const void*)trace_kernel_plain<R, true, TRACE_BLOCK, false, true> : (const void*)trace_kernel_plain<R, true, TRACE_BLOCK, false, false>)
                                  : (gen ? (const void*)trace_kernel_plain<R, false, TRACE_BLOCK, false, true> : (const void*)trace_kernel_plain<R, false, TRACE_BLOCK, false, false>));
            const size_t lds_bytes = lds_form_bytes(want_lds ? n4 : 0u, rc.stack_depth, uint32_t(block));
            if (lds_bytes > 160 * 1024) { set_last_error("render: traversal stacks do not fit in LDS"); return RTTNW_ERR_UNSUPPORTED; }
            HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
            int blocks_per_cu = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kernel, block, lds_bytes));
            blocks_per_cu = std::max(1, std::min(blocks_per_cu, 8));
            const size_t waves_per_block = size_t(block) / 64;
            const size_t grid = std::max<size_t>(1, std::min<size_t>(size_t(d->num_cus) * blocks_per_cu, ((n_jobs + 63) / 64 + waves_per_block - 1) / waves_per_block));
            if (int g = grow_spill(grid * size_t(block))) return g;
            if (n_jobs > 0 && !prepare_only) {
                R bg0 = R(p->background[0]), bg1 = R(p->background[1]), bg2 = R(p->background[2]), tmin = R(p->t_min);
                R* part = (R*)d->partial;
                unsigned long long* jc = d->job_counter;
                SceneView<R> view = ds.view;
                CameraRec<R> camv = camr;
                int32_t* sp = (int32_t*)d->spill;
                void* args[] = {&view, &camv, &rc, &bg0, &bg1, &bg2, &tmin, &part, &jc, &dc, &sp};
                HIP_TRY(hipLaunchKernel(kernel, dim3(uint32_t(grid)), dim3(block), args, lds_bytes, stream));
            }
        } else if (streamf) {
            const bool gen = s->flat.needs_general;
            rc.lds_nodes = n4_all;
            const int sel = (count ? 4 : 0) | (gen ? 2 : 0) | (stream_hits_lds ? 1 : 0);
            typedef void (*StreamKernel)(SceneView<R>, CameraRec<R>, RenderConsts, R, R, R, R, R*, unsigned long long*, DeviceCounters*, R*, uint32_t*, uint32_t, int32_t*);
            static const StreamKernel kernels[8] = {
                trace_kernel_stream<R, false, STREAM_BLOCK, false, false>, trace_kernel_stream<R, false, STREAM_BLOCK, false, true>,
                trace_kernel_stream<R, false, STREAM_BLOCK, true, false>,  trace_kernel_stream<R, false, STREAM_BLOCK, true, true>,
                trace_kernel_stream<R, true, STREAM_BLOCK, false, false>,  trace_kernel_stream<R, true, STREAM_BLOCK, false, true>,
                trace_kernel_stream<R, true, STREAM_BLOCK, true, false>,   trace_kernel_stream<R, true, STREAM_BLOCK, true, true>};
            const StreamKernel kernel = kernels[sel];
            const size_t lds_bytes = stream_form_bytes(n4_all, rc.stack_depth, STREAM_BLOCK, sizeof(R), stream_hits_lds);
            HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)));
            constexpr size_t waves_per_block = STREAM_BLOCK / 64;
            const size_t waves_needed = (n_jobs + SLOTS_PER_WAVE - 1) / SLOTS_PER_WAVE;
            const size_t grid = std::max<size_t>(1, std::min<size_t>(size_t(d->num_cus), (waves_needed + waves_per_block - 1) / waves_per_block));
            const size_t n_slots = grid * waves_per_block * SLOTS_PER_WAVE;
            if (int g = grow(&d->pool_r, &d->pool_r_bytes, n_slots * PR_COUNT * sizeof(R))) return g;
            if (int g = grow(&d->pool_u, &d->pool_u_bytes, n_slots * PU_COUNT * sizeof(uint32_t))) return g;
            if (int g = grow_spill(grid * size_t(STREAM_BLOCK))) return g;
            if (n_jobs > 0 && !prepare_only) {
                hipLaunchKernelGGL(kernel, dim3(uint32_t(grid)), dim3(STREAM_BLOCK), lds_bytes, stream, ds.view, camr, rc, R(p->background[0]),
                                   R(p->background[1]), R(p->background[2]), R(p->t_min), (R*)d->partial, d->job_counter, dc, (R*)d->pool_r,
                                   (uint32_t*)d->pool_u, uint32_t(n_slots), (int32_t*)d->spill);
                HIP_TRY(hipGetLastError());
            }
        } else {
            const bool gen = s->flat.needs_general;
            auto kernel = count ? (gen ? trace_kernel<R, true, true> : trace_kernel<R, true, false>) : (gen ? trace_kernel<R, false, true> : trace_kernel<R, false, false>);
            const size_t lds_bytes = size_t(wave_lds_bytes<R>(rc.stack_depth)) * (TRACE_BLOCK / 64);
            size_t grid = 1;
            if (int g = persistent_grid((const void*)kernel, lds_bytes, (n_jobs + SLOTS_PER_WAVE - 1) / SLOTS_PER_WAVE, grid)) return g;
            const size_t n_slots = grid * (TRACE_BLOCK / 64) * SLOTS_PER_WAVE;
            if (int g = grow(&d->pool_r, &d->pool_r_bytes, n_slots * PR_COUNT * sizeof(R))) return g;
            if (int g = grow(&d->pool_u, &d->pool_u_bytes, n_slots * PU_COUNT * sizeof(uint32_t))) return g;
            if (int g = grow_spill(grid * size_t(TRACE_BLOCK))) return g;
            if (n_jobs > 0 && !prepare_only) {
                hipLaunchKernelGGL(kernel, dim3(uint32_t(grid)), dim3(TRACE_BLOCK), lds_bytes, stream, ds.view, camr, rc, R(p->background[0]),
                                   R(p->background[1]), R(p->background[2]), R(p->t_min), (R*)d->partial, d->job_counter, dc, (R*)d->pool_r,
                                   (uint32_t*)d->pool_u, uint32_t(n_slots), (int32_t*)d->spill);
                HIP_TRY(hipGetLastError());
            }
        }
        return 0;
    };
    if (stats && !prepare_only) HIP_TRY(hipEventRecord(d->ev0, stream));
    for (uint32_t k = 0; k < n_pass; ++k) {
        const uint32_t s0 = pass_begin(p->spp, n_pass, k), s1 = pass_begin(p->spp, n_pass, k + 1);
        rc.spp = s1 - s0;                          // this pass's samples; the kernels see a render of [sample_begin, +spp)
        rc.sample_begin = uint64_t(p->sample_begin) + s0;
        plan_chunks(rc, rc.spp, p->spp_chunk, uint64_t(L.n_tiles) * 64, 3 * sizeof(R)); // whole image: the same schedule on every rank
        if (!plan_jobs(rc)) { set_last_error("render: more than 2^32 jobs; use a larger spp_chunk"); return RTTNW_ERR_UNSUPPORTED; }
        if (int g = grow(&d->partial, &d->partial_bytes, std::max<size_t>(size_t(rc.jobs_per_chunk) * rc.n_chunks, 1) * 3 * sizeof(R))) return g;
        if (k && !prepare_only) HIP_TRY(hipMemsetAsync(d->job_counter, 0, sizeof(unsigned long long), stream)); // the job counter only: statistics add up
        if (int g = trace_pass()) return g;
        if (prepare_only) continue;
        if (stats && k + 1 == n_pass) HIP_TRY(hipEventRecord(d->ev1, stream));
        hipLaunchKernelGGL(resolve_kernel<R>, dim3((L.pixels_per_rank + 255) / 256), dim3(256), 0, stream, (const R*)d->partial,
                           (R*)d_packed, rc, L.pixels_per_rank, uint32_t(k == 0), uint32_t(k + 1 == n_pass), p->spp);
        HIP_TRY(hipGetLastError());
    }
    rc.spp = p->spp;
    if (prepare_only) return RTTNW_OK;

    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        if (sync_for_stats) {
            HIP_TRY(hipStreamSynchronize(stream));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, d->ev0, d->ev1));
            stats->kernel_ms = ms;
        }
        // samples traced by this rank: pixels of its tiles that lie inside the image
        uint64_t px_count = 0;
        for (uint32_t t = 0; t < rc.my_tiles; ++t) {
            uint32_t tx, ty;
            tile_unpermute(rc.tile_rank + t * rc.tile_world, rc.tiles_x, tx, ty);
            uint32_t w = std::min(8u, rc.width - tx * 8), h = std::min(8u, rc.height - ty * 8);
            px_count += uint64_t(w) * h;
        }
        stats->samples = px_count * rc.spp;
        if (count && sync_for_stats) {
            DeviceCounters hc;
            HIP_TRY(hipMemcpy(&hc, dc, sizeof(hc), hipMemcpyDeviceToHost));
            stats->rays = hc.rays; stats->nodes_visited = hc.nodes; stats->prims_tested = hc.prims; stats->texel_fetches = hc.texels;
            if (getenv("RTTNW_DEBUG_SCHED") && plain) {
                const double tot = double(hc.dbg[0] + hc.dbg[1] + hc.dbg[2] + hc.dbg[3]);
                fprintf(stderr, "[plain] wave clock: hand-out %.1f%%  begin %.1f%%  walk %.1f%%  shade %.1f%% (media + hit record %.1f%%, material %.1f%%)\n", 100 * hc.dbg[0] / tot,
                        100 * hc.dbg[1] / tot, 100 * hc.dbg[2] / tot, 100 * hc.dbg[3] / tot, 100 * hc.dbg[15] / tot, 100 * (hc.dbg[3] - hc.dbg[15]) / tot);
                fprintf(stderr, "[plain] walk: %.1f lockstep iterations/round (%.1f with node lanes, %.1f with leaf lanes); lanes served per iteration %.1f of 64\n",
                        double(hc.dbg[4]) / hc.dbg[9], double(hc.dbg[7]) / hc.dbg[9], double(hc.dbg[8]) / hc.dbg[9],
                        double(hc.dbg[5] + hc.dbg[6]) / hc.dbg[4]);
                fprintf(stderr, "[plain] walk clock: node steps %.1f%%, leaf steps %.1f%% of the walk\n", 100.0 * hc.dbg[13] / hc.dbg[2], 100.0 * hc.dbg[14] / hc.dbg[2]);
                for (uint32_t m = 1; m < 64; ++m)
                    if (hc.dbg[80 + m] * 200 > hc.dbg[8])
                        fprintf(stderr, "[plain]   leaf iterations serving {%s%s%s%s%s%s}: %.1f%% of them, %.1f%% of the leaf clock, %.0f clocks each\n", m & 1 ? "sphere " : "",
                                m & 2 ? "moving " : "", m & 4 ? "rect " : "", m & 8 ? "box " : "", m & 16 ? "instance " : "", m & 32 ? "empty " : "",
                                100.0 * hc.dbg[80 + m] / hc.dbg[8], 100.0 * hc.dbg[16 + m] / hc.dbg[14], double(hc.dbg[16 + m]) / hc.dbg[80 + m]);
                fprintf(stderr, "[plain]   node iterations: %.0f clocks each\n", double(hc.dbg[13]) / hc.dbg[7]);
                if (rc.profile == 3u) { // collect_counters = 3: distribution of walk lengths, in trips
                    fprintf(stderr, "[plain] trips per walk (lanes):");
                    for (int k = 0; k < 64; ++k) fprintf(stderr, " %llu", hc.dbg[16 + k]);
                    fprintf(stderr, "\n[plain] walks / mean trips by result (miss, sphere, moving, rect, box, -, in instance):");
                    for (int k = 0; k < 7; ++k) fprintf(stderr, " %llu / %.1f", hc.dbg[152 + k], hc.dbg[152 + k] ? double(hc.dbg[144 + k]) / hc.dbg[152 + k] : 0.0);
                    fprintf(stderr, "\n[plain] trips of the longest walk per round (waves):");
                    for (int k = 0; k < 64; ++k) fprintf(stderr, " %llu", hc.dbg[80 + k]);
                    fprintf(stderr, "\n");
                }
                fprintf(stderr, "[plain] node lanes per node iteration %.1f, leaf lanes per leaf iteration %.1f; rounds/sample %.2f, lanes alive per round %.1f; begin in %.0f%% of rounds, %.1f lanes each\n",
                        double(hc.dbg[5]) / hc.dbg[7], double(hc.dbg[6]) / hc.dbg[8], double(hc.dbg[9]) * 64 / stats->samples,
                        double(hc.dbg[10]) / hc.dbg[9], 100.0 * hc.dbg[11] / hc.dbg[9], hc.dbg[11] ? double(hc.dbg[12]) / hc.dbg[11] : 0.0);
            } else if (getenv("RTTNW_DEBUG_SCHED")) {
                const double w64 = double(stats->samples) / 64.0;
                fprintf(stderr, "[decoupled] bursts/64smp %.1f  shades/64smp %.2f (lanes %.1f)  refills/64smp %.1f (lanes %.1f)\n", hc.dbg[8] / w64, hc.dbg[9] / w64,
                        hc.dbg[9] ? double(hc.dbg[10]) / hc.dbg[9] : 0.0, hc.dbg[11] / w64, hc.dbg[11] ? double(hc.dbg[12]) / hc.dbg[11] : 0.0);
                if (hc.dbg[13]) { // the stream form: its walk trips and the wave clock of its phases
                    const double tot = double(hc.dbg[0] + hc.dbg[1] + hc.dbg[2] + hc.dbg[3] + hc.dbg[4]);
                    fprintf(stderr, "[stream] trips/64smp %.1f, lanes walking per trip %.1f\n", hc.dbg[13] / w64, double(hc.dbg[14]) / hc.dbg[13]);
                    fprintf(stderr, "[stream] wave clock: state load %.1f%%  path_shade %.1f%%  jobs + new paths %.1f%%  store + hand-over %.1f%%  walk %.1f%%;  per shade batch (clocks): %.0f %.0f %.0f %.0f, per trip %.0f\n",
                            100 * hc.dbg[0] / tot, 100 * hc.dbg[1] / tot, 100 * hc.dbg[2] / tot, 100 * hc.dbg[3] / tot, 100 * hc.dbg[4] / tot, 16.0 * hc.dbg[0] / hc.dbg[9], 16.0 * hc.dbg[1] / hc.dbg[9],
                            16.0 * hc.dbg[2] / hc.dbg[9], 16.0 * hc.dbg[3] / hc.dbg[9], 16.0 * hc.dbg[4] / hc.dbg[13]);
                }
            }
        }
        stats->n_nodes = uint32_t(s->flat.nodes4.size());
        stats->n_prims = s->flat.n_prims_in_bvh;
        stats->scene_bytes = uint32_t(std::min<size_t>(ds.bytes, 0xFFFFFFFFu));
        stats->reserved = plain ? 0u : 1u; // which kernel form ran: 0 lane-owns-path, 1 decoupled
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
    DevBuf<double> d_out; // released on every exit path
    DevBuf<int32_t> d_n;
    if (int r = d_out.upload(std::vector<double>(size_t(max_out) * PROBE_STRIDE + 4, 0.0))) return r;
    if (int r = d_n.upload(std::vector<int32_t>(1, 0))) { d_out.release(); return r; }
    struct Release { DevBuf<double>& a; DevBuf<int32_t>& b; ~Release() { a.release(); b.release(); } } release{d_out, d_n};
    DevBuf<int32_t> d_spill;
    if (int r = d_spill.upload(std::vector<int32_t>(std::max<size_t>(rc.stack_depth, 1), 0))) { d_out.release(); d_n.release(); return r; }
    struct Release2 { DevBuf<int32_t>& a; ~Release2() { a.release(); } } release2{d_spill};
    const size_t lds = size_t(LDS_STACK_ENTRIES + 1) * 64 * sizeof(int32_t);
    hipLaunchKernelGGL(probe_path_kernel<R>, dim3(1), dim3(64), lds, 0, ds.view, narrow_camera<R>(cam64), rc, R(p->t_min), px, row,
                       sample, d_out.p, max_out, d_n.p, d_spill.p);
    HIP_TRY(hipGetLastError());
    int32_t n = 0;
    HIP_TRY(hipMemcpy(&n, d_n.p, sizeof(n), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out, d_out.p, size_t(n) * PROBE_STRIDE * sizeof(double), hipMemcpyDeviceToHost));
    // radiance of the sample (path_step loop) is returned after the last possible bounce record
    HIP_TRY(hipMemcpy(out + size_t(max_out) * PROBE_STRIDE, d_out.p + size_t(max_out) * PROBE_STRIDE, 4 * sizeof(double), hipMemcpyDeviceToHost));
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
    out->n_nodes = uint32_t(s->flat.nodes4.size());
    out->n_prims = s->flat.n_prims_in_bvh;
    const auto& f = s->flat;
    size_t b32 = f.nodes4.size() * sizeof(rt::Bvh4Node) + f.spheres.size() * sizeof(rt::SphereRec<float>) +
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
    return p->precision == RTTNW_F32 ? rt::render_tiles_t<float>(s, s->device, cam, p, d_packed, stream, stats)
                                     : rt::render_tiles_t<double>(s, s->device, cam, p, d_packed, stream, stats);
}

int rttnw_untile_device(uint32_t width, uint32_t height, uint32_t world, uint32_t precision, const void* d_gathered,
                        void* d_linear_rgb, uint8_t* d_rgba8, void* hip_stream) {
    if (!width || !height || !world || !d_gathered || (precision != RTTNW_F32 && precision != RTTNW_F64)) {
        rt::set_last_error("untile_device: bad arguments");
        return RTTNW_ERR_INVALID;
    }
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
struct MultiComms { // one communicator set per distinct list of devices, kept for the life of the process
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
};
static std::vector<MultiComms*> g_comms;

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
        int rc = p.precision == RTTNW_F32 ? render_tiles_t<float>(s, st[r], cam, &pr, dst, st[r]->stream, nullptr, false, true)
                                          : render_tiles_t<double>(s, st[r], cam, &pr, dst, st[r]->stream, nullptr, false, true);
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
        int rc = p.precision == RTTNW_F32 ? render_tiles_t<float>(s, d, cam, &pr, dst, d->stream, stats ? &stats[r] : nullptr, false)
                                          : render_tiles_t<double>(s, d, cam, &pr, dst, d->stream, stats ? &stats[r] : nullptr, false);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(ev[2 * r + 1], d->stream));
    }

    // ---- gather to the root: a device-to-device copy for ranks on the root's device, RCCL send/recv over xGMI for the others.
    // RTTNW_MULTI_FORCE_RCCL=1 (tests): the ranks on the root's device travel through RCCL too — a grouped ncclSend / ncclRecv
    // of the root to itself — so that the dlopen'ed entry points, the communicator set-up, the stream ordering and the
    // error paths run on a box with ONE GPU as well.
    const char* force_env = getenv("RTTNW_MULTI_FORCE_RCCL");
    const bool force_rccl = force_env && force_env[0] == '1';
    if (distinct.size() > 1 || force_rccl) {
        if (!g_rccl.load(err)) { set_last_error("render_multi: " + err); return RTTNW_ERR_HIP; }
        MultiComms* mc = nullptr;
        for (MultiComms* c : g_comms)
            if (c->devices == distinct) mc = c;
        if (!mc) {
            mc = new MultiComms();
            mc->devices = distinct;
            mc->comms.resize(distinct.size());
            ncclResult_t nr = g_rccl.CommInitAll(mc->comms.data(), int(distinct.size()), distinct.data());
            if (nr != ncclSuccess) { set_last_error(std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(nr)); delete mc; return RTTNW_ERR_HIP; }
            g_comms.push_back(mc); // kept for the life of the process: communicator set-up costs ~100 ms; freed by the OS at exit
            if (getenv("RTTNW_DEBUG_MULTI")) fprintf(stderr, "[render_multi] RCCL communicators over %zu device(s)\n", distinct.size());
        }
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
        if (getenv("RTTNW_DEBUG_MULTI")) fprintf(stderr, "[render_multi] %u rank buffer(s) of %zu bytes through ncclSend / ncclRecv\n", n_sent, chunk);
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
    HIP_TRY(hipSetDevice(root->device));
    if (out_rgba8) HIP_TRY(hipMemcpy(out_rgba8, root->rgba, npx * 4, hipMemcpyDeviceToHost));
    if (out_linear_rgb) {
        if (p.precision == RTTNW_F64) {
            HIP_TRY(hipMemcpy(out_linear_rgb, root->linear, npx * 3 * sizeof(double), hipMemcpyDeviceToHost));
        } else {
            std::vector<float> tmp(npx * 3);
            HIP_TRY(hipMemcpy(tmp.data(), root->linear, npx * 3 * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < npx * 3; ++i) out_linear_rgb[i] = double(tmp[i]);
        }
    }
    return RTTNW_OK;
}
