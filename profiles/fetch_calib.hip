// fetch_calib.hip — what does rocprofv3's FETCH_SIZE report for the access pattern of the BVH walk?
//
// MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports exactly HALF the bytes of a wide coalesced streaming read
// (TCC_EA0_RDREQ counts 128-B requests at 64 B) and says other access shapes must be calibrated on a known byte count.
// The walk's shape: every LANE reads 7 x 16 B of ONE 128-byte record (a Bvh4Node, rt_types.hpp) chosen independently of
// its neighbours.  This program does exactly that from a table of N records with pseudo-random indices, so that the bytes
// that must leave the memory side are known: every touched record is one 128-B line, fetched once if the table is far
// larger than the caches (2 GiB) — lanes x rounds x 128 B — or served by the Infinity Cache if it is small (64 MiB).
//
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib profiles/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
// prints, per kernel name, the expected byte count; profiles/r03/README.md holds the comparison with the counter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct alignas(16) Rec { int4 q[8]; }; // 128 B

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// every lane: `rounds` dependent gathers of 7 x 16 B out of one random record (the walk's node fetch)
template <int TAG> __global__ void gather_records(const Rec* __restrict__ table, unsigned n_mask, int rounds, int* __restrict__ sink) {
    unsigned idx = hash32(blockIdx.x * blockDim.x + threadIdx.x + 0x9e3779b9u * TAG);
    int acc = 0;
    for (int r = 0; r < rounds; ++r) {
        const int4* rec = table[idx & n_mask].q;
#pragma unroll
        for (int q = 0; q < 7; ++q) { const int4 v = rec[q]; acc += v.x ^ v.y ^ v.z ^ v.w; }
        idx = hash32(idx + unsigned(acc) + unsigned(r)); // dependent: the next record is known only now, like a child index
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}
// the guide's calibration case for comparison: a wide coalesced streaming read, 16 B per lane
__global__ void stream_read(const int4* __restrict__ src, size_t n16, int* __restrict__ sink) {
    int acc = 0;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n16; i += size_t(gridDim.x) * blockDim.x) { const int4 v = src[i]; acc += v.x ^ v.w; }
    if (acc == 0x7fffffff) sink[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const size_t big = size_t(2) << 30, small = size_t(64) << 20;
    Rec* t_big; Rec* t_small; int* sink;
    CK(hipMalloc(&t_big, big)); CK(hipMalloc(&t_small, small)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(t_big, 1, big)); CK(hipMemset(t_small, 1, small));
    const int blocks = 256 * 8, threads = 256, rounds = 64;
    const double lanes = double(blocks) * threads;
    // 1: records of a 2 GiB table (beyond the 256 MiB Infinity Cache): every visit is a 128-B line from HBM
    hipLaunchKernelGGL(gather_records<1>, dim3(blocks), dim3(threads), 0, 0, t_big, unsigned(big / sizeof(Rec) - 1), rounds, sink);
    CK(hipDeviceSynchronize());
    printf("gather_records<1> table 2 GiB : %.0f visits, %.6g bytes if every visit fetches its 128-B line once\n", lanes * rounds, lanes * rounds * 128.0);
    // 2: records of a 64 MiB table (Infinity-Cache resident after the first touches, like spheres_1m's 57 MB of nodes)
    hipLaunchKernelGGL(gather_records<2>, dim3(blocks), dim3(threads), 0, 0, t_small, unsigned(small / sizeof(Rec) - 1), rounds, sink);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(gather_records<3>, dim3(blocks), dim3(threads), 0, 0, t_small, unsigned(small / sizeof(Rec) - 1), rounds, sink);
    CK(hipDeviceSynchronize());
    printf("gather_records<2>, <3> table 64 MiB (second launch warm): %.0f visits, %.6g bytes past L2 if none hit L2 (an XCD's 4 MiB L2 holds 1/16 of the table)\n", lanes * rounds, lanes * rounds * 128.0);
    // 3: the guide's case
    hipLaunchKernelGGL(stream_read, dim3(blocks), dim3(threads), 0, 0, (const int4*)t_big, big / 16, sink);
    CK(hipDeviceSynchronize());
    printf("stream_read 2 GiB : %.6g bytes\n", double(big));
    return 0;
}
