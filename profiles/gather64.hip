// gather64.hip — do 64-byte (or 32-byte) records cost less than 128-byte records when gathered at random?  (Would a compressed BVH node
// halve the walk's traffic?  No: L2 fills whole 128-byte lines — profiles/r03/README.md.)
//   hipcc --offload-arch=gfx950 -O3 -o gather64 profiles/gather64.hip && ./gather64
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./gather64
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int BYTES, int TAG> __global__ void gather(const int4* __restrict__ table, unsigned n_mask, int rounds, int* __restrict__ sink) {
    unsigned idx = hash32(blockIdx.x * blockDim.x + threadIdx.x + 0x9e3779b9u * TAG);
    int acc = 0;
    for (int r = 0; r < rounds; ++r) {
        const int4* rec = table + size_t(idx & n_mask) * (BYTES / 16);
#pragma unroll
        for (int q = 0; q < (BYTES == 128 ? 7 : BYTES / 16); ++q) { const int4 v = rec[q]; acc += v.x ^ v.y ^ v.z ^ v.w; }
        idx = hash32(idx + unsigned(acc) + unsigned(r));
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int BYTES, int TAG> int run(const int4* t, size_t n_records, const char* what, int* sink) {
    const int blocks = 256 * 12, threads = 256, rounds = 256;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((gather<BYTES, TAG>), dim3(blocks), dim3(threads), 0, 0, t, unsigned(n_records - 1), rounds, sink);
        CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double visits = double(blocks) * threads * rounds;
        if (rep == 2) printf("%-44s %4d-B records x %8zu (%6.1f MiB): %.3f ms, %.2f G visits/s, %.2f TB/s of record bytes\n", what, BYTES, n_records, n_records * double(BYTES) / 1048576, ms, visits / ms * 1e-6, visits * BYTES / ms * 1e-9);
    }
    return 0;
}
int main() {
    int4* t; int* sink; const size_t big = size_t(2) << 30;
    CK(hipMalloc(&t, big)); CK(hipMalloc(&sink, 4)); CK(hipMemset(t, 1, big));
    // same NUMBER of records (the tree has as many nodes either way)
    run<128, 1>(t, 1u << 19, "512k records", sink); run<64, 2>(t, 1u << 19, "512k records", sink); run<32, 3>(t, 1u << 19, "512k records", sink);
    run<128, 4>(t, 1u << 24, "16M records (HBM)", sink); run<64, 5>(t, 1u << 24, "16M records (HBM)", sink); run<32, 6>(t, 1u << 24, "16M records (HBM)", sink);
    run<128, 7>(t, 1u << 15, "32k records (L2)", sink); run<64, 8>(t, 1u << 15, "32k records (L2)", sink);
    return 0;
}
