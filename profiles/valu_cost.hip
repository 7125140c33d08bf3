// valu_cost.hip — what a wave64 vector instruction costs a gfx950 SIMD in ISSUE cycles, by class: the prices bench.py's "valu" roofline
// puts on the SQ_INSTS_VALU_* class counters (profiles/r04/valu_cost.txt is this program's output on an MI355X).
//
// Every wave runs a loop of 32 INDEPENDENT instructions of one kind (eight accumulator chains, so no chain is a latency bound at 8
// waves per SIMD) and reads s_memtime (shader cycles) around it; with W waves resident on each SIMD the SIMD's cost per instruction is
//   elapsed cycles x 1 / (W x instructions per wave)      — reported for W = 1, 4, 8.
// Build / run:  hipcc -O3 --offload-arch=gfx950 profiles/valu_cost.hip -o /tmp/valu_cost && /tmp/valu_cost
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND> __device__ __forceinline__ void body(float (&f)[8], double (&d)[8], unsigned (&u)[8], unsigned long long (&q)[8]) {
#define ONE(i)                                                                                                                        \
    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));                                                     \
    if constexpr (KIND == 1) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[i]));                                                     \
    if constexpr (KIND == 2) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[i]));                                                         \
    if constexpr (KIND == 3) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[i]));                                                         \
    if constexpr (KIND == 4) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(u[i]));                                                      \
    if constexpr (KIND == 5) asm volatile("v_mul_hi_u32 %0, %0, %0" : "+v"(u[i]));                                                      \
    if constexpr (KIND == 6) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(q[i]) : "v"(u[i]) : "vcc");                        \
    if constexpr (KIND == 7) asm volatile("v_xor_b32 %0, %0, %0" : "+v"(u[i]));                                                         \
    if constexpr (KIND == 8) asm volatile("v_cndmask_b32 %0, %0, %0, s[20:21]" : "+v"(u[i]));                                                \
    if constexpr (KIND == 9) asm volatile("v_lshrrev_b64 %0, 7, %0" : "+v"(q[i]));                                                      \
    if constexpr (KIND == 10) asm volatile("v_cmp_lt_f64 vcc, %0, %0" : : "v"(d[i]) : "vcc");                                           \
    if constexpr (KIND == 11) asm volatile("v_cmp_lt_f32 vcc, %0, %0" : : "v"(f[i]) : "vcc");                                           \
    if constexpr (KIND == 12) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));                                                            \
    if constexpr (KIND == 13) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));                                                            \
    if constexpr (KIND == 14) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f[i]) : "v"(u[i]));                                         \
    if constexpr (KIND == 15) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(f[i]));                                                   \
    if constexpr (KIND == 16) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(u[i]));                                            \
    if constexpr (KIND == 17) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[i]));                                                           \
    if constexpr (KIND == 18) asm volatile("v_add_co_u32 %0, vcc, %0, %0" : "+v"(u[i]) : : "vcc");                                      \
    if constexpr (KIND == 19) asm volatile("v_bcnt_u32_b32 %0, %0, %0" : "+v"(u[i]));                                                   \
    if constexpr (KIND == 20) asm volatile("v_fma_mix_f32 %0, %1, %0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(f[i]) : "v"(u[i]));   \
    if constexpr (KIND == 21) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f[i]) : "v"(u[i]));                                            \
    if constexpr (KIND == 22) asm volatile("v_min_f32 %0, %0, %0" : "+v"(f[i]));                                                        \
    if constexpr (KIND == 23) asm volatile("v_mov_b32 %0, %1" : "=v"(f[i]) : "v"(u[i]));                                                \
    if constexpr (KIND == 24) asm volatile("v_perm_b32 %0, %0, %1, s20" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));                            \
    if constexpr (KIND == 25) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));                           \
    if constexpr (KIND == 26) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));                                  \
    if constexpr (KIND == 27) asm volatile("v_lshl_or_b32 %0, %0, 7, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));                           \
    if constexpr (KIND == 28) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(u[i]));                                                      \
    if constexpr (KIND == 29) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));                                            \
    if constexpr (KIND == 30) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));                                            \
    if constexpr (KIND == 31) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(f[i]));                                                        \
    if constexpr (KIND == 32) asm volatile("v_lshlrev_b32 %0, 7, %0" : "+v"(u[i]));                                                     \
    if constexpr (KIND == 33) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(u[i]));                                               \
    if constexpr (KIND == 34) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f[i]) : "v"(u[i]));                                         \
    if constexpr (KIND == 35) asm volatile("v_add_u32 %0, %0, %0" : "+v"(u[i]));                                                        \
    if constexpr (KIND == 36) asm volatile("v_lshl_add_u32 %0, %0, 2, %0" : "+v"(u[i]));                                                \
    if constexpr (KIND == 37) asm volatile("v_div_scale_f64 %0, vcc, %0, %0, %0" : "+v"(d[i]) : : "vcc");                               \
    if constexpr (KIND == 38) asm volatile("v_div_fmas_f64 %0, %0, %0, %0" : "+v"(d[i]) : : "vcc");                                     \
    if constexpr (KIND == 39) asm volatile("v_div_fixup_f64 %0, %0, %0, %0" : "+v"(d[i]));                                              \
    if constexpr (KIND == 40) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(u[i]));                                            \
    if constexpr (KIND == 41) asm volatile("v_cmp_lt_u32 vcc, %0, %0" : : "v"(u[i]) : "vcc");                                           \
    if constexpr (KIND == 42) asm volatile("v_and_b32 %0, %0, %0" : "+v"(u[i]));                                                        \
    if constexpr (KIND == 43) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));                                                           \
    if constexpr (KIND == 44) asm volatile("v_fmac_f32 %0, %0, %0" : "+v"(f[i]));                                                       \
    if constexpr (KIND == 45) asm volatile("v_mad_u32_u24 %0, %0, %0, %0" : "+v"(u[i]));
    REP8(ONE)
#undef ONE
}

template <int KIND> __global__ void cost_kernel(int iters, unsigned long long* cycles, float* sink) {
    float f[8];
    double d[8];
    unsigned u[8];
    unsigned long long q[8];
    for (int i = 0; i < 8; ++i) { f[i] = 1.0f + threadIdx.x * 1e-3f + i; d[i] = 1.0 + threadIdx.x * 1e-3 + i; u[i] = threadIdx.x * 2654435761u + i; q[i] = u[i] * 0x9E3779B97F4A7C15ull; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        body<KIND>(f, d, u, q);
        body<KIND>(f, d, u, q);
        body<KIND>(f, d, u, q);
        body<KIND>(f, d, u, q);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i] + float(d[i]) + float(u[i]) + float(q[i]);
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (s == 123.456f) sink[0] = s;
}

static double g_ticks_per_ns = 0; // s_memtime ticks per nanosecond of the last run (calibration against hipEvents)
template <int KIND> double run(const char* name, int waves_per_simd) {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, iters = 4096;
    const int block = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;          // 4 SIMDs x waves x 64 lanes per CU
    const int blocks_per_cu = 256 * waves_per_simd / block;
    const int grid = cus * blocks_per_cu;
    unsigned long long* d_cycles;
    float* d_sink;
    (void)hipMalloc(&d_cycles, grid * sizeof(unsigned long long));
    (void)hipMalloc(&d_sink, 4);
    hipLaunchKernelGGL(cost_kernel<KIND>, dim3(grid), dim3(block), 0, 0, 16, d_cycles, d_sink); // warm-up
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(cost_kernel<KIND>, dim3(grid), dim3(block), 0, 0, iters, d_cycles, d_sink);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    (void)hipMemcpy(h.data(), d_cycles, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : h) mean += double(c);
    mean /= grid;
    g_ticks_per_ns = mean / (double(ms) * 1e6); // (the kernel is one loop: its wall time ~ a wave's ticks)
    (void)hipFree(d_cycles);
    (void)hipFree(d_sink);
    return mean / (double(waves_per_simd) * iters * 32.0); // SIMD cycles per wave64 instruction
}

#define ROW(K, NAME) { const double a = run<K>(NAME, 1), b = run<K>(NAME, 4), c = run<K>(NAME, 8); std::printf("%-18s %8.2f %8.2f %8.2f   (%.3f ticks/ns at W=8 -> %.2f ns per instruction and SIMD)\n", NAME, a, b, c, g_ticks_per_ns, c / g_ticks_per_ns); }

// "pmc" mode (profiles/valu_cost_cycles.sh runs it under rocprofv3 --pmc GRBM_GUI_ACTIVE): ONE long launch per kind at 8 waves per SIMD; the
// script divides the launch's busy cycles by the instructions a SIMD issued: cycles per instruction from the counter the kernels' own
// PMC summaries use.
template <int KIND> void one_long_launch() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    unsigned long long* d_cycles;
    float* d_sink;
    const int grid = prop.multiProcessorCount * 2;
    (void)hipMalloc(&d_cycles, grid * sizeof(unsigned long long));
    (void)hipMalloc(&d_sink, 4);
    hipLaunchKernelGGL(cost_kernel<KIND>, dim3(grid), dim3(1024), 0, 0, 32768, d_cycles, d_sink); // 8 waves per SIMD x 32768 iterations x 32 instructions
    (void)hipDeviceSynchronize();
    (void)hipFree(d_cycles);
    (void)hipFree(d_sink);
}
#define LONG(K) one_long_launch<K>();

int main(int argc, char** argv) {
    if (argc > 1) {
        LONG(0) LONG(15) LONG(11) LONG(14) LONG(13) LONG(7) LONG(8) LONG(18) LONG(19) LONG(4) LONG(5) LONG(6) LONG(9) LONG(1) LONG(2) LONG(3) LONG(10) LONG(16) LONG(12) LONG(17) LONG(20) LONG(21) LONG(22) LONG(23)
        return 0;
    }
    std::printf("# SIMD issue cycles per wave64 instruction (s_memtime ticks), W waves per SIMD: W=1, W=4, W=8\n");
    ROW(0, "v_fma_f32") ROW(15, "v_max3_f32") ROW(11, "v_cmp_lt_f32") ROW(14, "v_cvt_f32_ubyte1") ROW(13, "v_rcp_f32")
    ROW(7, "v_xor_b32") ROW(8, "v_cndmask_b32") ROW(18, "v_add_co_u32") ROW(19, "v_bcnt_u32_b32") ROW(4, "v_mul_lo_u32") ROW(5, "v_mul_hi_u32") ROW(6, "v_mad_u64_u32") ROW(9, "v_lshrrev_b64")
    ROW(1, "v_fma_f64") ROW(2, "v_add_f64") ROW(3, "v_mul_f64") ROW(10, "v_cmp_lt_f64") ROW(16, "v_cvt_f64_u32") ROW(12, "v_rcp_f64") ROW(17, "v_sqrt_f64") ROW(20, "v_fma_mix_f32") ROW(21, "v_cvt_f32_f16") ROW(22, "v_min_f32") ROW(23, "v_mov_b32")
    ROW(24, "v_perm_b32") ROW(25, "v_and_or_b32") ROW(26, "v_min_u32") ROW(27, "v_lshl_or_b32") ROW(28, "v_bfe_u32") ROW(29, "v_cvt_f32_f64") ROW(30, "v_cvt_f64_f32") ROW(31, "v_mul_f32")
    ROW(32, "v_lshlrev_b32") ROW(33, "v_cndmask_b32 vcc") ROW(34, "v_cvt_f32_ubyte0") ROW(35, "v_add_u32") ROW(36, "v_lshl_add_u32") ROW(37, "v_div_scale_f64") ROW(38, "v_div_fmas_f64")
    ROW(39, "v_div_fixup_f64") ROW(40, "v_cvt_f32_u32") ROW(41, "v_cmp_lt_u32") ROW(42, "v_and_b32") ROW(43, "v_sqrt_f32") ROW(44, "v_fmac_f32") ROW(45, "v_mad_u32_u24")
    return 0;
}
