// render_f64_strict.hip — the F64 kernels a second time, in the IEEE-strict build of the arithmetic (rt_core.hpp RT_STRICT_F64,
// namespace rt::ieee_strict; precision RTTNW_F64_STRICT): -ffp-contract=off, every f64 quotient an IEEE division.  The same
// operations in the same order as the reference's Rust (rustc contracts nothing), so every path decision equals the CPU
// reference's — the mode the full-size parity tests pin bit for bit, and a caller's choice when reproducibility against the CPU
// build matters more than the 5-10 % the contracted build gains.
#define RT_STRICT_F64 1
#if defined(__FAST_MATH__)
#error "render_f64_strict.hip must not be built with fast-math flags: its results are the CPU reference's bit for bit"
#endif
#pragma clang fp contract(off) // (beside the Makefile's trailing -ffp-contract=off: honoured should the unit ever be built under fast-honor-pragmas)
#include "render_tiles.hpp"

namespace rt {
inline namespace RT_ARITH_NS {
template int render_tiles_t<double>(::rttnw_scene*, DeviceState*, const rttnw_camera_desc*, const rttnw_params*, void*, hipStream_t, rttnw_stats*, bool, bool);
template int probe_path_t<double>(::rttnw_scene*, const rttnw_camera_desc*, const rttnw_params*, uint32_t, uint32_t, uint32_t, double*, uint32_t);
} // namespace RT_ARITH_NS
} // namespace rt
