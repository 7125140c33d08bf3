// render_f64.hip — the F64 kernels (the reference's arithmetic) and their launch code: one translation unit per precision
// (render_common.hpp); built with its own flags (Makefile HIPFLAGS_F64).
#define RT_F64_WAVE_KERNELS_ELSEWHERE 1 // the decoupled kernel's instantiations: render_f64_wave.hip (its own scheduler flags)
#include "render_tiles.hpp"

namespace rt {
inline namespace RT_ARITH_NS {
RT_INSTANTIATE_PRECISION(double)
} // namespace RT_ARITH_NS
} // namespace rt
