// render_f32.hip — the F32 (throughput) kernels and their launch code: one translation unit per precision (render_common.hpp).
#include "render_tiles.hpp"

namespace rt {
inline namespace RT_ARITH_NS {
RT_INSTANTIATE_PRECISION(float)
} // namespace RT_ARITH_NS
} // namespace rt
