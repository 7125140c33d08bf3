// render_f64_wave.hip — the decoupled kernel's contracted f64 instantiations (trace_kernels.hpp trace_kernel<double, *, *>), in a translation
// unit of their own because they want the pre-RA machine scheduler that render_f64.hip's flags turn off for the lane-owns-path kernel
// (Makefile HIPFLAGS_F64_WAVE; spheres_1m f64 280 -> 308 Msamples/s).  render_f64.hip launches them.
#include "render_tiles.hpp"

namespace rt {
inline namespace RT_ARITH_NS {
#define RT_WAVE_INST(COUNT, GENERAL)                                                                                                              \
    template __global__ void trace_kernel<double, COUNT, GENERAL>(SceneView<double>, CameraRec<double>, RenderConsts, double, double, double, double, \
                                                                  double*, unsigned long long*, DeviceCounters*, double*, uint32_t*, uint32_t, int32_t*);
RT_WAVE_INST(false, SHAPES_FAST) RT_WAVE_INST(false, SHAPES_GENERAL) RT_WAVE_INST(false, SHAPES_NONE) RT_WAVE_INST(false, SHAPES_NONE_NT) RT_WAVE_INST(true, SHAPES_FAST) RT_WAVE_INST(true, SHAPES_GENERAL)
#undef RT_WAVE_INST
} // namespace RT_ARITH_NS
} // namespace rt
