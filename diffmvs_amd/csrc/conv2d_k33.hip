// conv2d.hip, the tiled kernel instantiated for the 3x3 stride-1 layers (see conv2d_tiled.h)
#include "conv2d_tiled.h"

namespace dmvs_detail {
int launch_conv2d_331(const dmvs_conv2d_desc& d, hipStream_t st) { return launch_conv2d<3, 3, 1>(d, st); }
}  // namespace dmvs_detail
