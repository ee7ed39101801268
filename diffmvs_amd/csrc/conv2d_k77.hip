// conv2d.hip, the tiled kernel instantiated for the 7x7 layer and the 3x3 stride-2 layers (see conv2d_tiled.h)
#include "conv2d_tiled.h"

namespace dmvs_detail {
int launch_conv2d_771(const dmvs_conv2d_desc& d, hipStream_t st) { return launch_conv2d<7, 7, 1>(d, st); }
int launch_conv2d_332(const dmvs_conv2d_desc& d, hipStream_t st) { return launch_conv2d<3, 3, 2>(d, st); }
}  // namespace dmvs_detail
