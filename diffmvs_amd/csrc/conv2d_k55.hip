// conv2d.hip, the tiled kernel instantiated for the 5x5 layers (stride 2, and stride 1 = their input gradient) (see conv2d_tiled.h)
#include "conv2d_tiled.h"

namespace dmvs_detail {
int launch_conv2d_552(const dmvs_conv2d_desc& d, hipStream_t st) { return launch_conv2d<5, 5, 2>(d, st); }
int launch_conv2d_551(const dmvs_conv2d_desc& d, hipStream_t st) { return launch_conv2d<5, 5, 1>(d, st); }
}  // namespace dmvs_detail
