// conv2d.hip, the tiled kernel instantiated for the 1x5 / 5x1 GRU layers and the tiled 1x1 form (see conv2d_tiled.h)
#include "conv2d_tiled.h"

namespace dmvs_detail {
int launch_conv2d_151(const dmvs_conv2d_desc& d, hipStream_t st) { return launch_conv2d<1, 5, 1>(d, st); }
int launch_conv2d_511(const dmvs_conv2d_desc& d, hipStream_t st) { return launch_conv2d<5, 1, 1>(d, st); }
int launch_conv2d_111(const dmvs_conv2d_desc& d, hipStream_t st) { return launch_conv2d<1, 1, 1>(d, st); }
}  // namespace dmvs_detail
