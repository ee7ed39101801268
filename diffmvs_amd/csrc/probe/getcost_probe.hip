// libdmvs_probe.so, part 1: the memory-system ceiling probe of GetCost (include/dmvs_probe.h).  A bench-only library: measurement
// infrastructure, never loaded by the depth-estimation path.
//
// The kernel is getcost_quad_body of warp_quad_core.h -- the product's hypotheses, projection, per-pixel texel masks, bit scans, addresses
// and loads, from the same quads in the same order -- with a texel Body that only WAITS for the loaded registers.  What it takes per launch
// is what the L1 / L2 / HBM path needs for the product's own line-request stream; the product's distance to it is the arithmetic that does
// not hide under the misses.  (Rounds 4-5 built this as -DDMVS_GC_EXP=4 variants of warp_quad.hip; it is its own translation unit now.)
#include "warp_quad_core.h"

#include "dmvs_probe.h"

namespace {

struct QuadLoadsOnly {
    static constexpr bool kPairUnits = false;
    template <int C, int FT, int NH, int TPT>
    static __device__ __forceinline__ void texels(const Feat<C, FT> (&t)[TPT], const bool (&)[TPT], const float (&)[TPT], const float (&)[TPT],
                                                  const float (&)[(NH + 3) / 4], const float (&)[(NH + 3) / 4], const float (&)[C / 4], float,
                                                  float (&)[NH]) {
#pragma unroll
        for (int i = 0; i < TPT; ++i)
#pragma unroll
            for (int j = 0; j < Feat<C, FT>::NW; ++j) asm volatile("" ::"v"(t[i].w[j]));      // the loads must land; nothing is computed from them
    }
};

template <int C, int N, int TPT, int FT>
__global__ void __launch_bounds__(GC_BLOCK) getcost_loads_probe_kernel(const dmvs_getcost_desc d) {
    getcost_quad_body<QuadLoadsOnly, C, N, TPT, FT>(d);
}

// the same stream with the texel PAIR (x, x + 1) as the load unit (16-bit features with 16 channels: warp_quad_core.h, quad_accumulate)
struct QuadPairLoadsOnly : QuadLoadsOnly {
    static constexpr bool kPairUnits = true;
};
template <int N, int TPT, int FT>
__global__ void __launch_bounds__(GC_BLOCK) getcost_pair_loads_probe_kernel(const dmvs_getcost_desc d) {
    getcost_quad_body<QuadPairLoadsOnly, 16, N, TPT, FT>(d);
}

template <int FT>
int launch_probe(const dmvs_getcost_desc& d, dim3 grid, dim3 block, hipStream_t st) {
#define DMVS_GCP(CC, NN) hipLaunchKernelGGL((getcost_loads_probe_kernel<CC, NN, QUAD_TPT, FT>), grid, block, 0, st, d)
    if (d.C == 32 && d.n == 6) DMVS_GCP(32, 6);
    else if (d.C == 32 && d.n == 4) DMVS_GCP(32, 4);
    else if (d.C == 16 && d.n == 4) DMVS_GCP(16, 4);
    else return DMVS_EINVAL;
#undef DMVS_GCP
    return dmvs_launch_status();
}

}  // namespace

extern "C" int dmvs_probe_abi_version(void) { return DMVS_PROBE_ABI_VERSION; }

extern "C" int dmvs_probe_getcost_loads_f32(const dmvs_getcost_desc* dp, void* stream) {
    if (!dp || !getcost_desc_ok(*dp)) return DMVS_EINVAL;
    const dmvs_getcost_desc& d = *dp;
    const dim3 grid = getcost_grid(d), block(GC_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    if (d.feat_dtype == DMVS_DTYPE_BF16) return launch_probe<DMVS_DTYPE_BF16>(d, grid, block, st);
    if (d.feat_dtype == DMVS_DTYPE_F16) return launch_probe<DMVS_DTYPE_F16>(d, grid, block, st);
    if (d.feat_dtype == DMVS_DTYPE_F32) return launch_probe<DMVS_DTYPE_F32>(d, grid, block, st);
    return DMVS_EINVAL;
}

extern "C" int dmvs_probe_getcost_pair_loads_f32(const dmvs_getcost_desc* dp, void* stream) {
    if (!dp || !getcost_desc_ok(*dp)) return DMVS_EINVAL;
    const dmvs_getcost_desc& d = *dp;
    if (d.C != 16 || d.n != 4 || (d.W & 1)) return DMVS_EINVAL;
    const dim3 grid = getcost_grid(d), block(GC_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    if (d.feat_dtype == DMVS_DTYPE_BF16) hipLaunchKernelGGL((getcost_pair_loads_probe_kernel<4, QUAD_TPT, DMVS_DTYPE_BF16>), grid, block, 0, st, d);
    else if (d.feat_dtype == DMVS_DTYPE_F16) hipLaunchKernelGGL((getcost_pair_loads_probe_kernel<4, QUAD_TPT, DMVS_DTYPE_F16>), grid, block, 0, st, d);
    else return DMVS_EINVAL;
    return dmvs_launch_status();
}
