// The tail of a refinement stage in ONE kernel: the mask head's 1x1 convolution (64 -> 9*4*4 = 144 channels, reference
// models/update.py:335-339, :473 `mask = .25 * self.mask(context)`) + upsample_depth (models/module.py:237-248: softmax over the 9 taps,
// convex combination of the 3x3 neighbourhood, pixel shuffle x4) + disp_to_depth (:220-227) -- what models/diffusion.py:281-283 does with the
// last iterate of the DiffMVS stage.
//
// Why fused: as two launches the 144-channel mask was the step's largest tensor that exists only to be consumed once -- 1.13 GB written by the
// 1x1 layer (write-bound at 2.6 TB/s) and read back by the upsampling, per 96-view step, for 0.13 GB of depth maps.  Here the 144 logits of a
// pixel never leave the registers of the lanes that computed them.
//
// Mapping.  The 1x1 layer is the GEMM  logits[co][p] = W[co][ci] X[ci][p]  on v_mfma_f32_16x16x4_f32 with A = weights (cout = lane & 15,
// k = lane >> 4) and B = pixels: D leaves lane (m, kq) with output channels 16t + 4kq + j (j = 0..3) of pixel m for n-tile t.  Channel
// c of the mask is tap * 16 + jy * 4 + jx (mask.view(N, 1, 9, 4, 4, H, W), module.py:241), so n-tile t IS tap t and the lane holds, for its
// pixel, all 9 taps of sub-pixel row jy = kq, columns jx = 0..3: the softmax over the taps and the convex combination are per-lane register
// arithmetic, and the four results are 16 contiguous bytes of output row 4y + kq.  A wave owns 16 * MT consecutive pixels.
//
// Same products in the same k order as conv1x1_px4_kernel (conv2d.hip), the same epilogue order ((acc + bias) * post_scale) and the same
// softmax / combination arithmetic as convex_upsample_kernel (misc.hip): the results are those of the two launches bit for bit
// (tests/test_ops.py::test_mask_upsample4).
#include "dmvs_common.h"
#include "dmvs_lds_poison.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MU_CIN = 64, MU_NT = 9, MU_WS = 144;      // weight row pitch 144 = 16 mod 32: the k-groups of a 32-lane half on disjoint banks

// (4 waves per SIMD = the 4 workgroups per CU the 36 KB weight slab allows: the register allocator is held to 128 VGPRs)
template <int MT>
__global__ void __launch_bounds__(DMVS_BLOCK, 4)
mask_upsample4_kernel(const float* __restrict__ x, const float* __restrict__ weight, const float* __restrict__ bias, float post_scale,
                      const float* __restrict__ inv, const float* __restrict__ disp_min, const float* __restrict__ disp_max,
                      float* __restrict__ out_inv, float* __restrict__ out_depth, int H, int W, int blocks_per_item, int nblocks) {
    __shared__ float s_w[MU_CIN * MU_WS];
    DMVS_LDS_POISON(s_w);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int HW = H * W;
    for (int e = tid; e < MU_CIN * MU_WS; e += DMVS_BLOCK) s_w[e] = weight[e];
    __syncthreads();

    // resident workgroups walk the pixel blocks (the 36 KB weight slab is read once per workgroup)
    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int b = blk / blocks_per_item, tb = blk - b * blocks_per_item;
        const int p0 = (tb * 4 + wave) * (16 * MT) + m;              // this lane's pixel of pixel group 0
        const float* xb = x + (size_t)b * MU_CIN * HW + (size_t)kq * HW;
        f32x4 acc[MT][MU_NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < MU_NT; ++t) acc[mt][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        // B operand: X[ci = 4g + kq][pixel]; unconditional clamped loads, the value zeroed when consumed.  Two halves of 8 channel groups: the
        // second half's loads are in flight under the first half's MFMAs.
        float bv[2][8][MT];
        auto load_half = [&](int hf) {
#pragma unroll
            for (int g = 0; g < 8; ++g)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int p = p0 + 16 * mt;
                    bv[hf][g][mt] = xb[(size_t)(4 * (8 * hf + g)) * HW + (p < HW ? p : HW - 1)];
                }
        };
        auto mma_half = [&](int hf) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float* wp = s_w + (4 * (8 * hf + g) + kq) * MU_WS + m;
                float xv[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xv[mt] = (p0 + 16 * mt < HW) ? bv[hf][g][mt] : 0.0f;
#pragma unroll
                for (int t = 0; t < MU_NT; ++t) {
                    const float aw = wp[16 * t];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, xv[mt], acc[mt][t], 0, 0, 0);
                }
            }
        };
        load_half(0);
        load_half(1);
        __builtin_amdgcn_sched_barrier(0);
        mma_half(0);
        mma_half(1);

        const float dmin = disp_min[b], dmax = disp_max[b];
        const float* invb = inv + (size_t)b * HW;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int p = p0 + 16 * mt;
            if (p >= HW) continue;
            const int y = p / W, xx = p - y * W;
            float v[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int yy = y + k / 3 - 1, xn = xx + k % 3 - 1;
                v[k] = (yy >= 0 && yy < H && xn >= 0 && xn < W) ? invb[yy * W + xn] : 0.0f;
            }
            f32x4 up;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float lg[9], mx = -3.0e38f;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    lg[k] = (acc[mt][k][j] + (bias ? bias[16 * k + 4 * kq + j] : 0.0f)) * post_scale;      // (36 cached scalars-per-lane: not worth 36 registers)
                    mx = fmaxf(mx, lg[k]);
                }
                float sum = 0.0f, a = 0.0f;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float e = expf(lg[k] - mx);
                    sum += e;
                    a = fmaf(e, v[k], a);
                }
                up[j] = a / sum;
            }
            const size_t o = ((size_t)b * 4 * H + (size_t)(4 * y + kq)) * (size_t)(4 * W) + (size_t)(4 * xx);
            if (out_inv) *reinterpret_cast<f32x4*>(out_inv + o) = up;
            if (out_depth) {
                f32x4 dd;
#pragma unroll
                for (int j = 0; j < 4; ++j) dd[j] = dmvs_disp_to_depth(up[j], dmin, dmax);
                *reinterpret_cast<f32x4*>(out_depth + o) = dd;
            }
        }
    }
}

}  // namespace

extern "C" int dmvs_mask_upsample4_f32(const float* x, const float* weight, const float* bias, float post_scale, const float* inv,
                                       const float* disp_min, const float* disp_max, float* out_inv, float* out_depth, int32_t B,
                                       int32_t cin, int32_t cout_pad, int32_t H, int32_t W, void* stream) {
    if (!x || !weight || !inv || !disp_min || !disp_max || (!out_inv && !out_depth)) return DMVS_EINVAL;
    if (cin != MU_CIN || cout_pad != MU_WS || B < 1 || H < 1 || W < 1 || (long)H * W >= (1L << 24)) return DMVS_EINVAL;
    if ((((uintptr_t)out_inv | (uintptr_t)out_depth) & 15) != 0) return DMVS_EINVAL;          // 16-byte stores (rows of 4 W floats)
    constexpr int MT = 2;
    const int HW = H * W;
    const int blocks_per_item = (HW + 64 * MT - 1) / (64 * MT);
    const long nblocks = (long)blocks_per_item * B;
    if (nblocks > 0x7fffffffL) return DMVS_EINVAL;
    int grid = dmvs_resident_workgroups(reinterpret_cast<const void*>(&mask_upsample4_kernel<MT>));
    if ((long)grid > nblocks) grid = (int)nblocks;
    hipLaunchKernelGGL(mask_upsample4_kernel<MT>, dim3((unsigned)grid), dim3(DMVS_BLOCK), 0, (hipStream_t)stream, x, weight, bias, post_scale, inv,
                       disp_min, disp_max, out_inv, out_depth, H, W, blocks_per_item, (int)nblocks);
    return dmvs_launch_status();
}
