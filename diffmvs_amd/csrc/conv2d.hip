// fp32 2-D convolution as an implicit GEMM on the CDNA4 matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32, k-ordered fma chain, 157 TF dense peak).
//
//   D[cout, pixel] += W[cout, k] * X[k, pixel],   k = (ci, ky, kx)
//
// Work decomposition
//   workgroup (4 waves)  = one 16x16 tile of output pixels x (NT*16) output channels
//   wave w               = rows 4w..4w+3 of the tile: 4 pixel-tiles of 16 consecutive x
//   MFMA operands        = A: weights  [cout = lane&15][k = lane>>4]
//                          B: inputs   [k = lane>>4][pixel = lane&15]
//                          D: 4 couts (4*(lane>>4)+r) of pixel lane&15  -> NCHW stores are 64 B
//                             runs per cout, NHWC stores are 16 B per lane
//   K loop               = chunks of CK=8 (or 4) input channels staged in LDS (input halo tile
//                          [CK][TH][TW] + weight slab [CK][KH*KW][NT*16]); per (tap, 4-channel
//                          group) every wave issues 4 ds_read_b32 (B) + NT ds_read_b32 (A)
//                          for 4*NT MFMAs.  Channel strides in LDS are padded to 16 mod 32
//                          words so that the four k-groups of a wave hit disjoint banks.
// Everything around the convolution is fused into staging (channel concat, nearest-x2
// upsampling, pixel-unshuffle, r*h gating, zero padding) or into the epilogue (folded BN /
// bias, residual adds, activations, post-scale, GRU blend, NHWC / channel-offset output);
// see include/dmvs.h for the contract.
#include "dmvs_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kTile = 16;   // output tile edge (pixels)


constexpr int pad16mod32(int n) {   // smallest m >= n with m % 32 == 16
    int m = n;
    while (m % 32 != 16) ++m;
    return m;
}

// logical input element (b, ci, iy, ix) with zero padding; handles concat / upsample / unshuffle / gating
__device__ __forceinline__ float conv_in(const dmvs_conv2d_desc& d, int b, int ci, int iy, int ix) {
    if (ci >= d.c0 + d.c1 || iy < 0 || iy >= d.Hin || ix < 0 || ix >= d.Win) return 0.0f;
    if (ci >= d.c0) return d.in1[(((size_t)b * d.c1 + (ci - d.c0)) * d.Hin + iy) * d.Win + ix];
    if (d.in_mode == DMVS_IN_PLAIN) {
        const size_t o = (((size_t)b * d.c0 + ci) * d.Hin + iy) * d.Win + ix;
        float v = d.in0[o];
        if (d.mul0) v *= d.mul0[o];
        return v;
    }
    if (d.in_mode == DMVS_IN_UPSAMPLE2) {
        const int pH = d.Hin >> 1, pW = d.Win >> 1;
        return d.in0[(((size_t)b * d.c0 + ci) * pH + (iy >> 1)) * pW + (ix >> 1)];
    }
    // pixel-unshuffle: logical channel ci = c*4 + p1*2 + p2 reads physical (c, 2*iy+p1, 2*ix+p2)
    const int pH = d.Hin << 1, pW = d.Win << 1;
    return d.in0[(((size_t)b * (d.c0 >> 2) + (ci >> 2)) * pH + (iy * 2 + ((ci >> 1) & 1))) * pW + ix * 2 + (ci & 1)];
}

template <int KH, int KW, int S, int NT>
__global__ void __launch_bounds__(DMVS_BLOCK) conv2d_mfma_kernel(const dmvs_conv2d_desc d, int tiles_x, int tiles_y) {
    constexpr int T = KH * KW;
    constexpr int TW = (kTile - 1) * S + KW, TH = (kTile - 1) * S + KH;
    constexpr int PLANE = pad16mod32(TH * TW);
    constexpr int NW = NT * 16;
    constexpr int WPAD = pad16mod32(T * NW);
    // input channels per LDS chunk: 8, or 4 when 8 would push a workgroup past 48 KB of LDS
    constexpr int kCK = (8 * (PLANE + WPAD) * 4 > 49152) ? 4 : 8;
    __shared__ float lds[kCK * PLANE + kCK * WPAD];
    float* s_in = lds;
    float* s_w = lds + kCK * PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    int tile = blockIdx.x;
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y;
    const int b = tile / tiles_y;
    const int ox0 = tx * kTile, oy0 = ty * kTile;
    const int nbase = blockIdx.y * NW;
    const int cin = d.c0 + d.c1;
    const int gy0 = oy0 * S - d.pad_h, gx0 = ox0 * S - d.pad_w;

    f32x4 acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    for (int c0 = 0; c0 < cin; c0 += kCK) {
        __syncthreads();
        for (int e = tid; e < kCK * TH * TW; e += DMVS_BLOCK) {
            const int ci = e / (TH * TW), rem = e % (TH * TW);
            const int r = rem / TW, c = rem % TW;
            s_in[ci * PLANE + r * TW + c] = conv_in(d, b, c0 + ci, gy0 + r, gx0 + c);
        }
        for (int e = tid; e < kCK * T * NW; e += DMVS_BLOCK) {
            const int ci = e / (T * NW), rem = e % (T * NW);
            const int t = rem / NW, n = rem % NW;
            float w = 0.0f;
            if (c0 + ci < cin && nbase + n < d.cout_pad) w = d.weight[((size_t)(c0 + ci) * T + t) * d.cout_pad + nbase + n];
            s_w[ci * WPAD + t * NW + n] = w;
        }
        __syncthreads();
        const int live_c = cin - c0 < kCK ? cin - c0 : kCK;
        const int nc4 = (live_c + 3) >> 2;          // skip all-zero 4-channel groups of the last chunk
#pragma unroll 1
        for (int ky = 0; ky < KH; ++ky) {
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
#pragma unroll
                for (int c4 = 0; c4 < kCK / 4; ++c4) {
                    if (c4 >= nc4) break;
                    const int ci = c4 * 4 + kq;
                    float av[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) av[nt] = s_w[ci * WPAD + (ky * KW + kx) * NW + nt * 16 + m];
                    const float* ip = s_in + ci * PLANE + (wave * 4 * S + ky) * TW + m * S + kx;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const float bv = ip[mt * S * TW];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[nt], bv, acc[mt][nt], 0, 0, 0);
                    }
                }
            }
        }
    }

    // epilogue: this lane holds couts nbase + nt*16 + 4*kq + r of pixel (oy0 + 4*wave + mt, ox0 + m)
    const int ox = ox0 + m;
    if (ox >= d.Wout) return;
    const size_t oplane = (size_t)d.Hout * d.Wout;
    const bool rup = d.res_mode == DMVS_IN_UPSAMPLE2;
    const int rW = rup ? (d.Wout >> 1) : d.Wout, rH = rup ? (d.Hout >> 1) : d.Hout;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int oy = oy0 + wave * 4 + mt;
        if (oy >= d.Hout) continue;
        const size_t opix = (size_t)oy * d.Wout + ox;
        const size_t rpix = rup ? (size_t)(oy >> 1) * rW + (ox >> 1) : opix;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cg = nbase + nt * 16 + kq * 4 + r;
                if (cg >= d.cout) continue;
                float y = acc[mt][nt][r];
                if (d.scale) y *= d.scale[cg];
                if (d.shift) y += d.shift[cg];
                float res = 0.0f;
                if (d.residual) res = d.residual[((size_t)b * d.cout + cg) * ((size_t)rH * rW) + rpix];
                if (d.residual && !d.res_after_act) y += res;
                y = dmvs_act(y, d.act) * d.post_scale;
                if (d.residual && d.res_after_act) y += res;
                if (d.gru_z) {
                    const size_t gi = ((size_t)b * d.cout + cg) * oplane + opix;
                    const float z = d.gru_z[gi];
                    y = (1.0f - z) * d.gru_h[gi] + z * y;
                }
                if (d.out_layout == DMVS_LAYOUT_NCHW)
                    d.out[((size_t)b * d.out_cstride + d.out_coffset + cg) * oplane + opix] = y;
                else
                    d.out[((size_t)b * oplane + opix) * d.out_cstride + d.out_coffset + cg] = y;
            }
        }
    }
}

template <int KH, int KW, int S>
int launch_conv2d(const dmvs_conv2d_desc& d, hipStream_t st) {
    const int tiles_x = (d.Wout + kTile - 1) / kTile, tiles_y = (d.Hout + kTile - 1) / kTile;
    const int ntiles = (d.cout_pad + 15) / 16;
    // output channels per workgroup: up to 4 MFMA n-tiles share one staged input tile
    int nt = ntiles <= 4 ? ntiles : (ntiles % 3 == 0 ? 3 : 4);
    dim3 grid((unsigned)(tiles_x * tiles_y * d.B), (unsigned)((ntiles + nt - 1) / nt)), block(DMVS_BLOCK);
    switch (nt) {
        case 1: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 1>), grid, block, 0, st, d, tiles_x, tiles_y); break;
        case 2: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 2>), grid, block, 0, st, d, tiles_x, tiles_y); break;
        case 3: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 3>), grid, block, 0, st, d, tiles_x, tiles_y); break;
        default: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 4>), grid, block, 0, st, d, tiles_x, tiles_y); break;
    }
    return dmvs_launch_status();
}

}  // namespace

extern "C" int dmvs_conv2d_f32(const dmvs_conv2d_desc* dp, void* stream) {
    if (!dp) return DMVS_EINVAL;
    const dmvs_conv2d_desc& d = *dp;
    hipStream_t st = (hipStream_t)stream;
    if (d.cout_pad % 8 || d.cout > d.cout_pad || d.B <= 0 || !d.in0 || !d.weight || !d.out) return DMVS_EINVAL;
    if (d.c1 > 0 && (!d.in1 || d.in_mode != DMVS_IN_PLAIN)) return DMVS_EINVAL;
    if (d.mul0 && d.in_mode != DMVS_IN_PLAIN) return DMVS_EINVAL;
    if (d.in_mode == DMVS_IN_UNSHUFFLE2 && (d.c0 % 4)) return DMVS_EINVAL;
    if (d.in_mode == DMVS_IN_UPSAMPLE2 && ((d.Hin | d.Win) & 1)) return DMVS_EINVAL;
    if (d.gru_z && (!d.gru_h || d.act != DMVS_ACT_TANH)) return DMVS_EINVAL;
    const int eh = (d.Hin + 2 * d.pad_h - d.kh) / d.stride + 1, ew = (d.Win + 2 * d.pad_w - d.kw) / d.stride + 1;
    if (eh != d.Hout || ew != d.Wout) return DMVS_EINVAL;
    const int key = d.kh * 100 + d.kw * 10 + d.stride;
    switch (key) {
        case 111: return launch_conv2d<1, 1, 1>(d, st);
        case 331: return launch_conv2d<3, 3, 1>(d, st);
        case 332: return launch_conv2d<3, 3, 2>(d, st);
        case 552: return launch_conv2d<5, 5, 2>(d, st);
        case 771: return launch_conv2d<7, 7, 1>(d, st);
        case 151: return launch_conv2d<1, 5, 1>(d, st);
        case 511: return launch_conv2d<5, 1, 1>(d, st);
        default: return DMVS_EINVAL;
    }
}
