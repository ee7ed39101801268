// Direct fp32 2-D convolution for the small-channel layers of DiffMVS (3..64 in, 1..144 out).
//
// Mapping (CDNA4): one lane = one output pixel, CO output channels live in VGPRs as
// accumulators; the weights of a (ci,ky,kx) tap are wave-uniform, so they are fetched with
// scalar loads (s_load_dwordxN from [cin][kh][kw][cout_pad], cout fastest) and consumed as
// SGPR operands of v_fmac_f32 -- CO FMAs per vector load.  Activations are planar NCHW so
// that the 64 lanes of a wave read 64 consecutive floats of one channel plane.
// Channel concatenation, nearest-x2 upsampling, pixel-unshuffle, r*h gating, folded BN /
// bias, residual adds, activations, the GRU blend and NHWC output are fused (see dmvs.h).
#include "dmvs_common.h"

template <int CO, int KH, int KW, int STRIDE>
__global__ void __launch_bounds__(DMVS_BLOCK) conv2d_kernel(const dmvs_conv2d_desc d) {
    const int total = d.B * d.Hout * d.Wout;
    const int p = blockIdx.x * DMVS_BLOCK + threadIdx.x;
    const int co0 = blockIdx.y * CO;
    const bool live = p < total;
    const int pp = live ? p : total - 1;
    const int ox = pp % d.Wout;
    const int tq = pp / d.Wout;
    const int oy = tq % d.Hout;
    const int b = tq / d.Hout;

    float acc[CO];
#pragma unroll
    for (int i = 0; i < CO; ++i) acc[i] = 0.0f;

    const int iy0 = oy * STRIDE - d.pad_h;
    const int ix0 = ox * STRIDE - d.pad_w;
    const int cin = d.c0 + d.c1;
    const int Hin = d.Hin, Win = d.Win;
    const int mode = d.in_mode;
    // physical plane geometry of in0
    const int pW = mode == DMVS_IN_UPSAMPLE2 ? (Win >> 1) : (mode == DMVS_IN_UNSHUFFLE2 ? (Win << 1) : Win);
    const int pH = mode == DMVS_IN_UPSAMPLE2 ? (Hin >> 1) : (mode == DMVS_IN_UNSHUFFLE2 ? (Hin << 1) : Hin);
    const int pc0 = mode == DMVS_IN_UNSHUFFLE2 ? (d.c0 >> 2) : d.c0;
    const size_t plane0 = (size_t)pH * pW;
    const size_t plane1 = (size_t)Hin * Win;

    for (int ci = 0; ci < cin; ++ci) {
        const float* wrow = d.weight + (size_t)ci * (KH * KW) * d.cout_pad + co0;
        const float* plane;
        const float* mplane = nullptr;
        int sub = 0;   // unshuffle: offset of the (p1,p2) phase inside the 2x2 cell
        if (ci < d.c0) {
            if (mode == DMVS_IN_UNSHUFFLE2) {
                plane = d.in0 + ((size_t)b * pc0 + (ci >> 2)) * plane0;
                sub = ((ci >> 1) & 1) * pW + (ci & 1);
            } else {
                plane = d.in0 + ((size_t)b * pc0 + ci) * plane0;
                if (d.mul0) mplane = d.mul0 + ((size_t)b * pc0 + ci) * plane0;
            }
        } else {
            plane = d.in1 + ((size_t)b * d.c1 + (ci - d.c0)) * plane1;
        }
        const bool first = ci < d.c0;
#pragma unroll
        for (int ky = 0; ky < KH; ++ky) {
            const int iy = iy0 + ky;
            const bool yin = iy >= 0 && iy < Hin;
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                const int ix = ix0 + kx;
                const bool inb = yin && ix >= 0 && ix < Win;
                float v = 0.0f;
                if (inb) {
                    int off;
                    if (!first || mode == DMVS_IN_PLAIN) off = iy * Win + ix;
                    else if (mode == DMVS_IN_UPSAMPLE2) off = (iy >> 1) * pW + (ix >> 1);
                    else off = (iy * 2) * pW + ix * 2 + sub;
                    v = plane[off];
                    if (mplane) v *= mplane[off];
                }
                const float* wt = wrow + (ky * KW + kx) * d.cout_pad;
#pragma unroll
                for (int co = 0; co < CO; ++co) acc[co] = fmaf(v, wt[co], acc[co]);
            }
        }
    }

    if (!live) return;
    const size_t opix = (size_t)oy * d.Wout + ox;
    const size_t oplane = (size_t)d.Hout * d.Wout;
    const int rW = d.res_mode == DMVS_IN_UPSAMPLE2 ? (d.Wout >> 1) : d.Wout;
    const int rH = d.res_mode == DMVS_IN_UPSAMPLE2 ? (d.Hout >> 1) : d.Hout;
    const size_t rpix = d.res_mode == DMVS_IN_UPSAMPLE2 ? (size_t)(oy >> 1) * rW + (ox >> 1) : opix;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const int cg = co0 + co;
        if (cg >= d.cout) break;
        float y = acc[co];
        if (d.scale) y *= d.scale[cg];
        if (d.shift) y += d.shift[cg];
        float r = 0.0f;
        if (d.residual) r = d.residual[((size_t)b * d.cout + cg) * ((size_t)rH * rW) + rpix];
        if (d.residual && !d.res_after_act) y += r;
        y = dmvs_act(y, d.act) * d.post_scale;
        if (d.residual && d.res_after_act) y += r;
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

template <int KH, int KW, int STRIDE>
static int launch_conv2d(const dmvs_conv2d_desc& d, hipStream_t st) {
    const long total = (long)d.B * d.Hout * d.Wout;
    const unsigned pb = dmvs_ceil_div(total, DMVS_BLOCK);
    // CO tile: wide tiles amortise the activation loads, narrow tiles give more workgroups;
    // 256 CUs want >= ~1024 workgroups before a wide tile pays.
    int co = 8;
    if (d.cout_pad % 32 == 0 && pb >= 1024) co = 32;
    else if (d.cout_pad % 16 == 0 && (long)pb * (d.cout_pad / 16) >= 512) co = 16;
    dim3 grid(pb, d.cout_pad / co), block(DMVS_BLOCK);
    if (co == 32) hipLaunchKernelGGL((conv2d_kernel<32, KH, KW, STRIDE>), grid, block, 0, st, d);
    else if (co == 16) hipLaunchKernelGGL((conv2d_kernel<16, KH, KW, STRIDE>), grid, block, 0, st, d);
    else hipLaunchKernelGGL((conv2d_kernel<8, KH, KW, STRIDE>), grid, block, 0, st, d);
    return dmvs_launch_status();
}

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
