// fp32 2-D convolution as an implicit GEMM on the CDNA4 matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32, k-ordered fma chain, 157 TF dense peak).
//
//   D[cout, pixel] += W[cout, k] * X[k, pixel],   k = (ci, ky, kx)
//
// Work decomposition
//   workgroup (4 waves)  = a 16 x (4*MT) tile of output pixels x (NT*16) output channels
//   wave w               = rows MT*w .. MT*w+MT-1 of the tile: MT pixel-tiles of 16 consecutive x
//   MFMA operands        = A: weights  [cout = lane&15][k = lane>>4]
//                          B: inputs   [k = lane>>4][pixel = lane&15]
//                          D: 4 couts (4*(lane>>4)+r) of pixel lane&15  -> NCHW stores are 64 B
//                             runs per cout, NHWC stores are 16 B per lane
//   K loop               = chunks of CK=8 (or 4) input channels staged in LDS (input halo tile
//                          [CK][TH][TW] + weight slab [CK][KH*KW][NT*16]); per (tap, 4-channel
//                          group) every wave issues MT ds_read (B) + NT ds_read (A) for MT*NT
//                          MFMAs.  Channel strides in LDS are padded to 16 mod 32 words so that
//                          the four k-groups of a wave hit disjoint banks.
//   staging              = LDS-DMA (global_load_lds): chunk c+1 streams from HBM/L2 straight into the
//                          other half of a double-buffered LDS image while the matrix cores sweep
//                          chunk c -- no VGPR round trip, one barrier per chunk.
// Everything around the convolution is fused into staging (channel concat, nearest-x2
// upsampling, pixel-unshuffle, r*h gating, zero padding) or into the epilogue (folded BN /
// bias, residual adds, activations, post-scale, GRU blend, NHWC / channel-offset output);
// see include/dmvs.h for the contract.
#include "conv2d_tiled.h"

namespace {

// ------------------------------------------------------------------------------------------
// 1x1 convolutions without the LDS input tile.  A 1x1 layer has no halo and no spatial structure: it is the GEMM
// out[cout][p] = W[cout][ci] X[ci][p] over the flat pixel index p, and at 16..64 input channels it is bound by memory and by the
// per-tile fixed costs of the tiled kernel (input DMA + barrier per 8 channels, 16 x 16 tiles of 64-byte rows), which ran
// these layers at 0.12-0.27 of the MFMA peak and ~2 TB/s.  Round 2's direct kernel (a wave = 64 pixels as the A operand, one float
// per lane and load; it won for one n-tile only) was replaced in round 4 by the 16-byte form below.
constexpr int kC11MaxCin = 64;

// ------------------------------------------------------------------------------------------
// 1x1 layers with 16-BYTE accesses on both sides (round 4).  Round 2's direct kernel fed the matrix cores one float per lane and load
// (lane = pixel m, channel 4c + kq: 256 bytes per wave-level load), which is why its wide instantiations were no faster than the tiled
// kernel.  Here the PIXELS are the B operand and a lane loads FOUR consecutive pixels of its channel (global_load_dwordx4: lane
// (n, kq) holds pixels 4n .. 4n+3 of channel 4c + kq); the four values feed four MFMAs, MFMA r computing the strided pixel set
// {4n + r}.  D[cout][n] then leaves lane (n, kq) with output channels 4kq .. 4kq+3 of pixels 4n .. 4n+3 -- one 16-byte NCHW store per
// channel (acc[0..3][nt][j]), or one 16-byte channel-last store per pixel (acc[r][nt]).  Per 4 input channels and 64 pixels: one load,
// NT weight reads (LDS), 4 NT MFMAs.  Same products in the same k order as the other forms: bit-identical results.
// A wave owns 64 consecutive pixels and every output channel; 3 channel groups of loads in flight.
// ACTX: sigmoid / tanh / SiLU epilogues (inlined transcendental code per value: its own instantiation, one n-tile only)
template <int NT, int OT, bool ACTX = false>
__global__ void __launch_bounds__(DMVS_BLOCK) conv1x1_px4_kernel(const dmvs_conv2d_desc d, int tiles_per_item) {
    constexpr int NW = NT * 16, WS = (NW % 32 == 0) ? NW + 16 : NW;      // weight row stride: the four k-groups on disjoint banks
    __shared__ float s_w[kC11MaxCin * WS];
    DMVS_LDS_POISON(s_w);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int cin = d.c0 + d.c1, cin4 = (cin + 3) >> 2;
    const int HW = d.Hout * d.Wout;
    const int nbase = blockIdx.y * NW;                                   // output channels of this workgroup: nbase .. nbase + NW - 1
    for (int e = tid; e < cin4 * 4 * NW; e += DMVS_BLOCK) {
        const int k = e / NW, co = e - k * NW;
        s_w[k * WS + co] = (k < cin && nbase + co < d.cout_pad) ? d.weight[k * d.cout_pad + nbase + co] : 0.0f;
    }
    const int b = blockIdx.x / tiles_per_item, t = blockIdx.x - b * tiles_per_item;
    const int p = (t * 4 + wave) * 64 + 4 * n;                           // this lane's four pixels (HW % 4 == 0: together or not at all)
    const bool pok = p < HW;
    const int pcl = pok ? p : HW - 4;
    const float* in0b = d.in0 + (size_t)b * d.c0 * HW;
    const float* in1b = d.in1 ? d.in1 + (size_t)b * d.c1 * HW : d.in0;
    __syncthreads();

    f32x4 acc[4][NT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[r][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // Two channel groups ahead: the loads of groups c+1, c+2 are in flight while group c goes through the matrix cores.  The loads are
    // UNCONDITIONAL (channel and pixel clamped into the tensor, the value zeroed when it is consumed): predicated loads sit in their own
    // basic blocks, and hipcc then waits vmcnt(0) -- i.e. also for the prefetch -- before the MFMAs.  Three register sets in rotation
    // (renamed, not moved); the scheduling barrier keeps each prefetch in front of the current group's MFMAs.
    auto load = [&](int c) -> f32x4 {
        int ci = 4 * c + kq;
        ci = ci < cin ? ci : cin - 1;
        const float* src = ci < d.c0 ? in0b + (size_t)ci * HW : in1b + (size_t)(ci - d.c0) * HW;
        return *reinterpret_cast<const f32x4*>(src + pcl);
    };
    auto mma = [&](int c, const f32x4& x) {
        const bool ok = pok && 4 * c + kq < cin;
        const float* wp = s_w + (4 * c + kq) * WS + n;                   // A operand: weights [cout = lane & 15][k = lane >> 4]
        float xv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) xv[r] = ok ? x[r] : 0.0f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float aw = wp[nt * 16];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, xv[r], acc[r][nt], 0, 0, 0);
        }
    };
    f32x4 a0 = load(0), a1 = load(1), a2;
    for (int c = 0;;) {
        a2 = load(c + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(c, a0);
        if (++c >= cin4) break;
        a0 = load(c + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(c, a1);
        if (++c >= cin4) break;
        a1 = load(c + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(c, a2);
        if (++c >= cin4) break;
    }
    if (!pok) return;

    // epilogue: this lane holds output channels nt*16 + 4*kq + j of pixels p .. p+3 (one image row: Wout % 4 == 0)
    const bool rup = d.res_mode == DMVS_IN_UPSAMPLE2;
    const int rW = rup ? (d.Wout >> 1) : d.Wout, rplane = rup ? (d.Hout >> 1) * rW : HW;
    const float* const resb = d.residual ? d.residual + (size_t)b * d.cout * rplane : nullptr;
    const int oy = rup ? p / d.Wout : 0, ox = rup ? p - oy * d.Wout : 0;
    const int rpix = rup ? (oy >> 1) * rW + (ox >> 1) : p;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        f32x4 y[4], res[4];                                               // [j] = channel 4kq + j: its four pixels
        // the four residual loads of an n-tile are issued together, ahead of the arithmetic (one at a time, hipcc waits vmcnt(0) for
        // each before its store: the epilogue then runs at one load latency per channel)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cg = nbase + nt * 16 + 4 * kq + j;
            res[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (resb && cg < d.cout) {
                if (rup) {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 rv = *reinterpret_cast<const f32x2*>(resb + (size_t)cg * rplane + rpix);
                    res[j] = f32x4{rv[0], rv[0], rv[1], rv[1]};
                } else {
                    res[j] = *reinterpret_cast<const f32x4*>(resb + (size_t)cg * HW + p);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cg = nbase + nt * 16 + 4 * kq + j;
            const bool okc = cg < d.cout;
            const float sc = d.scale ? d.scale[okc ? cg : 0] : 1.0f, sh = d.shift ? d.shift[okc ? cg : 0] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[j][r] = acc[r][nt][j] * sc + sh;
            if (!d.res_after_act) y[j] += res[j];
            if constexpr (ACTX) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[j][r] = dmvs_act(y[j][r], d.act);
            } else if (d.act == DMVS_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[j][r] = fmaxf(y[j][r], 0.0f);
            }
            y[j] *= d.post_scale;
            if (d.res_after_act) y[j] += res[j];
        }
        if constexpr (OT != DMVS_DTYPE_F32) {                             // channel-last, 16-bit elements: 8 bytes per pixel and lane
            const int c0o = nbase + nt * 16 + 4 * kq;
            uint16_t* const ob = reinterpret_cast<uint16_t*>(d.out) + ((size_t)b * HW + p) * d.out_cstride + d.out_coffset + c0o;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c0o + j < d.cout) ob[(size_t)r * d.out_cstride + j] = dmvs_to_x16<OT>(y[j][r]);
        } else if (d.out_layout == DMVS_LAYOUT_NCHW) {
            float* const outb = d.out + ((size_t)b * d.out_cstride + d.out_coffset) * HW + p;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cg = nbase + nt * 16 + 4 * kq + j;
                if (cg < d.cout) *reinterpret_cast<f32x4*>(outb + (size_t)cg * HW) = y[j];
            }
        } else {                                                          // channel-last fp32: 16 bytes (4 channels) per pixel and lane
            const int c0o = nbase + nt * 16 + 4 * kq;
            float* const ob = d.out + ((size_t)b * HW + p) * d.out_cstride + d.out_coffset + c0o;
            const bool full = c0o + 3 < d.cout && ((d.out_cstride | d.out_coffset) & 3) == 0 && ((uintptr_t)d.out & 15) == 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (full) {
                    *reinterpret_cast<f32x4*>(ob + (size_t)r * d.out_cstride) = f32x4{y[0][r], y[1][r], y[2][r], y[3][r]};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c0o + j < d.cout) ob[(size_t)r * d.out_cstride + j] = y[j][r];
                }
            }
        }
    }
}

// the 16-byte form applies: plain 1x1 (optionally two concatenated inputs), <= 64 input channels, <= 144 output channels, planes of
// 16-byte multiples on 16-byte aligned tensors, none of the GRU / GroupNorm fusions; NCHW or channel-last output (fp32 / 16-bit)
static bool conv1x1_px4_ok(const dmvs_conv2d_desc& d) {
    if (d.kh != 1 || d.kw != 1 || d.stride != 1 || d.pad_h || d.pad_w || d.in_mode != DMVS_IN_PLAIN) return false;
    if (d.mul0 || d.gru_z || d.gn_stats || d.out_mul || d.in0_cstride || (d.tune & DMVS_TUNE_1X1_TILED)) return false;
    if (d.act > DMVS_ACT_RELU && (d.cout_pad > 16 || d.out_layout != DMVS_LAYOUT_NCHW)) return false;      // (ACTX: one n-tile, planar output)
    if (d.c0 + d.c1 > kC11MaxCin || d.cout_pad > 144 || (d.Wout & 3)) return false;
    if (d.res_mode == DMVS_IN_UPSAMPLE2 && ((d.Hout | d.Wout) & 1)) return false;
    if ((((uintptr_t)d.in0 | (uintptr_t)d.in1 | (uintptr_t)d.residual) & 15) != 0) return false;
    if (d.out_layout == DMVS_LAYOUT_NCHW && ((uintptr_t)d.out & 15)) return false;
    return true;
}

template <int NT>
static int launch_conv1x1_px4(const dmvs_conv2d_desc& d, hipStream_t st, int ngroups) {
    const int HW = d.Hout * d.Wout;
    const int tiles = (HW + 255) / 256;
    dim3 grid((unsigned)(tiles * d.B), (unsigned)ngroups), block(DMVS_BLOCK);
    if (d.out_layout == DMVS_LAYOUT_NHWC_BF16) hipLaunchKernelGGL((conv1x1_px4_kernel<NT, DMVS_DTYPE_BF16>), grid, block, 0, st, d, tiles);
    else if (d.out_layout == DMVS_LAYOUT_NHWC_F16) hipLaunchKernelGGL((conv1x1_px4_kernel<NT, DMVS_DTYPE_F16>), grid, block, 0, st, d, tiles);
    else hipLaunchKernelGGL((conv1x1_px4_kernel<NT, DMVS_DTYPE_F32>), grid, block, 0, st, d, tiles);
    return dmvs_launch_status();
}

// up to 4 n-tiles (64 output channels) per workgroup -- 90 VGPRs; wider layers split into groups that re-read the input (64 -> 144:
// 3 groups of 3: the 64-channel input is read three times, a third of the output's bytes)
static int conv1x1_px4(const dmvs_conv2d_desc& d, hipStream_t st) {
    if (d.act > DMVS_ACT_RELU) {
        const int HW = d.Hout * d.Wout, tiles = (HW + 255) / 256;
        hipLaunchKernelGGL((conv1x1_px4_kernel<1, DMVS_DTYPE_F32, true>), dim3((unsigned)(tiles * d.B)), dim3(DMVS_BLOCK), 0, st, d, tiles);
        return dmvs_launch_status();
    }
    const int ntiles = (d.cout_pad + 15) / 16;
    // n-tiles per workgroup: 4 (3 for 5..6 and 9 tiles), but 2 with a residual -- its reads and the 64 accumulators of 4 n-tiles leave 3
    // waves per SIMD; measured at B = 96 (profiles/r4_conv1x1_px4_ab.txt): FeatureNet inner1 (32 -> 64 + nearest-x2 residual, 576 images)
    // 1.79 ms with 4 n-tiles, 1.44 with 2 x 2 although the 32-channel input is then read twice; without a residual the wide groups win
    // (64 -> 144: 0.62 against 0.71 ms, 64 -> 48 channel-last: 0.30 against 0.46)
    const int maxnt = d.residual ? 2 : 4;
    const int ngroups = (ntiles + maxnt - 1) / maxnt;
    const int nt = (ntiles + ngroups - 1) / ngroups;
    switch (nt) {
        case 1: return launch_conv1x1_px4<1>(d, st, ngroups);
        case 2: return launch_conv1x1_px4<2>(d, st, ngroups);
        case 3: return launch_conv1x1_px4<3>(d, st, ngroups);
        default: return launch_conv1x1_px4<4>(d, st, ngroups);
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient:  gw[ci][t][co] += sum_pixels dY[co][p] * X[ci][p (+) tap t]   as an MFMA GEMM with the
// reduction over pixels:  A = dY [co = lane&15][k = pixel], B = X [k = pixel][j = (ci,t) = lane&15].
// Workgroup = (8 input channels) x (16 output channels), sweeping 16x16 pixel tiles in a grid-stride loop;
// wave w reduces rows 4w..4w+3 of each tile into its own accumulators, which are added to gw with hardware
// fp32 atomics once at the end.  Same LDS-DMA staged halo tile as the forward.
//
// V16 (round 6): both tiles staged in 16-BYTE LDS-DMA pieces.  An LDS-DMA instruction costs the texture path ~55-60 cycles whatever its
// width (HISTORY 4.0 / 4.1), and the 4-byte form needs 27 of them per lane and tile (11 for the 8 x 18 x 18 halo, 16 for the 16 x 256 dY
// tile) against 80 MFMAs per wave: the kernel was DMA-issue-bound at 0.15 of the matrix peak.  Here a halo row is the 16-byte aligned
// cover of its 18 floats (SLACK extra floats on the left, pitch TWL = 24, as in conv2d_mfma_kernel's V16) and a dY row of 16 pixels is 4
// pieces (row pitch 260 floats): 4 + 5 instructions per lane and tile.  A piece lies wholly inside or outside the image (rows of 16-byte
// multiples: the dispatcher checks that, the alignment of the bases, a PLAIN un-gated input and "same" padding).  Same products, same order.
template <int KH, int KW, int S, bool V16>
__global__ void __launch_bounds__(DMVS_BLOCK) conv2d_wgrad_kernel(const dmvs_conv2d_desc d, const float* __restrict__ gout,
                                                                  float* __restrict__ ws, int want_bias, int tiles_x, int tiles_y) {
    constexpr int T = KH * KW, CK = 8;
    constexpr int TW = 15 * S + KW, TH = 15 * S + KH;
    constexpr int SLACK = V16 ? (4 - ((KW - 1) / 2) % 4) % 4 : 0;      // halo column 0 inside an LDS row
    constexpr int TWL = V16 ? (SLACK + TW + 3) / 4 * 4 : TW;           // LDS row pitch of the halo tile
    constexpr int PLANE = pad16mod32(TH * TWL);
    constexpr int GROW = V16 ? 260 : 257;             // dY tile row pitch: 256 pixels + 1 float (bank spread) / + one padding piece
    constexpr int NTN = (CK * T + 1 + 15) / 16;       // MFMA n-tiles over the (ci, tap) pairs of the chunk + the bias column
    // staged units per thread: 4-byte elements, or (V16) 16-byte pieces -- piece e of the halo buffer = LDS floats 4e .. 4e+3
    constexpr int IN_IT = ((V16 ? CK * PLANE / 4 : CK * PLANE) + DMVS_BLOCK - 1) / DMVS_BLOCK;
    constexpr int G_IT = ((V16 ? 16 * GROW / 4 : 16 * GROW) + DMVS_BLOCK - 1) / DMVS_BLOCK;
    __shared__ __attribute__((aligned(16))) float lds[CK * PLANE + 16 * GROW];
    DMVS_LDS_POISON(lds);
    float* s_in = lds;
    float* s_g = lds + CK * PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int c0 = blockIdx.y * CK, cobase = blockIdx.z * 16;
    const int cin = d.c0 + d.c1;
    const int mode = d.in_mode;
    const bool halfres = mode == DMVS_IN_UPSAMPLE2 || mode == DMVS_IN_ZEROINSERT2;
    const int pW = halfres ? (d.Win >> 1) : (mode == DMVS_IN_UNSHUFFLE2 ? (d.Win << 1) : d.Win);
    const int pH = halfres ? (d.Hin >> 1) : (mode == DMVS_IN_UNSHUFFLE2 ? (d.Hin << 1) : d.Hin);
    const int plane0 = pH * pW, plane1 = d.Hin * d.Win;
    const int pc0 = mode == DMVS_IN_UNSHUFFLE2 ? (d.c0 >> 2) : d.c0;
    const size_t oplane = (size_t)d.Hout * d.Wout;

    // this lane's (ci, tap) column of every n-tile -> fixed LDS offset inside the halo tile
    int boff[NTN];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
        const int jj = nt * 16 + m;
        const int ci = jj / T, t = jj - ci * T;
        boff[nt] = jj < CK * T ? ci * PLANE + (t / KW) * TWL + SLACK + (t % KW) : 0;
    }
    f32x4 acc[NTN];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) acc[nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const float bias_one = (want_bias && blockIdx.y == 0) ? 1.0f : 0.0f;

    // tile-independent decode of this lane's staged elements (the integer divisions used to run for every element of
    // every tile and outweighed the MFMA work): halo element -> (input channel, row, column); dY element -> (cout, pixel)
    int in_cig[IN_IT], in_rc[IN_IT];
#pragma unroll
    for (int i = 0; i < IN_IT; ++i) {
        const int e = i * DMVS_BLOCK + tid;
        if constexpr (V16) {      // piece e -> (channel, halo row, first column of the piece relative to the halo's column 0: a multiple of 4 - SLACK)
            const int ci = e / (PLANE / 4), rem = e - ci * (PLANE / 4);
            const int r = rem / (TWL / 4), c = 4 * (rem - r * (TWL / 4)) - SLACK;
            const bool ok = e < CK * PLANE / 4 && rem < TH * TWL / 4 && c0 + ci < cin;
            in_cig[i] = c0 + ci;
            in_rc[i] = ok ? (r | ((c + 8) << 16)) : -1;       // (+8: the slack columns are negative)
        } else {
            const int ci = e / PLANE, rem = e - ci * PLANE;
            const int r = rem / TW, c = rem - r * TW;
            const bool ok = e < CK * PLANE && rem < TH * TW && c0 + ci < cin;
            in_cig[i] = c0 + ci;
            in_rc[i] = ok ? (r | ((c + 8) << 16)) : -1;
        }
    }
    int g_off[G_IT], g_pp[G_IT];
#pragma unroll
    for (int i = 0; i < G_IT; ++i) {
        const int e = i * DMVS_BLOCK + tid;
        const int co = V16 ? e / (GROW / 4) : e / GROW;
        const int p = V16 ? 4 * (e - co * (GROW / 4)) : e - co * GROW;      // first pixel of the piece / the pixel
        const bool ok = e < (V16 ? 16 * GROW / 4 : 16 * GROW) && p < 256 && cobase + co < d.cout;
        g_off[i] = (cobase + co) * (int)oplane + (p >> 4) * d.Wout + (p & 15);       // relative to the tile's first pixel
        g_pp[i] = ok ? p : -1;
    }

    const int ntiles = tiles_x * tiles_y * d.B;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int tq = tile;
        const int tx = tq % tiles_x; tq /= tiles_x;
        const int ty = tq % tiles_y;
        const int b = tq / tiles_y;
        const int ox0 = tx * 16, oy0 = ty * 16;
        const int gy0 = oy0 * S - d.pad_h, gx0 = ox0 * S - d.pad_w;
        const float* in0b = d.in0 + (size_t)b * pc0 * plane0;
        const float* mul0b = d.mul0 ? d.mul0 + (size_t)b * (d.gate_cstride ? d.gate_cstride : pc0) * plane0 : nullptr;
        const float* in1b = d.in1 ? d.in1 + (size_t)b * d.c1 * plane1 : d.in0;
        const float* gb = gout + (size_t)b * d.cout * oplane + (size_t)oy0 * d.Wout + ox0;
        __syncthreads();                              // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < IN_IT; ++i) {
            if (i * DMVS_BLOCK + tid < (V16 ? CK * PLANE / 4 : CK * PLANE)) {
                const int cig = in_cig[i], iy = gy0 + (in_rc[i] & 0xffff), ix = gx0 + (in_rc[i] >> 16) - 8;
                const bool ok = in_rc[i] >= 0 && iy >= 0 && iy < d.Hin && ix >= 0 && ix < d.Win &&
                                !(mode == DMVS_IN_ZEROINSERT2 && ((iy | ix) & 1));
                int off;
                if (V16 || mode == DMVS_IN_PLAIN) off = cig * plane0 + iy * pW + ix;
                else if (halfres) off = cig * plane0 + (iy >> 1) * pW + (ix >> 1);
                else off = (cig >> 2) * plane0 + (iy * 2 + ((cig >> 1) & 1)) * pW + ix * 2 + (cig & 1);
                const float* src = !ok ? dmvs_zero16 : (cig < d.c0 ? in0b + off : in1b + ((cig - d.c0) * plane1 + iy * d.Win + ix));
                if constexpr (V16) __builtin_amdgcn_global_load_lds(src, DMVS_LDS(s_in + (i * DMVS_BLOCK + wave * 64) * 4), 16, 0, 0);
                else __builtin_amdgcn_global_load_lds(src, DMVS_LDS(s_in + i * DMVS_BLOCK + wave * 64), 4, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < G_IT; ++i) {
            if (i * DMVS_BLOCK + tid < (V16 ? 16 * GROW / 4 : 16 * GROW)) {
                const int p = g_pp[i];
                const bool ok = p >= 0 && oy0 + (p >> 4) < d.Hout && ox0 + (p & 15) < d.Wout;
                const float* src = ok ? gb + g_off[i] : dmvs_zero16;
                if constexpr (V16) __builtin_amdgcn_global_load_lds(src, DMVS_LDS(s_g + (i * DMVS_BLOCK + wave * 64) * 4), 16, 0, 0);
                else __builtin_amdgcn_global_load_lds(src, DMVS_LDS(s_g + i * DMVS_BLOCK + wave * 64), 4, 0, 0);
            }
        }
        DMVS_DMA_BARRIER();
        if (!V16 && mul0b) {                           // r*h gating (GRU candidate conv): X = in0 * mul0 on the in0 channels (4-byte form only)
            for (int i = 0; i < IN_IT; ++i) {
                const int e = i * DMVS_BLOCK + tid;
                if (e < CK * PLANE) {
                    const int ci = e / PLANE, rem = e - ci * PLANE;
                    const int r = rem / TW, c = rem - r * TW;
                    const int cig = c0 + ci, iy = gy0 + r, ix = gx0 + c;
                    if (rem < TH * TW && cig < d.c0 && iy >= 0 && iy < d.Hin && ix >= 0 && ix < d.Win)
                        s_in[e] *= mul0b[cig * plane0 + iy * pW + ix];
                }
            }
            __syncthreads();
        }
#pragma unroll 1
        for (int rr = 0; rr < 4; ++rr) {
            const int row = wave * 4 + rr;
#pragma unroll
            for (int xg = 0; xg < 4; ++xg) {
                const float av = s_g[m * GROW + row * 16 + xg * 4 + kq];
                const float* ip = s_in + (row * S) * TWL + (xg * 4 + kq) * S;
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) {
                    float bv = ip[boff[nt]];
                    if (nt == (CK * T) / 16) bv = m == (CK * T) % 16 ? bias_one : bv;   // column of ones: sum(dY) = bias gradient
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[nt], 0, 0, 0);
                }
            }
        }
    }
    // D[co = 4*kq + r][j = m].  Deterministic block total: the four waves add their accumulators into LDS one after
    // the other, then the block stores ONE partial [NTN*16 (ci,tap)][16 co] into its workspace slot; a second kernel
    // sums the slots (same-address global atomics from ~10^3 workgroups serialise at ~0.1 us each).
    __syncthreads();
    float* red = lds;
    static_assert(NTN * 256 <= CK * PLANE + 16 * GROW, "block partial must fit the tile buffers");
#pragma unroll 1
    for (int w = 0; w < DMVS_BLOCK / 64; ++w) {
        if (wave == w) {
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* q = red + (nt * 16 + m) * 16 + kq * 4 + r;
                    *q = w == 0 ? acc[nt][r] : *q + acc[nt][r];
                }
        }
        __syncthreads();
    }
    float* slot = ws + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (NTN * 256);
    for (int e = tid; e < NTN * 256; e += DMVS_BLOCK) slot[e] = red[e];
}

// gw[co][ci][tap] (torch layout) = sum over the gx workspace slots of one (cin chunk, cout tile); the spare MFMA
// column CK*T of cin chunk 0 carries sum(dY) = the bias gradient.
template <int T>
__global__ void __launch_bounds__(DMVS_BLOCK) conv2d_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ gw,
                                                                         float* __restrict__ gb, int gx, int gy, int cin,
                                                                         int cout, int accumulate) {
    constexpr int CK = 8, NTN = (CK * T + 1 + 15) / 16, PER = NTN * 256, SL = 16;
    __shared__ float red[SL][17];
    DMVS_LDS_POISON(red);
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;                // element of the [NTN*16][16] partial
    const int by = blockIdx.y, bz = blockIdx.z;
    const float* base = ws + (size_t)(bz * gy + by) * gx * PER + e;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int x = sl;
    for (; x + 3 * SL < gx; x += 4 * SL) {
        a0 += base[(size_t)x * PER];
        a1 += base[(size_t)(x + SL) * PER];
        a2 += base[(size_t)(x + 2 * SL) * PER];
        a3 += base[(size_t)(x + 3 * SL) * PER];
    }
    for (; x < gx; x += SL) a0 += base[(size_t)x * PER];
    red[sl][el] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0) {
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < SL; ++i) t += red[i][el];
        const int jj = e >> 4, co = bz * 16 + (e & 15);
        const int ci = jj / T, tap = jj - ci * T;
        if (co < cout) {
            if (jj < CK * T && by * CK + ci < cin) {
                float* q = gw + ((size_t)co * cin + by * CK + ci) * T + tap;
                *q = accumulate ? *q + t : t;               // (DMVS_TUNE_WGRAD_ACCUMULATE: into the caller's running gradient)
            } else if (jj == CK * T && by == 0 && gb) {
                gb[co] = accumulate ? gb[co] + t : t;
            }
        }
    }
}

struct WgradGrid {
    int gx, gy, gz, ntn;
    long floats;
};
static WgradGrid wgrad_grid(const dmvs_conv2d_desc& d) {
    const int tiles_x = (d.Wout + 15) / 16, tiles_y = (d.Hout + 15) / 16;
    const long ntiles = (long)tiles_x * tiles_y * d.B;
    WgradGrid g;
    g.gy = (d.c0 + d.c1 + 7) / 8;
    g.gz = (d.cout + 15) / 16;
    // ~4 workgroups per CU so that tile loads of one overlap the MFMA phase of another; each adds a workspace slot
    long gx = (1024 + g.gy * g.gz - 1) / (g.gy * g.gz);
    if (gx > ntiles) gx = ntiles;
    g.gx = gx < 1 ? 1 : (int)gx;
    g.ntn = (8 * d.kh * d.kw + 1 + 15) / 16;
    g.floats = (long)g.gx * g.gy * g.gz * g.ntn * 256;
    return g;
}

template <int KH, int KW, int S>
int launch_wgrad(const dmvs_conv2d_desc& d, const float* gout, float* gw, float* gb, float* ws, hipStream_t st) {
    const int tiles_x = (d.Wout + 15) / 16, tiles_y = (d.Hout + 15) / 16;
    const WgradGrid g = wgrad_grid(d);
    // 16-byte staging pieces: plain un-gated inputs with "same" padding, rows of 16-byte multiples on 16-byte aligned tensors
    const bool v16 = !(d.tune & DMVS_TUNE_PIECES4) && d.in_mode == DMVS_IN_PLAIN && !d.mul0 && (d.Win & 3) == 0 && (d.Wout & 3) == 0 &&
                     d.pad_w == (KW - 1) / 2 && (((uintptr_t)d.in0 | (uintptr_t)d.in1 | (uintptr_t)gout) & 15) == 0;
    if (v16)
        hipLaunchKernelGGL((conv2d_wgrad_kernel<KH, KW, S, true>), dim3(g.gx, g.gy, g.gz), dim3(DMVS_BLOCK), 0, st, d, gout, ws,
                           gb ? 1 : 0, tiles_x, tiles_y);
    else
    hipLaunchKernelGGL((conv2d_wgrad_kernel<KH, KW, S, false>), dim3(g.gx, g.gy, g.gz), dim3(DMVS_BLOCK), 0, st, d, gout, ws,
                       gb ? 1 : 0, tiles_x, tiles_y);
    hipLaunchKernelGGL((conv2d_wgrad_reduce_kernel<KH * KW>), dim3(g.ntn * 16, g.gy, g.gz), dim3(DMVS_BLOCK), 0, st, ws, gw, gb,
                       g.gx, g.gy, d.c0 + d.c1, d.cout, (d.tune & DMVS_TUNE_WGRAD_ACCUMULATE) ? 1 : 0);
    return dmvs_launch_status();
}

}  // namespace

extern "C" int dmvs_conv2d_f32(const dmvs_conv2d_desc* dp, void* stream) {
    if (!dp) return DMVS_EINVAL;
    const dmvs_conv2d_desc& d = *dp;
    hipStream_t st = (hipStream_t)stream;
    if (d.cout_pad % 8 || d.cout > d.cout_pad || d.B <= 0 || !d.in0 || !d.weight || !d.out) return DMVS_EINVAL;
    if ((uintptr_t)d.weight & 15) return DMVS_EINVAL;      // the weight slab is staged in 16-byte LDS-DMA pieces
    if (d.c1 > 0 && (!d.in1 || d.in_mode != DMVS_IN_PLAIN)) return DMVS_EINVAL;
    if (d.mul0 && d.in_mode != DMVS_IN_PLAIN) return DMVS_EINVAL;
    if (d.in_mode == DMVS_IN_UNSHUFFLE2 && (d.c0 % 4)) return DMVS_EINVAL;
    if ((d.in_mode == DMVS_IN_UPSAMPLE2 || d.in_mode == DMVS_IN_ZEROINSERT2) && ((d.Hin | d.Win) & 1)) return DMVS_EINVAL;
    if (d.gru_z && (!d.gru_h || d.act != DMVS_ACT_TANH)) return DMVS_EINVAL;
    if (d.gate_cstride < 0 || (d.gate_cstride && d.in_mode != DMVS_IN_PLAIN)) return DMVS_EINVAL;
    if (d.in0_cstride < 0 || (d.in0_cstride && (d.in_mode != DMVS_IN_PLAIN || d.in0_cstride < d.c0))) return DMVS_EINVAL;
    if (d.out_mul && (d.out_mul_c0 < 0 || d.out_mul_c0 >= d.cout || d.gn_stats)) return DMVS_EINVAL;
    if (d.gn_stats && (d.gn_groups != 4 || d.cout % 4)) return DMVS_EINVAL;
    if (d.arith != DMVS_ARITH_F32 && d.arith != DMVS_ARITH_BF16 && d.arith != DMVS_ARITH_SPLIT) return DMVS_EINVAL;
    const int eh = (d.Hin + 2 * d.pad_h - d.kh) / d.stride + 1, ew = (d.Win + 2 * d.pad_w - d.kw) / d.stride + 1;
    if (eh != d.Hout || ew != d.Wout) return DMVS_EINVAL;
    // 32-bit element offsets inside one batch item
    if ((long)(d.c0 + d.c1) * d.Hin * d.Win * (d.in_mode == DMVS_IN_UNSHUFFLE2 ? 4 : 1) >= (1L << 31)) return DMVS_EINVAL;
    // epilogue: channel * plane + pixel in 24-bit x 24-bit multiplies and 32-bit sums
    if (d.out_layout < DMVS_LAYOUT_NCHW || d.out_layout > DMVS_LAYOUT_NHWC_F16) return DMVS_EINVAL;
    if (d.out_layout >= DMVS_LAYOUT_NHWC_BF16 && (d.gru_z || d.gn_stats)) return DMVS_EINVAL;
    const int ocs = d.out_cstride > d.cout ? d.out_cstride : d.cout;
    if ((long)d.Hout * d.Wout >= (1L << 24) || ocs >= (1 << 24) || (long)ocs * d.Hout * d.Wout >= (1L << 31)) return DMVS_EINVAL;
    const int key = d.kh * 100 + d.kw * 10 + d.stride;
    switch (key) {
        case 111:
            return conv1x1_px4_ok(d) ? conv1x1_px4(d, st) : dmvs_detail::launch_conv2d_111(d, st);
        case 331: return dmvs_detail::launch_conv2d_331(d, st);
        case 332: return dmvs_detail::launch_conv2d_332(d, st);
        case 552: return dmvs_detail::launch_conv2d_552(d, st);
        case 551: return dmvs_detail::launch_conv2d_551(d, st);   // input gradient of the 5x5 stride-2 layers
        case 771: return dmvs_detail::launch_conv2d_771(d, st);
        case 151: return dmvs_detail::launch_conv2d_151(d, st);
        case 511: return dmvs_detail::launch_conv2d_511(d, st);
        default: return DMVS_EINVAL;
    }
}

static int wgrad_check(const dmvs_conv2d_desc& d) {
    if (d.cout_pad % 8 || d.cout > d.cout_pad || d.B <= 0 || !d.in0) return DMVS_EINVAL;
    if (d.c1 > 0 && (!d.in1 || d.in_mode != DMVS_IN_PLAIN)) return DMVS_EINVAL;
    if (d.mul0 && d.in_mode != DMVS_IN_PLAIN) return DMVS_EINVAL;
    // the weight-gradient kernels read dense [B,c0,..] / [B,c1,..] inputs and dense [B,c0,..] gates only: a channel-slice in0
    // (in0_cstride), a gate slice (gate_cstride) or a producer-side output product (out_mul changes what the forward's consumer read)
    // would be read from the wrong memory -- refused, not ignored
    if (d.in0_cstride || d.gate_cstride || d.out_mul) return DMVS_EINVAL;
    const int eh = (d.Hin + 2 * d.pad_h - d.kh) / d.stride + 1, ew = (d.Win + 2 * d.pad_w - d.kw) / d.stride + 1;
    if (eh != d.Hout || ew != d.Wout) return DMVS_EINVAL;
    return 0;
}

extern "C" int dmvs_conv2d_wgrad_workspace_f32(const dmvs_conv2d_desc* dp, int64_t* bytes) {
    if (!dp || !bytes) return DMVS_EINVAL;
    if (int rc = wgrad_check(*dp)) return rc;
    *bytes = (int64_t)wgrad_grid(*dp).floats * 4;
    return 0;
}

extern "C" int dmvs_conv2d_wgrad_f32(const dmvs_conv2d_desc* dp, const float* grad_out, float* gw, float* gb, float* workspace,
                                     int64_t workspace_bytes, void* stream) {
    if (!dp || !grad_out || !gw || !workspace) return DMVS_EINVAL;
    const dmvs_conv2d_desc& d = *dp;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = wgrad_check(d)) return rc;
    if (workspace_bytes < (int64_t)wgrad_grid(d).floats * 4) return DMVS_EINVAL;
    const int key = d.kh * 100 + d.kw * 10 + d.stride;
    switch (key) {
        case 111: return launch_wgrad<1, 1, 1>(d, grad_out, gw, gb, workspace, st);
        case 331: return launch_wgrad<3, 3, 1>(d, grad_out, gw, gb, workspace, st);
        case 332: return launch_wgrad<3, 3, 2>(d, grad_out, gw, gb, workspace, st);
        case 552: return launch_wgrad<5, 5, 2>(d, grad_out, gw, gb, workspace, st);
        case 771: return launch_wgrad<7, 7, 1>(d, grad_out, gw, gb, workspace, st);
        case 151: return launch_wgrad<1, 5, 1>(d, grad_out, gw, gb, workspace, st);
        case 511: return launch_wgrad<5, 1, 1>(d, grad_out, gw, gb, workspace, st);
        default: return DMVS_EINVAL;
    }
}
