// Backward of the fused homography-warp + group-correlation kernels (training step; reference:
// autograd through models/module.py:212-218 grid_sample and :529-548 / :644-661).
//
// The reference builds the sampling grid under no_grad (module.py:187) and detaches the depth
// hypotheses and view weights of GetCost (update.py:442-445, module.py:573), so gradients flow to the
// image features only:
//   grad_ref[b,p,c] = 1/Cg * sum_{s,d} gcor[b,s,g(c),d,p] * warped_s[b,c,d,p]          (gather, registers)
//   grad_src[s,b,q,c] += 1/Cg * sum_{d,p: tap(p,d)=q} gcor[...] * ref[b,p,c] * w_tap     (scatter, fp32 atomics)
// Same "texel-coalesced" lane mapping as the forward (LPP lanes per pixel, CPL channels per lane).  The
// scatter is pre-accumulated in registers while consecutive hypotheses keep the same 2x2 footprint and
// flushed with one hardware fp32 atomic per (tap, channel) when it moves; atomics make the feature
// gradient order-dependent in the last bits (documented non-determinism).
//
// Channel -> lane mapping (template CS = channel stride between a lane's consecutive channels).  Rounds 1-5: lane `sub` of a pixel owns
// the CPL CONSECUTIVE channels sub*CPL .. +CPL-1 (CS = 1): 16-byte loads, but each of the CPL atomic instructions of a flush then touches
// 4 bytes out of every 16 of the texel -- LPP lanes spread over the whole 4C-byte texel, CPL times over.  Round 6 (CS = LPP, "interleaved"):
// lane `sub` owns channels sub, sub + LPP, ...: one atomic instruction covers LPP CONTIGUOUS floats of the texel (64 bytes at 16 lanes per
// pixel), i.e. the L2's atomic units see C*4/64 full 64-byte requests per (pixel, tap) instead of CPL * C*4/64 quarter-filled ones.
#include "dmvs_common.h"

namespace {

struct RayB {
    float rx, ry, rz, tx, ty, tz;
    __device__ __forceinline__ void init(const float* m, float x, float y) {
        rx = m[0] * x + m[1] * y + m[2];
        ry = m[3] * x + m[4] * y + m[5];
        rz = m[6] * x + m[7] * y + m[8];
        tx = m[9]; ty = m[10]; tz = m[11];
    }
};

struct SampB {
    int x0, y0;
    float w[4];      // tap weights (00, 01, 10, 11), zero for taps outside the image / non-finite projections
};

__device__ __forceinline__ SampB project_b(const RayB& r, float depth, int Hs, int Ws) {
    const float px = r.rx * depth + r.tx, py = r.ry * depth + r.ty;
    float pz = r.rz * depth + r.tz;
    if (pz == 0.0f) pz += 1e-8f;
    const float u = px / pz, v = py / pz;
    const bool fin = fabsf(u) < 1.0e9f && fabsf(v) < 1.0e9f;
    const float fx = floorf(u), fy = floorf(v);
    SampB s;
    s.x0 = fin ? (int)fx : -4;
    s.y0 = fin ? (int)fy : -4;
    const float wx1 = u - fx, wy1 = v - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
    const bool xa = s.x0 >= 0 && s.x0 < Ws, xb = s.x0 + 1 >= 0 && s.x0 + 1 < Ws;
    const bool ya = s.y0 >= 0 && s.y0 < Hs, yb = s.y0 + 1 >= 0 && s.y0 + 1 < Hs;
    s.w[0] = (fin && xa && ya) ? wx0 * wy0 : 0.0f;
    s.w[1] = (fin && xb && ya) ? wx1 * wy0 : 0.0f;
    s.w[2] = (fin && xa && yb) ? wx0 * wy1 : 0.0f;
    s.w[3] = (fin && xb && yb) ? wx1 * wy1 : 0.0f;
    return s;
}

// running scatter accumulator for one (pixel-lane, view): flushes to grad_src when the footprint moves.  Channel j of the lane lives at
// float offset j * CS from the lane's base.
template <int CPL, int CS>
struct Scatter {
    float g[4][CPL];
    int cx, cy;
    bool dirty;
    __device__ __forceinline__ void reset() {
        cx = -0x40000000; cy = -0x40000000; dirty = false;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < CPL; ++j) g[t][j] = 0.0f;
    }
    __device__ __forceinline__ void flush(float* gview, int Hs, int Ws, int C) {
        if (!dirty) return;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int x = cx + (t & 1), y = cy + (t >> 1);
            if (x < 0 || x >= Ws || y < 0 || y >= Hs) continue;     // padding taps carry zero weight anyway
            float* p = gview + ((long)y * Ws + x) * C;
#pragma unroll
            for (int j = 0; j < CPL; ++j)
                if (g[t][j] != 0.0f) atomicAdd(p + j * CS, g[t][j]);
        }
        dirty = false;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < CPL; ++j) g[t][j] = 0.0f;
    }
};

// one hypothesis of one view: accumulate grad_ref (registers) and the scatter accumulator.  gc[j] = the cost gradient of channel j's group
template <int CPL, int CS>
__device__ __forceinline__ void bwd_sample(const float* view, float* gview, const SampB& s, const float (&gc)[CPL], const float (&refv)[CPL],
                                           float (&gref)[CPL], Scatter<CPL, CS>& sc, int Hs, int Ws, int C) {
    if (s.x0 != sc.cx || s.y0 != sc.cy) {
        sc.flush(gview, Hs, Ws, C);
        sc.cx = s.x0;
        sc.cy = s.y0;
    }
    const int xa = min(max(s.x0, 0), Ws - 1), xb = min(max(s.x0 + 1, 0), Ws - 1);
    const int ya = min(max(s.y0, 0), Hs - 1), yb = min(max(s.y0 + 1, 0), Hs - 1);
    const float* t0 = view + ((long)ya * Ws + xa) * C;
    const float* t1 = view + ((long)ya * Ws + xb) * C;
    const float* t2 = view + ((long)yb * Ws + xa) * C;
    const float* t3 = view + ((long)yb * Ws + xb) * C;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const float smp = t0[j * CS] * s.w[0] + t1[j * CS] * s.w[1] + t2[j * CS] * s.w[2] + t3[j * CS] * s.w[3];
        gref[j] = fmaf(gc[j], smp, gref[j]);
        const float gr = gc[j] * refv[j];
#pragma unroll
        for (int t = 0; t < 4; ++t) sc.g[t][j] = fmaf(gr, s.w[t], sc.g[t][j]);
    }
    sc.dirty = true;
}

// lane `sub` of a pixel's LPP lanes: float offset of its channel 0 inside a texel, stride between its channels, and channel j's index
template <int C, int CPL, bool IL>
struct LaneCh {
    static constexpr int LPP = C / CPL, CS = IL ? LPP : 1;
    static __device__ __forceinline__ int base(int sub) { return IL ? sub : sub * CPL; }
    static __device__ __forceinline__ int ch(int sub, int j) { return base(sub) + j * CS; }
};

// ------------------------------------------------------------------------------------------
// backward of dmvs_warp_corr_init_f32.   gcor [B,S,G,D,H,W] -> gref [B,H,W,C] (written), gsrc [S,B,Hs,Ws,C] (+=)
template <int C, int CPL, bool IL>
__global__ void __launch_bounds__(DMVS_BLOCK)
warp_corr_init_bwd_kernel(const float* __restrict__ ref, const float* __restrict__ src, const float* __restrict__ rt,
                          const float* __restrict__ disp_min, const float* __restrict__ disp_max,
                          const float* __restrict__ gcor, float* __restrict__ gref, float* __restrict__ gsrc, int B, int S,
                          int D, int H, int W, int Hs, int Ws) {
    using L = LaneCh<C, CPL, IL>;
    constexpr int G = 4, LPP = C / CPL, PPB = DMVS_BLOCK / LPP, CS = L::CS;
    const int sub = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    const long npix = (long)B * H * W;
    const long pix = (long)blockIdx.x * PPB + slot;
    if (pix >= npix) return;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    const long hw = (long)H * W, yx = (long)y * W + x;
    const float inv_cg = 1.0f / (float)(C / G);
    float refv[CPL], gr[CPL];
    int gj[CPL];                       // correlation group of the lane's channel j (the same for every j unless interleaved)
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        refv[j] = ref[pix * C + L::ch(sub, j)] * inv_cg;
        gr[j] = 0.0f;
        gj[j] = L::ch(sub, j) / (C / G);
    }
    const float dmin = disp_min[b], dmax = disp_max[b], dm1 = (float)(D - 1);
    for (int s = 0; s < S; ++s) {
        RayB ray;
        ray.init(rt + ((long)b * S + s) * 12, (float)x, (float)y);
        const long voff = ((long)s * B + b) * (long)Hs * Ws * C + L::base(sub);
        const float* view = src + voff;
        float* gview = gsrc + voff;
        const float* gp = gcor + (((long)b * S + s) * G * D) * hw + yx;
        Scatter<CPL, CS> sc;
        sc.reset();
        for (int d = 0; d < D; ++d) {
            const SampB sp = project_b(ray, dmvs_disp_to_depth((float)d / dm1, dmin, dmax), Hs, Ws);
            float gc[CPL];
#pragma unroll
            for (int j = 0; j < CPL; ++j) gc[j] = (IL || j == 0) ? gp[((long)gj[j] * D + d) * hw] : gc[0];
            bwd_sample<CPL, CS>(view, gview, sp, gc, refv, gr, sc, Hs, Ws, C);
        }
        sc.flush(gview, Hs, Ws, C);
    }
#pragma unroll
    for (int j = 0; j < CPL; ++j) gref[pix * C + L::ch(sub, j)] = gr[j] * inv_cg;
}

// ------------------------------------------------------------------------------------------
// backward of dmvs_getcost_f32 w.r.t. the features.  gcost [B,G*n,H,W] (contiguous) -> gref (written), gsrc (+=)
// TILED: only the 16x16 tiles the window kernel (warp_bwd_win.hip) left in d.worklist; plain launch inside a hybrid call
// (d.worklist set): only when the pre-pass said "everything here".
template <int C, int CPL, int N, bool TILED, bool IL>
__global__ void __launch_bounds__(DMVS_BLOCK) getcost_bwd_kernel(const dmvs_getcost_desc d, const float* __restrict__ gcost,
                                                                 float* __restrict__ gref, float* __restrict__ gsrc) {
    using L = LaneCh<C, CPL, IL>;
    constexpr int G = 4, LPP = C / CPL, PPB = DMVS_BLOCK / LPP, CS = L::CS;
    const int sub = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    const int H = d.H, W = d.W;
    const long hw = (long)H * W;
    int x, y, b;
    bool flat = !TILED;
    if (TILED && d.worklist[1]) {           // pre-pass verdict "every tile": plain pixel order inside this launch
        flat = true;
        if (blockIdx.x >= dmvs_ceil_div_dev((long)d.B * hw, PPB)) return;
    }
    if (!flat) {
        constexpr int T = DMVS_GETCOST_TILE, BPT = T * T / PPB;
        const unsigned nvalid = (unsigned)d.worklist[0] * BPT;
        if (blockIdx.x >= nvalid) return;
        const unsigned vb = dmvs_xcd_contiguous_block(blockIdx.x, nvalid);
        int tq = d.worklist[4 + gridDim.x / BPT + (int)(vb / BPT)];
        const int tiles_x = (W + T - 1) / T, tiles_y = (H + T - 1) / T;
        const int txi = tq % tiles_x; tq /= tiles_x;
        const int tyi = tq % tiles_y;
        b = tq / tiles_y;
        const int p = (int)(vb % BPT) * PPB + slot;
        x = txi * T + (p & (T - 1));
        y = tyi * T + p / T;
        if (x >= W || y >= H) return;
    } else {
        const long npix = (long)d.B * hw;
        const long pix0 = (long)blockIdx.x * PPB + slot;
        if (pix0 >= npix) return;
        x = (int)(pix0 % W);
        y = (int)((pix0 / W) % H);
        b = (int)(pix0 / hw);
    }
    const long yx = (long)y * W + x, pix = (long)b * hw + yx;
    const float cur_inv = d.inv_depth[pix];
    float radius = (float)(N / 2) * d.interval;
    if (d.confidence) {
        const float r0 = d.min_radius * radius, r1 = d.max_radius * radius;
        radius = r0 + (1.0f - d.confidence[pix]) * (r1 - r0);
    }
    const float lo = cur_inv - radius, hi = cur_inv + radius, step = (hi - lo) / (float)(N - 1);
    const float dmin = d.disp_min[b], dmax = d.disp_max[b];
    float depth[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float sk = (float)k * step;
        sk += lo;
        depth[k] = dmvs_disp_to_depth(fminf(fmaxf(sk, 0.0f), 1.0f), dmin, dmax);
    }
    const float inv_cg = 1.0f / (float)(C / G);
    float refv[CPL], gr[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        refv[j] = d.ref[pix * C + L::ch(sub, j)] * inv_cg;
        gr[j] = 0.0f;
    }
    const int Hv = H >> d.vw_shift, Wv = W >> d.vw_shift;
    const long vwi = (long)(y >> d.vw_shift) * Wv + (x >> d.vw_shift);
    float wsum = 1e-8f;
    for (int s = 0; s < d.S; ++s) wsum += d.view_w[((long)b * d.S + s) * Hv * Wv + vwi];
    // cost gradient of the lane's channels: one group for all of them (consecutive channels), or one per channel (interleaved)
    constexpr int NG = IL ? CPL : 1;
    float gk[NG][N];
#pragma unroll
    for (int jj = 0; jj < NG; ++jj) {
        const int g = L::ch(sub, jj) / (C / G);
#pragma unroll
        for (int k = 0; k < N; ++k) gk[jj][k] = gcost[((long)b * G * N + g * N + k) * hw + yx] / wsum;
    }
    for (int s = 0; s < d.S; ++s) {
        const float w = d.view_w[((long)b * d.S + s) * Hv * Wv + vwi];
        RayB ray;
        ray.init(d.rt + ((long)b * d.S + s) * 12, (float)x, (float)y);
        const long voff = ((long)s * d.B + b) * hw * C + L::base(sub);
        Scatter<CPL, CS> sc;
        sc.reset();
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const SampB sp = project_b(ray, depth[k], H, W);
            float gc[CPL];
#pragma unroll
            for (int j = 0; j < CPL; ++j) gc[j] = gk[IL ? j : 0][k] * w;
            bwd_sample<CPL, CS>(d.src + voff, gsrc + voff, sp, gc, refv, gr, sc, H, W, C);
        }
        sc.flush(gsrc + voff, H, W, C);
    }
#pragma unroll
    for (int j = 0; j < CPL; ++j) gref[pix * C + L::ch(sub, j)] = gr[j] * inv_cg;
}

}  // namespace

int dmvs_warp_init_bwd_win_dispatch(const float* ref, const float* src, const float* rt, const float* disp_min, const float* disp_max,
                                    const float* gcor, float* gref, float* gsrc, int B, int S, int C, int D, int H, int W, int Hs,
                                    int Ws, hipStream_t st);      // warp_init_bwd_win.hip

extern "C" int dmvs_warp_corr_init_bwd_f32(const float* ref, const float* src, const float* rt, const float* disp_min,
                                           const float* disp_max, const float* gcor, float* gref, float* gsrc, int32_t B,
                                           int32_t S, int32_t C, int32_t G, int32_t D, int32_t H, int32_t W, int32_t Hs,
                                           int32_t Ws, int32_t gather, void* stream) {
    if (G != 4 || D < 2 || !ref || !src || !rt || !gcor || !gref || !gsrc) return DMVS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (!gather && C == 48 && D <= 256)        // the model's stage 1: LDS windows (warp_init_bwd_win.hip)
        return dmvs_warp_init_bwd_win_dispatch(ref, src, rt, disp_min, disp_max, gcor, gref, gsrc, B, S, C, D, H, W, Hs, Ws, st);
    const long npix = (long)B * H * W;
    dim3 block(DMVS_BLOCK);
#define DMVS_WIB_BWD(CC, CPLL, ILL)                                                                                                       \
    hipLaunchKernelGGL((warp_corr_init_bwd_kernel<CC, CPLL, ILL>), dim3(dmvs_ceil_div(npix, DMVS_BLOCK / (CC / CPLL))), block, 0, st, ref, src, \
                       rt, disp_min, disp_max, gcor, gref, gsrc, B, S, D, H, W, Hs, Ws)
    if (gather == DMVS_BWD_GATHER_INTERLEAVED) {      // 16 lanes per pixel, lane = channel mod 16: 64-byte atomic requests
        if (C == 48) DMVS_WIB_BWD(48, 3, true);
        else if (C == 32) DMVS_WIB_BWD(32, 2, true);
        else if (C == 16) DMVS_WIB_BWD(16, 1, true);
        else return DMVS_EINVAL;
    } else if (C == 48) {
        DMVS_WIB_BWD(48, 3, false);
    } else if (C == 32) {
        DMVS_WIB_BWD(32, 4, false);
    } else if (C == 16) {
        DMVS_WIB_BWD(16, 4, false);
    } else {
        return DMVS_EINVAL;
    }
#undef DMVS_WIB_BWD
    return dmvs_launch_status();
}

template <int C, int CPL, bool IL>
static int launch_getcost_bwd(const dmvs_getcost_desc& d, const float* gcost, float* gref, float* gsrc, hipStream_t st) {
    dim3 grid(dmvs_ceil_div((long)d.B * d.H * d.W, DMVS_BLOCK / (C / CPL))), block(DMVS_BLOCK);
    if (d.n == 4) hipLaunchKernelGGL((getcost_bwd_kernel<C, CPL, 4, false, IL>), grid, block, 0, st, d, gcost, gref, gsrc);
    else if (d.n == 6) hipLaunchKernelGGL((getcost_bwd_kernel<C, CPL, 6, false, IL>), grid, block, 0, st, d, gcost, gref, gsrc);
    else return DMVS_EINVAL;
    return dmvs_launch_status();
}

template <int C, int CPL, bool IL>
static int launch_getcost_bwd_tiles(const dmvs_getcost_desc& d, const float* gcost, float* gref, float* gsrc, hipStream_t st) {
    constexpr int PPB = DMVS_BLOCK / (C / CPL), T = DMVS_GETCOST_TILE, BPT = T * T / PPB;
    const long tiles = (long)d.B * ((d.H + T - 1) / T) * ((d.W + T - 1) / T);
    dim3 grid((unsigned)(tiles * BPT)), block(DMVS_BLOCK);
    if (d.n == 4) hipLaunchKernelGGL((getcost_bwd_kernel<C, CPL, 4, true, IL>), grid, block, 0, st, d, gcost, gref, gsrc);
    else if (d.n == 6) hipLaunchKernelGGL((getcost_bwd_kernel<C, CPL, 6, true, IL>), grid, block, 0, st, d, gcost, gref, gsrc);
    else return DMVS_EINVAL;
    return dmvs_launch_status();
}

int dmvs_getcost_bwd_win_dispatch(const dmvs_getcost_desc& d, const float* gcost, float* gref, float* gsrc, hipStream_t st);

extern "C" int dmvs_getcost_bwd_f32(const dmvs_getcost_desc* dp, const float* gcost, float* gref, float* gsrc, void* stream) {
    if (!dp || !gcost || !gref || !gsrc) return DMVS_EINVAL;
    dmvs_getcost_desc d = *dp;
    if (d.G != 4 || !d.ref || !d.src || !d.rt || !d.inv_depth || !d.view_w || (d.n != 4 && d.n != 6)) return DMVS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool il = (d.tune & DMVS_TUNE_BWD_INTERLEAVED) != 0;
    if (d.worklist && (d.C == 32 || d.C == 16) && d.S <= DMVS_GETCOST_MAX_WINDOW_VIEWS) {
        // LDS-window tiles first, then one launch for the rest: the listed tiles, or (pre-pass: mostly misfits) every pixel
        if (int rc = dmvs_getcost_bwd_win_dispatch(d, gcost, gref, gsrc, st)) return rc;
        if (il) return d.C == 32 ? launch_getcost_bwd_tiles<32, 2, true>(d, gcost, gref, gsrc, st) : launch_getcost_bwd_tiles<16, 1, true>(d, gcost, gref, gsrc, st);
        return d.C == 32 ? launch_getcost_bwd_tiles<32, 4, false>(d, gcost, gref, gsrc, st)
                         : launch_getcost_bwd_tiles<16, 4, false>(d, gcost, gref, gsrc, st);
    }
    d.worklist = nullptr;
    if (il) {
        if (d.C == 48) return launch_getcost_bwd<48, 3, true>(d, gcost, gref, gsrc, st);
        if (d.C == 32) return launch_getcost_bwd<32, 2, true>(d, gcost, gref, gsrc, st);
        if (d.C == 16) return launch_getcost_bwd<16, 1, true>(d, gcost, gref, gsrc, st);
        return DMVS_EINVAL;
    }
    if (d.C == 48) return launch_getcost_bwd<48, 3, false>(d, gcost, gref, gsrc, st);
    if (d.C == 32) return launch_getcost_bwd<32, 4, false>(d, gcost, gref, gsrc, st);
    if (d.C == 16) return launch_getcost_bwd<16, 4, false>(d, gcost, gref, gsrc, st);
    return DMVS_EINVAL;
}

// ------------------------------------------------------------------------------------------
// backward of dmvs_view_aggregate_f32 (InitialCost training: the view weights come out of PixelViewWeight and
// DO require grad, module.py:539-541):  out = sum_s w_s cor_s / (1e-8 + sum_s w_s)
//   gcor[b,s,gd,p] = gout[b,gd,p] * w_s / wsum          gw[b,s,p] = sum_gd gout[b,gd,p] * (cor_s[b,gd,p] - out[b,gd,p]) / wsum
__global__ void __launch_bounds__(DMVS_BLOCK)
view_aggregate_bwd_kernel(const float* __restrict__ cor, const float* __restrict__ w, const float* __restrict__ out,
                          const float* __restrict__ gout, float* __restrict__ gcor, float* __restrict__ gw, int B, int S, int GD,
                          int HW) {
    const long i = (long)blockIdx.x * DMVS_BLOCK + threadIdx.x;      // over (b, s, p)
    if (i >= (long)B * S * HW) return;
    const int p = (int)(i % HW), s = (int)((i / HW) % S), b = (int)(i / ((long)HW * S));
    float wsum = 1e-8f;
    for (int t = 0; t < S; ++t) wsum += w[((long)b * S + t) * HW + p];
    const float ws = w[((long)b * S + s) * HW + p], inv = 1.0f / wsum;
    float acc = 0.0f;
    for (int gd = 0; gd < GD; ++gd) {
        const long oi = ((long)b * GD + gd) * HW + p, ci = (((long)b * S + s) * GD + gd) * HW + p;
        const float go = gout[oi];
        gcor[ci] = go * ws * inv;
        acc = fmaf(go, cor[ci] - out[oi], acc);
    }
    gw[i] = acc * inv;
}

extern "C" int dmvs_view_aggregate_bwd_f32(const float* cor, const float* w, const float* out, const float* gout, float* gcor,
                                           float* gw, int32_t B, int32_t S, int32_t GD, int32_t HW, void* stream) {
    if (!cor || !w || !out || !gout || !gcor || !gw) return DMVS_EINVAL;
    hipLaunchKernelGGL(view_aggregate_bwd_kernel, dim3(dmvs_ceil_div((long)B * S * HW, DMVS_BLOCK)), dim3(DMVS_BLOCK), 0,
                       (hipStream_t)stream, cor, w, out, gout, gcor, gw, B, S, GD, HW);
    return dmvs_launch_status();
}
