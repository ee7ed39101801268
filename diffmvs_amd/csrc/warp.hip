// Homography warp + group-wise correlation, the hot path of DiffMVS
// (reference models/module.py:181-218 differentiable_warping, :514-548 / :630-661).
//
// CDNA4 mapping ("texel-coalesced, batch-fetched"):
//  * features are channel-last (NHWC): one bilinear tap is one contiguous C-vector
//    (192/128/64 B for C = 48/32/16).  A pixel is owned by LPP = C/CPL adjacent lanes, each
//    holding CPL = 3|4 channels, so the LPP lanes of a pixel fetch one whole texel with a
//    single coalesced request; a 64-lane wave serves 4/8/16 pixels.
//  * the projection of a hypothesis (p = R*(x,y,1)*depth + t, perspective divide, floor,
//    bilinear weights) is computed ONCE per pixel, by the lane whose index in the pixel group
//    equals the hypothesis index, and broadcast to the group with wave shuffles -- not
//    recomputed by every channel lane.
//  * the 4 taps of ALL hypotheses of a batch (NB <= 8) are requested back to back before the
//    first one is consumed: one memory latency per batch instead of one per hypothesis.  The
//    loads are branch-free (clamped offset + select); a hypothesis whose 2x2 footprint equals
//    its predecessor's (consecutive hypotheses walk the epipolar line in sub-texel steps)
//    degenerates to a load of the view's first texel -- one extra cache line for the whole
//    wave -- and re-uses the predecessor's registers.
//  * the channel reduction of a correlation group (C/G = 12/8/4 channels = LPG = 4/2/1 lanes)
//    is a wave shuffle tree; no LDS, no atomics, every output written exactly once.
//
// Semantics kept from the reference: per-tap zero padding with align_corners=True pixel
// coordinates, NO behind-camera mask, z == 0 -> z + 1e-8, non-finite coordinates sample 0.
#include "dmvs_common.h"

namespace {

struct Ray {   // p(depth) = rot * (x, y, 1) * depth + trans   (reference :199-205)
    float rx, ry, rz, tx, ty, tz;
    __device__ __forceinline__ void init(const float* m, float x, float y) {
        rx = m[0] * x + m[1] * y + m[2];
        ry = m[3] * x + m[4] * y + m[5];
        rz = m[6] * x + m[7] * y + m[8];
        tx = m[9];
        ty = m[10];
        tz = m[11];
    }
};

struct Samp {   // where a hypothesis lands in the source view, as broadcast to the pixel's lanes
    int x0, y0;                  // top-left texel of the 2x2 footprint (0,0 with zero weights if not finite)
    float w00, w01, w10, w11;    // bilinear tap weights, already zeroed for taps that fall outside (zero padding)
};

// projection + perspective divide + bilinear weights, done by ONE lane per (pixel, hypothesis)
__device__ __forceinline__ Samp project(const Ray& r, float depth, int Hs, int Ws) {
    const float px = r.rx * depth + r.tx;
    const float py = r.ry * depth + r.ty;
    float pz = r.rz * depth + r.tz;
    if (pz == 0.0f) pz += 1e-8f;
    const float u = px / pz, v = py / pz;
    // |u|,|v| < 1e9 is false for NaN / inf; beyond +-2 texels of the image every tap is padding anyway
    const bool fin = fabsf(u) < 1.0e9f && fabsf(v) < 1.0e9f;
    const float fx = floorf(u), fy = floorf(v);
    const int x0 = fin ? (int)fx : -4, y0 = fin ? (int)fy : -4;
    const float wx1 = u - fx, wy1 = v - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
    const bool xa = x0 >= 0 && x0 < Ws, xb = x0 + 1 >= 0 && x0 + 1 < Ws;
    const bool ya = y0 >= 0 && y0 < Hs, yb = y0 + 1 >= 0 && y0 + 1 < Hs;
    Samp s;
    s.x0 = x0;
    s.y0 = y0;
    s.w00 = (fin && xa && ya) ? wx0 * wy0 : 0.0f;
    s.w01 = (fin && xb && ya) ? wx1 * wy0 : 0.0f;
    s.w10 = (fin && xa && yb) ? wx0 * wy1 : 0.0f;
    s.w11 = (fin && xb && yb) ? wx1 * wy1 : 0.0f;
    return s;
}

template <int CPL>
struct Tex {   // this lane's CPL-channel slice of the 4 taps: [tap][channel]
    float v[4][CPL];
};

// one lane's slice of a texel: a single 16-byte (CPL=4) or 12-byte (CPL=3) load.  `base` is the
// wave-uniform tensor base (SGPR pair), `byte_off` a 32-bit per-lane offset -> saddr + voffset form.
template <int CPL>
__device__ __forceinline__ void load_slice(const char* base, unsigned byte_off, float (&v)[CPL]) {
    if constexpr (CPL == 4) {
        const float4 q = *reinterpret_cast<const float4*>(base + byte_off);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
        const float3 q = *reinterpret_cast<const float3*>(base + byte_off);
        v[0] = q.x; v[1] = q.y; v[2] = q.z;
    }
}

// fetch of the 2x2 footprint: taps are clamped into the image (their weights are already zero when
// they were outside), so every address is valid and no value needs masking
template <int CPL>
__device__ __forceinline__ void fetch4(const char* base, unsigned view_off, int x0, int y0, int Hs, int Ws, int C,
                                       Tex<CPL>& t) {
    const int xa = min(max(x0, 0), Ws - 1), xb = min(max(x0 + 1, 0), Ws - 1);
    const int ya = min(max(y0, 0), Hs - 1), yb = min(max(y0 + 1, 0), Hs - 1);
    const unsigned ra = view_off + (unsigned)(ya * Ws) * (unsigned)(C * 4), rb = view_off + (unsigned)(yb * Ws) * (unsigned)(C * 4);
    const unsigned ca = (unsigned)xa * (unsigned)(C * 4), cb = (unsigned)xb * (unsigned)(C * 4);
    load_slice<CPL>(base, ra + ca, t.v[0]);
    load_slice<CPL>(base, ra + cb, t.v[1]);
    load_slice<CPL>(base, rb + ca, t.v[2]);
    load_slice<CPL>(base, rb + cb, t.v[3]);
}

template <int CPL>
__device__ __forceinline__ float bilinear_dot(const Tex<CPL>& t, float w00, float w01, float w10, float w11,
                                              const float (&refv)[CPL]) {
    float dot = 0.0f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const float smp = t.v[0][j] * w00 + t.v[1][j] * w01 + t.v[2][j] * w10 + t.v[3][j] * w11;
        dot = fmaf(smp, refv[j], dot);
    }
    return dot;
}

template <int LPG>
__device__ __forceinline__ float group_reduce(float v) {
#pragma unroll
    for (int o = LPG / 2; o > 0; o >>= 1) v += __shfl_down(v, o, LPG);
    return v;
}

// Evaluate NB hypotheses of one view for this lane's pixel.  own[j] holds the projection of
// hypothesis (sub + j*LPP) computed by this lane; results (group sums, valid on the group's
// first lane) go to dots[0..NB).  cur / (px0,py0) carry the footprint cache across batches.
// A hypothesis whose footprint equals its predecessor's issues no loads at all (exec-masked), so the
// texture path only sees distinct footprints; all fetches of the batch are in flight together.
template <int CPL, int LPP, int LPG, int NB, int KPL, int K0 = 0>
__device__ __forceinline__ void eval_batch(const char* base, unsigned view_off, const Samp (&own)[KPL], int Hs, int Ws,
                                           int C, const float (&refv)[CPL], Tex<CPL>& cur, int& px0, int& py0,
                                           float (&dots)[NB]) {
    int sx[NB], sy[NB];
    bool fresh[NB];
    Tex<CPL> t[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        sx[k] = __shfl(own[(K0 + k) / LPP].x0, (K0 + k) % LPP, LPP);
        sy[k] = __shfl(own[(K0 + k) / LPP].y0, (K0 + k) % LPP, LPP);
        const int qx = k == 0 ? px0 : sx[k - 1], qy = k == 0 ? py0 : sy[k - 1];
        fresh[k] = sx[k] != qx || sy[k] != qy;
        if (fresh[k]) fetch4<CPL>(base, view_off, sx[k], sy[k], Hs, Ws, C, t[k]);
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        // the tap weights are broadcast only now, while the fetches are in flight: 4 fewer live
        // registers per hypothesis during the fetch phase
        const Samp& o = own[(K0 + k) / LPP];
        const float w00 = __shfl(o.w00, (K0 + k) % LPP, LPP), w01 = __shfl(o.w01, (K0 + k) % LPP, LPP);
        const float w10 = __shfl(o.w10, (K0 + k) % LPP, LPP), w11 = __shfl(o.w11, (K0 + k) % LPP, LPP);
        if (fresh[k]) cur = t[k];
        dots[k] = group_reduce<LPG>(bilinear_dot<CPL>(cur, w00, w01, w10, w11, refv));
    }
    px0 = sx[NB - 1];
    py0 = sy[NB - 1];
}

// ------------------------------------------------------------------------------------------
// InitialCost volumes: grid = (pixel blocks, S).  out [B,S,G,D,H,W].  D is a multiple of NB.
template <int C, int CPL, int NB>
__global__ void __launch_bounds__(DMVS_BLOCK)
warp_corr_init_kernel(const float* __restrict__ ref, const float* __restrict__ src, const float* __restrict__ rt,
                      const float* __restrict__ disp_min, const float* __restrict__ disp_max,
                      float* __restrict__ out, int B, int S, int D, int H, int W, int Hs, int Ws) {
    constexpr int G = 4, LPP = C / CPL, LPG = LPP / G, PPB = DMVS_BLOCK / LPP;
    static_assert(NB <= LPP, "one projection per lane and batch");
    const int sub = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    const long npix = (long)B * H * W;
    const long pix = (long)dmvs_xcd_contiguous_block(blockIdx.x, gridDim.x) * PPB + slot;
    const bool live = pix < npix;
    const long pc = live ? pix : npix - 1;
    const int x = (int)(pc % W);
    const int y = (int)((pc / W) % H);
    const int b = (int)(pc / ((long)W * H));
    const int s = blockIdx.y;

    float refv[CPL];
    const float inv_cg = 1.0f / (float)(C / G);   // mean over the channels of a group
#pragma unroll
    for (int j = 0; j < CPL; ++j) refv[j] = ref[pc * C + sub * CPL + j] * inv_cg;

    Ray ray;
    ray.init(rt + ((long)b * S + s) * 12, (float)x, (float)y);
    const char* base = reinterpret_cast<const char*>(src);
    const unsigned view_off = (unsigned)((((long)s * B + b) * (long)Hs * Ws * C + sub * CPL) * 4);
    const float dmin = disp_min[b], dmax = disp_max[b];
    const float dm1 = (float)(D - 1);
    const int g = sub / LPG;
    const long hw = (long)H * W;
    float* op = out + ((((long)b * S + s) * G + g) * D) * hw + (long)y * W + x;
    const bool writer = live && (sub % LPG) == 0;

    Tex<CPL> cur;
#pragma unroll
    for (int tap = 0; tap < 4; ++tap)
#pragma unroll
        for (int j = 0; j < CPL; ++j) cur.v[tap][j] = 0.0f;
    int px0 = -0x40000000, py0 = -0x40000000;
    for (int d0 = 0; d0 < D; d0 += NB) {
        Samp own[1];
        const int dk = d0 + (sub < NB ? sub : 0);
        own[0] = project(ray, dmvs_disp_to_depth((float)dk / dm1, dmin, dmax), Hs, Ws);
        float dots[NB];
        eval_batch<CPL, LPP, LPG, NB, 1>(base, view_off, own, Hs, Ws, C, refv, cur, px0, py0, dots);
        if (writer) {
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (d0 + k < D) op[(long)(d0 + k) * hw] = dots[k];
        }
    }
}

// ------------------------------------------------------------------------------------------
// GetCost: hypotheses + S warps + correlation + view-weighted aggregation in one pass.
// TILED: the launch covers the 16x16 pixel tiles that the LDS-window kernel (warp_win.hip) could not take (or, on the
// pre-pass' "every tile" verdict, all pixels in plain order), listed in
// d.worklist (layout: warp_win.hip; [0] = count, [1] = "every tile" flag of the pre-pass); workgroups beyond the count retire.
template <int C, int CPL, int N, bool TILED>
__global__ void __launch_bounds__(DMVS_BLOCK) getcost_kernel(const dmvs_getcost_desc d) {
    constexpr int G = 4, LPP = C / CPL, LPG = LPP / G, PPB = DMVS_BLOCK / LPP, KPL = (N + LPP - 1) / LPP;
    const int sub = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    const int H = d.H, W = d.W;
    const long hw = (long)H * W;
    bool live;
    int x, y, b;
    bool flat = !TILED;
    unsigned nflat = gridDim.x;
    if (TILED && d.worklist[1]) {           // pre-pass verdict "every tile": this launch degenerates to the plain pixel order
        flat = true;
        nflat = dmvs_ceil_div_dev((long)d.B * hw, PPB);
        if (blockIdx.x >= nflat) return;
    }
    if (!flat) {
        constexpr int T = DMVS_GETCOST_TILE, BPT = T * T / PPB;       // workgroups per tile
        // the first count*BPT workgroups carry the work; each XCD (blockIdx % 8) walks a contiguous run of listed tiles
        const unsigned nvalid = (unsigned)d.worklist[0] * BPT;
        if (blockIdx.x >= nvalid) return;
        const unsigned vb = dmvs_xcd_contiguous_block(blockIdx.x, nvalid);
        const int entry = (int)(vb / BPT);
        int tq = d.worklist[4 + gridDim.x / BPT + entry];      // list[] follows flags[ntiles]
        const int tiles_x = (W + T - 1) / T, tiles_y = (H + T - 1) / T;
        const int txi = tq % tiles_x; tq /= tiles_x;
        const int tyi = tq % tiles_y;
        b = tq / tiles_y;
        const int p = (int)(vb % BPT) * PPB + slot;
        x = txi * T + (p & (T - 1));
        y = tyi * T + p / T;
        live = x < W && y < H;
        x = min(x, W - 1);
        y = min(y, H - 1);
    } else {
        const long npix = (long)d.B * hw;
        const long pix = (long)dmvs_xcd_contiguous_block(blockIdx.x, nflat) * PPB + slot;
        live = pix < npix;
        const long pq = live ? pix : npix - 1;
        x = (int)(pq % W);
        y = (int)((pq / W) % H);
        b = (int)(pq / hw);
    }
    const long yx = (long)y * W + x, pc = (long)b * hw + yx;

    // hypotheses in normalised inverse depth (reference :259-276)
    const float cur_inv = d.inv_depth[pc];
    float radius = (float)(N / 2) * d.interval;
    if (d.confidence) {
        const float r0 = d.min_radius * radius, r1 = d.max_radius * radius;
        radius = r0 + (1.0f - d.confidence[pc]) * (r1 - r0);
    }
    const float lo = cur_inv - radius, hi = cur_inv + radius;
    const float step = (hi - lo) / (float)(N - 1);
    const float dmin = d.disp_min[b], dmax = d.disp_max[b];
    // this lane projects hypotheses sub, sub+LPP, ...: it only needs those depths
    float own_depth[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const int k = sub + j * LPP;
        float sk = (float)(k < N ? k : 0) * step;
        sk += lo;
        sk = fminf(fmaxf(sk, 0.0f), 1.0f);
        own_depth[j] = dmvs_disp_to_depth(sk, dmin, dmax);
        if (live && k < N) d.out_samples[((long)b * d.samp_cstride + d.samp_coffset + k) * hw + yx] = sk;
    }

    float refv[CPL];
    const float inv_cg = 1.0f / (float)(C / G);
#pragma unroll
    for (int j = 0; j < CPL; ++j) refv[j] = d.ref[pc * C + sub * CPL + j] * inv_cg;

    float acc[N];
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = 0.0f;
    float wsum = 1e-8f;
    const int Hv = H >> d.vw_shift, Wv = W >> d.vw_shift;
    const long vwi = (long)(y >> d.vw_shift) * Wv + (x >> d.vw_shift);
    for (int s = 0; s < d.S; ++s) {
        const float w = d.view_w[((long)b * d.S + s) * Hv * Wv + vwi];
        wsum += w;
        Ray ray;
        ray.init(d.rt + ((long)b * d.S + s) * 12, (float)x, (float)y);
        const unsigned view_off = (unsigned)((((long)s * d.B + b) * hw * C + sub * CPL) * 4);
        Samp own[KPL];
#pragma unroll
        for (int j = 0; j < KPL; ++j) own[j] = project(ray, own_depth[j], H, W);
        Tex<CPL> cur;
#pragma unroll
        for (int tap = 0; tap < 4; ++tap)
#pragma unroll
            for (int j = 0; j < CPL; ++j) cur.v[tap][j] = 0.0f;
        int px0 = -0x40000000, py0 = -0x40000000;
        // two fetch batches of N/2 hypotheses: half the texel registers in flight -> one more wave per SIMD
        constexpr int NH = N / 2;
        float dots[NH];
        eval_batch<CPL, LPP, LPG, NH, KPL, 0>(reinterpret_cast<const char*>(d.src), view_off, own, H, W, C, refv, cur, px0, py0, dots);
#pragma unroll
        for (int k = 0; k < NH; ++k) acc[k] = fmaf(w, dots[k], acc[k]);
        eval_batch<CPL, LPP, LPG, NH, KPL, NH>(reinterpret_cast<const char*>(d.src), view_off, own, H, W, C, refv, cur, px0, py0, dots);
#pragma unroll
        for (int k = 0; k < NH; ++k) acc[NH + k] = fmaf(w, dots[k], acc[NH + k]);
    }
    if (live && (sub % LPG) == 0) {
        const int g = sub / LPG;
#pragma unroll
        for (int k = 0; k < N; ++k)
            d.out_cost[((long)b * d.cost_cstride + d.cost_coffset + g * N + k) * hw + yx] = acc[k] / wsum;
    }
}

template <int C, int CPL>
int launch_getcost(const dmvs_getcost_desc& d, hipStream_t st) {
    constexpr int PPB = DMVS_BLOCK / (C / CPL);
    dim3 grid(dmvs_ceil_div((long)d.B * d.H * d.W, PPB)), block(DMVS_BLOCK);
    if (d.n == 4) hipLaunchKernelGGL((getcost_kernel<C, CPL, 4, false>), grid, block, 0, st, d);
    else if (d.n == 6) hipLaunchKernelGGL((getcost_kernel<C, CPL, 6, false>), grid, block, 0, st, d);
    else return DMVS_EINVAL;
    return dmvs_launch_status();
}

// the tiles the window kernel listed in d.worklist
template <int C, int CPL>
int launch_getcost_tiles(const dmvs_getcost_desc& d, hipStream_t st) {
    constexpr int PPB = DMVS_BLOCK / (C / CPL), T = DMVS_GETCOST_TILE, BPT = T * T / PPB;
    const long tiles = (long)d.B * ((d.H + T - 1) / T) * ((d.W + T - 1) / T);
    dim3 grid((unsigned)(tiles * BPT)), block(DMVS_BLOCK);
    if (d.n == 4) hipLaunchKernelGGL((getcost_kernel<C, CPL, 4, true>), grid, block, 0, st, d);
    else if (d.n == 6) hipLaunchKernelGGL((getcost_kernel<C, CPL, 6, true>), grid, block, 0, st, d);
    else return DMVS_EINVAL;
    return dmvs_launch_status();
}

template <int C, int CPL, int NB>
int launch_init(const float* ref, const float* src, const float* rt, const float* disp_min, const float* disp_max,
                float* out, int B, int S, int D, int H, int W, int Hs, int Ws, hipStream_t st) {
    dim3 grid(dmvs_ceil_div((long)B * H * W, DMVS_BLOCK / (C / CPL)), S), block(DMVS_BLOCK);
    hipLaunchKernelGGL((warp_corr_init_kernel<C, CPL, NB>), grid, block, 0, st, ref, src, rt, disp_min, disp_max, out, B, S,
                       D, H, W, Hs, Ws);
    return dmvs_launch_status();
}

}  // namespace

int dmvs_warp_init_win_dispatch(const float* ref, const float* src, const float* rt, const float* disp_min, const float* disp_max,
                                float* out, int B, int S, int C, int D, int H, int W, int Hs, int Ws, hipStream_t st);   // warp_init_win.hip

// per-pixel gather through the texture path (every C)
extern "C" int dmvs_warp_corr_init_gather_f32(const float* ref, const float* src, const float* rt, const float* disp_min,
                                              const float* disp_max, float* out, int32_t B, int32_t S, int32_t C, int32_t G,
                                              int32_t D, int32_t H, int32_t W, int32_t Hs, int32_t Ws, void* stream) {
    if (G != 4 || D < 2 || !ref || !src || !rt || !out) return DMVS_EINVAL;
    if ((long)S * B * Hs * Ws * C * 4 >= (1L << 32)) return DMVS_EINVAL;   // 32-bit byte offsets over the source stack
    hipStream_t st = (hipStream_t)stream;
    // hypotheses per fetch batch: 8 texel sets in flight per lane (4 for the 4-lane C=16 pixel group)
    if (C == 48) return launch_init<48, 3, 8>(ref, src, rt, disp_min, disp_max, out, B, S, D, H, W, Hs, Ws, st);
    if (C == 32) return launch_init<32, 4, 8>(ref, src, rt, disp_min, disp_max, out, B, S, D, H, W, Hs, Ws, st);
    if (C == 16) return launch_init<16, 4, 4>(ref, src, rt, disp_min, disp_max, out, B, S, D, H, W, Hs, Ws, st);
    return DMVS_EINVAL;
}

extern "C" int dmvs_warp_corr_init_f32(const float* ref, const float* src, const float* rt, const float* disp_min,
                                       const float* disp_max, float* out, int32_t B, int32_t S, int32_t C, int32_t G,
                                       int32_t D, int32_t H, int32_t W, int32_t Hs, int32_t Ws, void* stream) {
    if (G != 4 || D < 2 || !ref || !src || !rt || !out) return DMVS_EINVAL;
    if (C == 48 && D <= 256)           // the model's stage 1: LDS-staged source windows (warp_init_win.hip)
        return dmvs_warp_init_win_dispatch(ref, src, rt, disp_min, disp_max, out, B, S, C, D, H, W, Hs, Ws, (hipStream_t)stream);
    return dmvs_warp_corr_init_gather_f32(ref, src, rt, disp_min, disp_max, out, B, S, C, G, D, H, W, Hs, Ws, stream);
}

int dmvs_getcost_win_dispatch(const dmvs_getcost_desc& d, hipStream_t st);   // warp_win.hip

static int getcost_check(const dmvs_getcost_desc& d) {
    if (d.G != 4 || !d.ref || !d.src || !d.rt || !d.inv_depth || !d.view_w || !d.out_cost || !d.out_samples)
        return DMVS_EINVAL;
    if (d.n != 4 && d.n != 6) return DMVS_EINVAL;
    if ((long)d.S * d.B * d.H * d.W * d.C * 4 >= (1L << 32)) return DMVS_EINVAL;   // 32-bit byte offsets over the source stack
    return 0;
}

// per-pixel gather through the texture path (every C; the only variant for C = 48)
extern "C" int dmvs_getcost_gather_f32(const dmvs_getcost_desc* dp, void* stream) {
    if (!dp) return DMVS_EINVAL;
    const dmvs_getcost_desc& d = *dp;
    if (int rc = getcost_check(d)) return rc;
    hipStream_t st = (hipStream_t)stream;
    dmvs_getcost_desc g = d;
    g.worklist = nullptr;          // plain launch, unconditional
    if (g.C == 48) return launch_getcost<48, 3>(g, st);
    if (g.C == 32) return launch_getcost<32, 4>(g, st);
    if (g.C == 16) return launch_getcost<16, 4>(g, st);
    return DMVS_EINVAL;
}

extern "C" int dmvs_getcost_f32(const dmvs_getcost_desc* dp, void* stream) {
    if (!dp) return DMVS_EINVAL;
    const dmvs_getcost_desc& d = *dp;
    if (int rc = getcost_check(d)) return rc;
    if ((d.C != 32 && d.C != 16) || !d.worklist || d.S > DMVS_GETCOST_MAX_WINDOW_VIEWS) return dmvs_getcost_gather_f32(dp, stream);
    hipStream_t st = (hipStream_t)stream;
    if (int rc = dmvs_getcost_win_dispatch(d, st)) return rc;          // pre-pass + tiles whose source windows fit LDS
    // the rest, one launch: the listed tiles, or (pre-pass: most tiles do not fit) every pixel in plain order
    return d.C == 32 ? launch_getcost_tiles<32, 4>(d, st) : launch_getcost_tiles<16, 4>(d, st);
}
