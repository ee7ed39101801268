// Homography warp + group-wise correlation, the hot path of DiffMVS
// (reference models/module.py:181-218 differentiable_warping, :514-548 / :630-661).
//
// CDNA4 mapping ("texel-coalesced"): features are channel-last (NHWC), so one bilinear tap
// is one contiguous C-vector (192/128/64 B for C = 48/32/16).  A pixel is owned by
// LPP = C/CPL adjacent lanes, each holding CPL = 3|4 channels, so the LPP lanes of a pixel
// fetch one whole texel with a single coalesced request; a 64-lane wave serves 4/8/16
// pixels.  The four texels of the current 2x2 footprint are cached in registers
// (4*CPL VGPRs) and only re-fetched when floor(u), floor(v) move -- consecutive depth
// hypotheses walk the epipolar line in sub-texel steps, so most hypotheses hit the cache.
// The channel reduction of a correlation group (C/G = 12/8/4 channels = LPG = 4/2/1 lanes)
// is a wave shuffle tree; no LDS, no atomics, outputs written once.
//
// Semantics kept from the reference: per-tap zero padding with align_corners=True pixel
// coordinates, NO behind-camera mask, z == 0 -> z + 1e-8, non-finite coordinates sample 0.
#include "dmvs_common.h"

namespace {

// one lane's CPL-channel slice of a texel: a single 16-byte (CPL=4) or 12-byte (CPL=3) load
template <int CPL>
__device__ __forceinline__ void load_texel(const float* p, bool ok, float (&v)[CPL]) {
#pragma unroll
    for (int j = 0; j < CPL; ++j) v[j] = 0.0f;
    if (ok) {
        if constexpr (CPL == 4) {
            const float4 q = *reinterpret_cast<const float4*>(p);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            const float3 q = *reinterpret_cast<const float3*>(p);
            v[0] = q.x; v[1] = q.y; v[2] = q.z;
        }
    }
}

// register-cached 2x2 footprint of one source view for one pixel-lane
template <int CPL>
struct Footprint {
    float t00[CPL], t01[CPL], t10[CPL], t11[CPL];
    int cx, cy;
    __device__ __forceinline__ void reset() { cx = -0x40000000; cy = -0x40000000; }
    // view base already offset by this lane's channel slice
    __device__ __forceinline__ void fetch(const float* view, int x0, int y0, int Hs, int Ws, int C) {
        if (x0 == cx && y0 == cy) return;
        cx = x0;
        cy = y0;
        const bool xa = x0 >= 0 && x0 < Ws, xb = x0 + 1 >= 0 && x0 + 1 < Ws;
        const bool ya = y0 >= 0 && y0 < Hs, yb = y0 + 1 >= 0 && y0 + 1 < Hs;
        const long base = ((long)y0 * Ws + x0) * C;
        load_texel<CPL>(view + base, xa && ya, t00);
        load_texel<CPL>(view + base + C, xb && ya, t01);
        load_texel<CPL>(view + base + (long)Ws * C, xa && yb, t10);
        load_texel<CPL>(view + base + (long)Ws * C + C, xb && yb, t11);
    }
};

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

// one hypothesis: project, (re)fetch the footprint, bilinear sample, dot with the reference slice
template <int CPL>
__device__ __forceinline__ float sample_dot(const Ray& r, float depth, const float* view, Footprint<CPL>& fp,
                                            const float (&refv)[CPL], int Hs, int Ws, int C) {
    const float px = r.rx * depth + r.tx;
    const float py = r.ry * depth + r.ty;
    float pz = r.rz * depth + r.tz;
    if (pz == 0.0f) pz += 1e-8f;
    const float u = px / pz, v = py / pz;
    const bool fin = fabsf(u) < 1.0e9f && fabsf(v) < 1.0e9f;   // false for NaN / inf
    const float fx = floorf(u), fy = floorf(v);
    const int x0 = fin ? (int)fx : -0x20000000, y0 = fin ? (int)fy : -0x20000000;
    fp.fetch(view, x0, y0, Hs, Ws, C);
    const float wx1 = fin ? u - fx : 0.0f, wy1 = fin ? v - fy : 0.0f;
    const float wx0 = fin ? 1.0f - wx1 : 0.0f, wy0 = 1.0f - wy1;
    const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
    float dot = 0.0f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const float s = fp.t00[j] * w00 + fp.t01[j] * w01 + fp.t10[j] * w10 + fp.t11[j] * w11;
        dot = fmaf(s, refv[j], dot);
    }
    return dot;
}

template <int LPG>
__device__ __forceinline__ float group_reduce(float v) {
#pragma unroll
    for (int o = LPG / 2; o > 0; o >>= 1) v += __shfl_down(v, o, LPG);
    return v;
}

// ------------------------------------------------------------------------------------------
// InitialCost volumes: grid = (pixel blocks, S).  out [B,S,G,D,H,W]
template <int C, int CPL>
__global__ void __launch_bounds__(DMVS_BLOCK)
warp_corr_init_kernel(const float* __restrict__ ref, const float* __restrict__ src, const float* __restrict__ rt,
                      const float* __restrict__ disp_min, const float* __restrict__ disp_max,
                      float* __restrict__ out, int B, int S, int D, int H, int W, int Hs, int Ws) {
    constexpr int G = 4, LPP = C / CPL, LPG = LPP / G, PPB = DMVS_BLOCK / LPP;
    const int sub = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    const long npix = (long)B * H * W;
    const long pix = (long)blockIdx.x * PPB + slot;
    const bool live = pix < npix;
    const long pc = live ? pix : npix - 1;
    const int x = (int)(pc % W);
    const int y = (int)((pc / W) % H);
    const int b = (int)(pc / ((long)W * H));
    const int s = blockIdx.y;

    float refv[CPL];
    const float inv_cg = 1.0f / (float)(C / G);   // mean over the channels of a group
#pragma unroll
    for (int j = 0; j < CPL; ++j) refv[j] = ref[((long)pc) * C + sub * CPL + j] * inv_cg;

    Ray ray;
    ray.init(rt + ((long)b * S + s) * 12, (float)x, (float)y);
    const float* view = src + ((long)s * B + b) * (long)Hs * Ws * C + sub * CPL;
    Footprint<CPL> fp;
    fp.reset();
    const float dmin = disp_min[b], dmax = disp_max[b];
    const float inv_dm1 = (float)(D - 1);
    const int g = sub / LPG;
    float* op = out + ((((long)b * S + s) * G + g) * D) * (long)H * W + (long)y * W + x;
    const bool writer = live && (sub % LPG) == 0;
    for (int d = 0; d < D; ++d) {
        const float depth = dmvs_disp_to_depth((float)d / inv_dm1, dmin, dmax);
        float dot = sample_dot<CPL>(ray, depth, view, fp, refv, Hs, Ws, C);
        dot = group_reduce<LPG>(dot);
        if (writer) op[(long)d * H * W] = dot;
    }
}

// ------------------------------------------------------------------------------------------
// GetCost: hypotheses + S warps + correlation + view-weighted aggregation in one pass.
template <int C, int CPL, int N>
__global__ void __launch_bounds__(DMVS_BLOCK) getcost_kernel(const dmvs_getcost_desc d) {
    constexpr int G = 4, LPP = C / CPL, LPG = LPP / G, PPB = DMVS_BLOCK / LPP;
    const int sub = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    const int H = d.H, W = d.W;
    const long npix = (long)d.B * H * W;
    const long pix = (long)blockIdx.x * PPB + slot;
    const bool live = pix < npix;
    const long pc = live ? pix : npix - 1;
    const int x = (int)(pc % W);
    const int y = (int)((pc / W) % H);
    const int b = (int)(pc / ((long)W * H));
    const long hw = (long)H * W, yx = (long)y * W + x;

    // hypotheses in normalised inverse depth (reference :259-276)
    const float cur = d.inv_depth[pc];
    float radius = (float)(N / 2) * d.interval;
    if (d.confidence) {
        const float r0 = d.min_radius * radius, r1 = d.max_radius * radius;
        radius = r0 + (1.0f - d.confidence[pc]) * (r1 - r0);
    }
    const float lo = cur - radius, hi = cur + radius;
    const float step = (hi - lo) / (float)(N - 1);
    const float dmin = d.disp_min[b], dmax = d.disp_max[b];
    float depth[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float sk = (float)k * step;
        sk += lo;
        sk = fminf(fmaxf(sk, 0.0f), 1.0f);
        depth[k] = dmvs_disp_to_depth(sk, dmin, dmax);
        if (live && sub == 0) d.out_samples[((long)b * d.samp_cstride + d.samp_coffset + k) * hw + yx] = sk;
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
        const float* view = d.src + ((long)s * d.B + b) * hw * C + sub * CPL;
        Footprint<CPL> fp;
        fp.reset();
#pragma unroll
        for (int k = 0; k < N; ++k) {
            float dot = sample_dot<CPL>(ray, depth[k], view, fp, refv, H, W, C);
            dot = group_reduce<LPG>(dot);
            acc[k] = fmaf(w, dot, acc[k]);
        }
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
    if (d.n == 4) hipLaunchKernelGGL((getcost_kernel<C, CPL, 4>), grid, block, 0, st, d);
    else if (d.n == 6) hipLaunchKernelGGL((getcost_kernel<C, CPL, 6>), grid, block, 0, st, d);
    else return DMVS_EINVAL;
    return dmvs_launch_status();
}

}  // namespace

extern "C" int dmvs_warp_corr_init_f32(const float* ref, const float* src, const float* rt, const float* disp_min,
                                       const float* disp_max, float* out, int32_t B, int32_t S, int32_t C, int32_t G,
                                       int32_t D, int32_t H, int32_t W, int32_t Hs, int32_t Ws, void* stream) {
    if (G != 4 || D < 2 || !ref || !src || !rt || !out) return DMVS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    dim3 block(DMVS_BLOCK);
    const long npix = (long)B * H * W;
    if (C == 48) {
        dim3 grid(dmvs_ceil_div(npix, DMVS_BLOCK / 16), S);
        hipLaunchKernelGGL((warp_corr_init_kernel<48, 3>), grid, block, 0, st, ref, src, rt, disp_min, disp_max, out, B, S,
                           D, H, W, Hs, Ws);
    } else if (C == 32) {
        dim3 grid(dmvs_ceil_div(npix, DMVS_BLOCK / 8), S);
        hipLaunchKernelGGL((warp_corr_init_kernel<32, 4>), grid, block, 0, st, ref, src, rt, disp_min, disp_max, out, B, S,
                           D, H, W, Hs, Ws);
    } else if (C == 16) {
        dim3 grid(dmvs_ceil_div(npix, DMVS_BLOCK / 4), S);
        hipLaunchKernelGGL((warp_corr_init_kernel<16, 4>), grid, block, 0, st, ref, src, rt, disp_min, disp_max, out, B, S,
                           D, H, W, Hs, Ws);
    } else {
        return DMVS_EINVAL;
    }
    return dmvs_launch_status();
}

extern "C" int dmvs_getcost_f32(const dmvs_getcost_desc* dp, void* stream) {
    if (!dp) return DMVS_EINVAL;
    const dmvs_getcost_desc& d = *dp;
    if (d.G != 4 || !d.ref || !d.src || !d.rt || !d.inv_depth || !d.view_w || !d.out_cost || !d.out_samples)
        return DMVS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (d.C == 48) return launch_getcost<48, 3>(d, st);
    if (d.C == 32) return launch_getcost<32, 4>(d, st);
    if (d.C == 16) return launch_getcost<16, 4>(d, st);
    return DMVS_EINVAL;
}
