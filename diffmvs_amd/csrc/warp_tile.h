// Shared by the tile-per-workgroup GetCost kernels (warp_win.hip forward, warp_bwd_win.hip backward): the tile / window
// geometry, the per-pixel projection helpers, the footprint-box reduction and the atomic-free pre-pass with its scratch
// layout.  Everything is file-local to the including translation unit (anonymous namespace).
#pragma once
#include "dmvs_common.h"

namespace {

typedef float f2 __attribute__((vector_size(8)));

constexpr int TW = 16, TH = 16;      // reference-pixel tile of a workgroup (one lane per pixel)
constexpr int WW = 24;               // window width in texels
template <int C> struct WinCfg { static constexpr int WH = C == 32 ? 22 : 24; };   // rows: 76 KB (C=32) / 46 KB (C=16)

#define DMVS_LDS3(p) ((__attribute__((address_space(3))) void*)(p))

struct RayW {
    float rx, ry, rz, tx, ty, tz;
    __device__ __forceinline__ void init(const float* m, float x, float y) {
        rx = m[0] * x + m[1] * y + m[2];
        ry = m[3] * x + m[4] * y + m[5];
        rz = m[6] * x + m[7] * y + m[8];
        tx = m[9]; ty = m[10]; tz = m[11];
    }
};

struct SampW {
    int x0, y0;
    float w00, w01, w10, w11;
};

__device__ __forceinline__ void project_uv(const RayW& r, float depth, float& u, float& v, float& pz, bool& fin) {
    const float px = r.rx * depth + r.tx, py = r.ry * depth + r.ty;
    pz = r.rz * depth + r.tz;
    if (pz == 0.0f) pz += 1e-8f;
    // one reciprocal (hardware estimate + one Newton step: within an ulp of the IEEE quotient) shared by u and v
    float inv = __builtin_amdgcn_rcpf(pz);
    inv = fmaf(fmaf(-pz, inv, 1.0f), inv, inv);
    u = px * inv;
    v = py * inv;
    fin = fabsf(u) < 1.0e9f && fabsf(v) < 1.0e9f;
}

__device__ __forceinline__ SampW make_samp(float u, float v, bool fin, int Hs, int Ws) {
    const float fx = floorf(u), fy = floorf(v);
    SampW s;
    s.x0 = fin ? (int)fx : -4;          // -4: all four taps fail the range tests below
    s.y0 = fin ? (int)fy : -4;
    const float wx1 = u - fx, wy1 = v - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
    // zero padding folded into the 1-D weights: a tap outside the image contributes nothing (grid_sample zeros)
    const float ax0 = (unsigned)s.x0 < (unsigned)Ws ? wx0 : 0.0f, ax1 = (unsigned)(s.x0 + 1) < (unsigned)Ws ? wx1 : 0.0f;
    const float ay0 = (unsigned)s.y0 < (unsigned)Hs ? wy0 : 0.0f, ay1 = (unsigned)(s.y0 + 1) < (unsigned)Hs ? wy1 : 0.0f;
    s.w00 = ax0 * ay0;
    s.w01 = ax1 * ay0;
    s.w10 = ax0 * ay1;
    s.w11 = ax1 * ay1;
    return s;
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// Footprint boxes of one 16x16 tile in every view, from the two end hypotheses of each pixel (the projection of a
// depth interval is a straight image segment); returns whether all of them fit a win_w x win_h texel window.
// Workgroup-collective (one barrier per view).  sbox == nullptr: only the verdict is wanted.
__device__ __forceinline__ bool tile_boxes(const dmvs_getcost_desc& d, int b, int xc, int yc, bool live, float depth_first,
                                           float depth_last, int (*red)[DMVS_BLOCK / 64][5], int (*sbox)[4], int max_s, int win_w,
                                           int win_h) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = d.H, W = d.W;
    bool allfit = d.S <= max_s;
    for (int s = 0; s < d.S && s < max_s; ++s) {
        RayW ray;
        ray.init(d.rt + ((long)b * d.S + s) * 12, (float)xc, (float)yc);
        float u0, v0, z0, u1, v1, z1;
        bool f0, f1;
        project_uv(ray, depth_first, u0, v0, z0, f0);
        project_uv(ray, depth_last, u1, v1, z1, f1);
        int bad = live && (!f0 || !f1 || ((z0 < 0.0f) != (z1 < 0.0f)));     // a pole between the ends: not a segment
        int bx0 = 0x3fffffff, by0 = 0x3fffffff, bx1 = -0x3fffffff, by1 = -0x3fffffff;
        if (live && !bad) {
            const int ax = max((int)floorf(fminf(u0, u1)), 0), cx = min((int)floorf(fmaxf(u0, u1)) + 1, W - 1);
            const int ay = max((int)floorf(fminf(v0, v1)), 0), cy = min((int)floorf(fmaxf(v0, v1)) + 1, H - 1);
            if (ax <= cx && ay <= cy) {      // else: every tap of every hypothesis of this pixel is padding
                bx0 = ax; bx1 = cx; by0 = ay; by1 = cy;
            }
        }
        bx0 = wave_min(bx0); by0 = wave_min(by0); bx1 = wave_max(bx1); by1 = wave_max(by1); bad = wave_max(bad);
        int (*rd)[5] = red[s & 1];           // double-buffered: a fast wave may already be writing the next view's
        if (lane == 0) {
            rd[wave][0] = bx0; rd[wave][1] = by0; rd[wave][2] = bx1; rd[wave][3] = by1; rd[wave][4] = bad;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < DMVS_BLOCK / 64; ++q) {
            bx0 = min(bx0, rd[q][0]); by0 = min(by0, rd[q][1]);
            bx1 = max(bx1, rd[q][2]); by1 = max(by1, rd[q][3]); bad = max(bad, rd[q][4]);
        }
        const int nc = bx1 - bx0 + 1, nr = by1 - by0 + 1;
        if (bad || (bx1 >= bx0 && (nc > win_w || nr > win_h))) allfit = false;
        if (sbox && tid == 0) {
            sbox[s][0] = bx0; sbox[s][1] = by0; sbox[s][2] = nc; sbox[s][3] = nr;
        }
    }
    return allfit;
}

// this tile's position and the two end hypotheses of the lane's pixel (reference module.py:259-276)
template <int N>
__device__ __forceinline__ void tile_pixel(const dmvs_getcost_desc& d, int tile, int tiles_x, int tiles_y, int& b, int& xc, int& yc,
                                           bool& live, float& lo, float& step) {
    int tq = tile;
    const int txi = tq % tiles_x; tq /= tiles_x;
    const int tyi = tq % tiles_y;
    b = tq / tiles_y;
    const int x = txi * TW + (threadIdx.x & (TW - 1)), y = tyi * TH + (threadIdx.x >> 4);
    live = x < d.W && y < d.H;
    xc = min(x, d.W - 1);
    yc = min(y, d.H - 1);
    const long pc = ((long)b * d.H + yc) * d.W + xc;
    const float cur_inv = d.inv_depth[pc];
    float radius = (float)(N / 2) * d.interval;
    if (d.confidence) {
        const float r0 = d.min_radius * radius, r1 = d.max_radius * radius;
        radius = r0 + (1.0f - d.confidence[pc]) * (r1 - r0);
    }
    lo = cur_inv - radius;
    step = (cur_inv + radius - lo) / (float)(N - 1);
}

__device__ __forceinline__ float hyp_depth(int k, float lo, float step, float dmin, float dmax) {
    float sk = (float)k * step;
    sk += lo;
    return dmvs_disp_to_depth(fminf(fmaxf(sk, 0.0f), 1.0f), dmin, dmax);
}

// Scratch layout (int32), n = number of tiles:  [0] tiles listed for the gather kernel, [1] mode (1 = the gather
// kernel takes every tile), [2..3] unused, flags[n] (1 = some view's footprint exceeds the window), list[n] (the
// flagged tiles, ascending), boxes[n][MAXS][4] (x0, y0, ncols, nrows per view).  No atomics anywhere: same-address
// device atomics from ~10^3 workgroups serialise at ~0.1 us each, more than the whole pre-pass costs.
constexpr int MAXS = DMVS_GETCOST_MAX_WINDOW_VIEWS;       // per-view footprint boxes kept per tile (entry points fall back to the gather kernels beyond)
__device__ __forceinline__ int* ws_flags(int* ws) { return ws + 4; }
__device__ __forceinline__ int* ws_list(int* ws, int ntiles) { return ws + 4 + ntiles; }
__device__ __forceinline__ int* ws_boxes(int* ws, int ntiles) { return ws + 4 + 2 * ntiles; }

// Pre-pass 1: every tile's per-view footprint boxes and its fit flag.
template <int N>
__global__ void __launch_bounds__(DMVS_BLOCK) getcost_fit_kernel(const dmvs_getcost_desc d, int tiles_x, int tiles_y, int win_w,
                                                                 int win_h) {
    __shared__ int red[2][DMVS_BLOCK / 64][5];
    __shared__ int sbox[MAXS][4];
    const int tile = blockIdx.x, ntiles = gridDim.x;
    int b, xc, yc;
    bool live;
    float lo, step;
    tile_pixel<N>(d, tile, tiles_x, tiles_y, b, xc, yc, live, lo, step);
    const float dmin = d.disp_min[b], dmax = d.disp_max[b];
    const bool fits = tile_boxes(d, b, xc, yc, live, hyp_depth(0, lo, step, dmin, dmax), hyp_depth(N - 1, lo, step, dmin, dmax), red,
                                 sbox, MAXS, win_w, win_h);
    __syncthreads();
    if (threadIdx.x < 4 * MAXS && threadIdx.x < 4 * d.S)
        ws_boxes(d.worklist, ntiles)[(size_t)tile * (4 * MAXS) + threadIdx.x] = sbox[threadIdx.x >> 2][threadIdx.x & 3];
    if (threadIdx.x == 0) ws_flags(d.worklist)[tile] = fits ? 0 : 1;
}

// Pre-pass 2 (one workgroup): count the flagged tiles, pick the mode, list the flagged tiles in ascending order.
// If most tiles are flagged (a depth map that is noise rather than surfaces, e.g. an untrained network) the window
// kernel stands down and the gather kernel takes every tile: the hybrid only pays off while the windows carry a good
// share of the work.
__global__ void __launch_bounds__(DMVS_BLOCK) getcost_compact_kernel(int* __restrict__ ws, int ntiles) {
    __shared__ int cnt[DMVS_BLOCK + 1];
    const int tid = threadIdx.x;
    const int seg = (ntiles + DMVS_BLOCK - 1) / DMVS_BLOCK, t0 = tid * seg, t1 = min(t0 + seg, ntiles);
    const int* flags = ws_flags(ws);
    int c = 0;
    for (int t = t0; t < t1; ++t) c += flags[t];
    cnt[tid + 1] = c;
    __syncthreads();
    if (tid == 0) {
        cnt[0] = 0;
        for (int i = 1; i <= DMVS_BLOCK; ++i) cnt[i] += cnt[i - 1];
        ws[0] = cnt[DMVS_BLOCK];
        ws[1] = (long)cnt[DMVS_BLOCK] * 4 > (long)ntiles * 3 ? 1 : 0;      // > 75 % flagged: gather everything
    }
    __syncthreads();
    int* list = ws_list(ws, ntiles);
    int o = cnt[tid];
    for (int t = t0; t < t1; ++t)
        if (flags[t]) list[o++] = t;
}

// launches the two pre-pass kernels for d (d.worklist must be set): per-tile fit flags, boxes, mode, ordered list
template <int C>
int launch_getcost_prepass(const dmvs_getcost_desc& d, hipStream_t st) {
    const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
    dim3 grid((unsigned)(tiles_x * tiles_y * d.B)), block(DMVS_BLOCK);
    if (d.n == 4) hipLaunchKernelGGL((getcost_fit_kernel<4>), grid, block, 0, st, d, tiles_x, tiles_y, WW, WinCfg<C>::WH);
    else if (d.n == 6) hipLaunchKernelGGL((getcost_fit_kernel<6>), grid, block, 0, st, d, tiles_x, tiles_y, WW, WinCfg<C>::WH);
    else return DMVS_EINVAL;
    hipLaunchKernelGGL(getcost_compact_kernel, dim3(1), block, 0, st, d.worklist, (int)grid.x);
    return dmvs_launch_status();
}

}  // namespace
