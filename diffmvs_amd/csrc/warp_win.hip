// GetCost through LDS-staged source windows (reference models/module.py:583-667; same arithmetic contract as
// warp.hip's getcost_kernel, which stays as the per-pixel-gather variant for C = 48).
//
// Why: the per-pixel gather sends 4 taps x C*4 bytes per (pixel, view, distinct footprint) through the one
// texture-address unit of a CU (64 B/clk) -- ~7x the algorithmic bytes, which caps that kernel near 20 % of the
// HBM roofline.  Neighbouring reference pixels sample neighbouring source texels, so here a 16x16 pixel tile
//   1. bounds the source footprint of ALL its hypotheses from the two end hypotheses (the projection of a depth
//      interval is a straight image segment), reduces the boxes over the workgroup,
//   2. streams that window (<= 24 x 22 texels) ONCE, row-contiguous, HBM/L2 -> LDS with LDS-DMA (no VGPRs),
//   3. lets one lane own one pixel: projection computed once per (pixel, hypothesis), no cross-lane traffic at
//      all; the 2x2 taps are ds_read_b128 from the window (128 B/clk/CU, twice the TA rate, and only for a
//      footprint that differs from the previous hypothesis'), texel stride padded to C+4 floats so that
//      neighbouring lanes hit disjoint bank groups;
//   4. uses  sum_c ref_c * (sum_t w_t tex_t,c) = sum_t w_t * (sum_c ref_c tex_t,c):  one group dot per TEXEL
//      (packed fp32 FMAs), then a 4-tap blend of 4 group values per hypothesis instead of C channels.
// A tile whose footprint does not fit the window in some view (depth discontinuity, very wide baseline, a z sign
// change along a ray) is appended to d.worklist and left to the gather kernel (warp.hip, TILED launch): any geometry
// stays correct, and smooth regions -- the bulk of a real depth map -- take the fast path.
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

// group dots of one texel with the lane's reference features: Dg[g] = sum_{c in group g} ref_c * tex_c
template <int C>
__device__ __forceinline__ void texel_dots(const float* tex, const f2 (&refp)[C / 2], float (&Dg)[4]) {
    constexpr int NCH = C / 4, CPG = NCH / 4;          // 16-byte chunks per texel / per correlation group
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f2 a = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < CPG; ++j) {
            const int ch = g * CPG + j;
            const float4 q = *reinterpret_cast<const float4*>(tex + ch * 4);
            const f2 lo = {q.x, q.y}, hi = {q.z, q.w};
            a = lo * refp[2 * ch] + a;
            a = hi * refp[2 * ch + 1] + a;
        }
        Dg[g] = a[0] + a[1];
    }
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
constexpr int MAXS = 16;
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

template <int C, int N>
__global__ void __launch_bounds__(DMVS_BLOCK, 2) getcost_win_kernel(const dmvs_getcost_desc d, int tiles_x, int tiles_y) {
    constexpr int G = 4, NCH = C / 4, TS = C + 4;       // texel stride in floats (padded: bank spread)
    constexpr int WH = WinCfg<C>::WH;
    constexpr int SLOTS = WW * (NCH + 1);               // 16-byte LDS slots per window row (one pad slot per texel)
    constexpr int SUBS = (SLOTS + 63) / 64;             // DMA instructions per row and wave
    static_assert(TW * TH == DMVS_BLOCK, "one lane per pixel");
    __shared__ __attribute__((aligned(16))) float win[WW * WH * TS];
    __shared__ int sbox[MAXS][4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = d.H, W = d.W;
    const long hw = (long)H * W;
    if (d.worklist[1]) return;           // the pre-pass sent every tile to the gather kernel
    const int tile = (int)dmvs_xcd_contiguous_block(blockIdx.x, gridDim.x);
    if (ws_flags(d.worklist)[tile]) return;      // this one too
    int tq = tile;
    const int txi = tq % tiles_x; tq /= tiles_x;
    const int tyi = tq % tiles_y;
    const int b = tq / tiles_y;
    const int x = txi * TW + (tid & (TW - 1)), y = tyi * TH + (tid >> 4);
    const bool live = x < W && y < H;
    const int xc = min(x, W - 1), yc = min(y, H - 1);
    const long yx = (long)yc * W + xc, pc = (long)b * hw + yx;

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
    float depth[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float sk = (float)k * step;
        sk += lo;
        sk = fminf(fmaxf(sk, 0.0f), 1.0f);
        depth[k] = dmvs_disp_to_depth(sk, dmin, dmax);
    }

    float acc[N][G];
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[k][g] = 0.0f;
    float wsum = 1e-8f;
    const int Hv = H >> d.vw_shift, Wv = W >> d.vw_shift;
    const long vwi = (long)(yc >> d.vw_shift) * Wv + (xc >> d.vw_shift);

    // this lane's share of a window row in the DMA: slot -> (texel column, 16-byte chunk)
    int dcol[SUBS], dch[SUBS];
#pragma unroll
    for (int i = 0; i < SUBS; ++i) {
        const int slot = i * 64 + lane;
        dcol[i] = slot / (NCH + 1);
        dch[i] = slot < SLOTS ? slot - dcol[i] * (NCH + 1) : NCH;      // NCH = pad slot / beyond the row: never loaded
    }

    // ---- every view's footprint box of the tile, up front: a tile whose box exceeds the window in ANY view is handed
    // to the gather kernel through the worklist and this workgroup retires without touching the sources
    if (tid < 4 * MAXS && tid < 4 * d.S)
        sbox[tid >> 2][tid & 3] = ws_boxes(d.worklist, (int)gridDim.x)[(size_t)tile * (4 * MAXS) + tid];
    f2 refp[C / 2];
    {
        const float inv_cg = 1.0f / (float)(C / G);
        const float4* rp = reinterpret_cast<const float4*>(d.ref + pc * C);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const float4 q = rp[j];
            refp[2 * j] = f2{q.x * inv_cg, q.y * inv_cg};
            refp[2 * j + 1] = f2{q.z * inv_cg, q.w * inv_cg};
        }
    }

#pragma unroll
    for (int k = 0; k < N; ++k) {
        float sk = (float)k * step;
        sk += lo;
        sk = fminf(fmaxf(sk, 0.0f), 1.0f);
        if (live) d.out_samples[((long)b * d.samp_cstride + d.samp_coffset + k) * hw + yx] = sk;
    }

    for (int s = 0; s < d.S; ++s) {
        const float w = d.view_w[((long)b * d.S + s) * Hv * Wv + vwi];
        wsum += w;
        RayW ray;
        ray.init(d.rt + ((long)b * d.S + s) * 12, (float)xc, (float)yc);
        const float* view = d.src + ((long)s * d.B + b) * hw * C;
        __syncthreads();        // the boxes are published (s = 0) / every lane is done reading the previous view's window
        const int bx0 = sbox[s][0], by0 = sbox[s][1], ncols = sbox[s][2], nrows = sbox[s][3];
        if (ncols <= 0) continue;           // every tap of the tile is padding in this view (workgroup-uniform)
        // rows of the window: each a contiguous run of ncols*C floats in the NHWC source
        for (int r = wave; r < nrows; r += DMVS_BLOCK / 64) {
            const float* rowp = view + ((long)(by0 + r) * W + bx0) * C;
#pragma unroll
            for (int i = 0; i < SUBS; ++i) {
                if (dch[i] < NCH && dcol[i] < ncols) {
                    const float* srcp = rowp + dcol[i] * C + dch[i] * 4;
                    float* dstp = win + (r * SLOTS + i * 64) * 4;          // wave-uniform; lane l lands at +16*l bytes
                    __builtin_amdgcn_global_load_lds(srcp, DMVS_LDS3(dstp), 16, 0, 0);
                }
            }
        }
        __syncthreads();        // window resident (the barrier drains the LDS-DMA)

        int pfx = -0x40000000, pfy = -0x40000000;
        float D[4][G];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < G; ++g) D[t][g] = 0.0f;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            float u, v, z;
            bool fin;
            project_uv(ray, depth[k], u, v, z, fin);
            const SampW sp = make_samp(u, v, fin, H, W);
            if (sp.x0 != pfx || sp.y0 != pfy) {      // LDS reads only for a footprint that moved
                pfx = sp.x0;
                pfy = sp.y0;
                const int xa = min(max(sp.x0 - bx0, 0), ncols - 1), xb = min(max(sp.x0 + 1 - bx0, 0), ncols - 1);
                const int ya = min(max(sp.y0 - by0, 0), nrows - 1), yb = min(max(sp.y0 + 1 - by0, 0), nrows - 1);
                const int ra = __mul24(ya, WW * TS), rb = __mul24(yb, WW * TS), ca = __mul24(xa, TS), cb = __mul24(xb, TS);
                texel_dots<C>(win + ra + ca, refp, D[0]);
                texel_dots<C>(win + ra + cb, refp, D[1]);
                texel_dots<C>(win + rb + ca, refp, D[2]);
                texel_dots<C>(win + rb + cb, refp, D[3]);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float dot = D[0][g] * sp.w00 + D[1][g] * sp.w01 + D[2][g] * sp.w10 + D[3][g] * sp.w11;
                acc[k][g] = fmaf(w, dot, acc[k][g]);
            }
            if (k + 1 < N) DMVS_ORDER_AFTER(depth[k + 1], acc[k][G - 1]);   // next hypothesis starts after this one
        }
    }
    if (live) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int k = 0; k < N; ++k)
                d.out_cost[((long)b * d.cost_cstride + d.cost_coffset + g * N + k) * hw + yx] = acc[k][g] / wsum;
    }
}

template <int C>
int launch_getcost_win(const dmvs_getcost_desc& d, hipStream_t st) {
    const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
    dim3 grid((unsigned)(tiles_x * tiles_y * d.B)), block(DMVS_BLOCK);
    if (d.n == 4) {
        hipLaunchKernelGGL((getcost_fit_kernel<4>), grid, block, 0, st, d, tiles_x, tiles_y, WW, WinCfg<C>::WH);
        hipLaunchKernelGGL(getcost_compact_kernel, dim3(1), block, 0, st, d.worklist, (int)grid.x);
        hipLaunchKernelGGL((getcost_win_kernel<C, 4>), grid, block, 0, st, d, tiles_x, tiles_y);
    } else if (d.n == 6) {
        hipLaunchKernelGGL((getcost_fit_kernel<6>), grid, block, 0, st, d, tiles_x, tiles_y, WW, WinCfg<C>::WH);
        hipLaunchKernelGGL(getcost_compact_kernel, dim3(1), block, 0, st, d.worklist, (int)grid.x);
        hipLaunchKernelGGL((getcost_win_kernel<C, 6>), grid, block, 0, st, d, tiles_x, tiles_y);
    } else {
        return DMVS_EINVAL;
    }
    return dmvs_launch_status();
}

}  // namespace

// called by dmvs_getcost_f32 (warp.hip) for C in {32, 16}
int dmvs_getcost_win_dispatch(const dmvs_getcost_desc& d, hipStream_t st) {
    if (d.C == 32) return launch_getcost_win<32>(d, st);
    if (d.C == 16) return launch_getcost_win<16>(d, st);
    return DMVS_EINVAL;
}
