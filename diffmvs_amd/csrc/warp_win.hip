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
#include "warp_tile.h"

namespace {

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
    if (int rc = launch_getcost_prepass<C>(d, st)) return rc;
    const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
    dim3 grid((unsigned)(tiles_x * tiles_y * d.B)), block(DMVS_BLOCK);
    if (d.n == 4) hipLaunchKernelGGL((getcost_win_kernel<C, 4>), grid, block, 0, st, d, tiles_x, tiles_y);
    else if (d.n == 6) hipLaunchKernelGGL((getcost_win_kernel<C, 6>), grid, block, 0, st, d, tiles_x, tiles_y);
    else return DMVS_EINVAL;
    return dmvs_launch_status();
}

}  // namespace

// called by dmvs_getcost_f32 (warp.hip) for C in {32, 16}
int dmvs_getcost_win_dispatch(const dmvs_getcost_desc& d, hipStream_t st) {
    if (d.C == 32) return launch_getcost_win<32>(d, st);
    if (d.C == 16) return launch_getcost_win<16>(d, st);
    return DMVS_EINVAL;
}
