// Backward of GetCost w.r.t. the image features through LDS windows (training step; autograd through reference
// models/module.py:630-661).  The per-pixel kernel (warp_bwd.hip) issues one global fp32 atomic per (pixel, hypothesis
// footprint, tap, channel) -- ~10^9 per launch at training sizes, and that is what bounds it.  Here a 16x16 pixel tile
// (one lane per pixel) works per view in two passes over ONE LDS window (the tile's source footprint, warp_tile.h):
//   A. the source window is staged by LDS-DMA and read for  grad_ref[p,c] += sum_t W_t[g(c)] * src_t[c];
//   B. the window is zeroed and becomes the gradient tile:  gwin_t[c] += W_t[g(c)] * ref[p,c]  with LDS atomics
//      (ds_add_f32), then flushed to grad_src with ONE global atomic per window texel-channel, row-contiguous.
// W_t[g] = sum over the hypotheses that share a 2x2 footprint of (gcost[g,k] * w_view / wsum) * tapweight_t: the
// hypotheses of a pixel walk the epipolar line in sub-texel steps, so both passes run once per distinct footprint.
// Tiles whose footprint exceeds the window take the per-pixel kernel (tile list / mode flag of the pre-pass).
#include "warp_tile.h"

namespace {

template <int C, int N>
__global__ void __launch_bounds__(DMVS_BLOCK, 2)
getcost_bwd_win_kernel(const dmvs_getcost_desc d, const float* __restrict__ gcost, float* __restrict__ gref,
                       float* __restrict__ gsrc, int tiles_x, int tiles_y) {
    constexpr int G = 4, NCH = C / 4, CPG = NCH / 4, TS = C + 4;
    constexpr int WH = WinCfg<C>::WH;
    constexpr int SLOTS = WW * (NCH + 1), SUBS = (SLOTS + 63) / 64;
    __shared__ __attribute__((aligned(16))) float win[WW * WH * TS];
    __shared__ int sbox[MAXS][4];

    if (d.worklist[1]) return;           // pre-pass: the per-pixel kernel takes every tile
    const int tile = (int)dmvs_xcd_contiguous_block(blockIdx.x, gridDim.x);
    if (ws_flags(d.worklist)[tile]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = d.H, W = d.W;
    const long hw = (long)H * W;
    int b, xc, yc;
    bool live;
    float lo, step;
    tile_pixel<N>(d, tile, tiles_x, tiles_y, b, xc, yc, live, lo, step);
    const long yx = (long)yc * W + xc, pc = (long)b * hw + yx;
    const float dmin = d.disp_min[b], dmax = d.disp_max[b];
    float depth[N];
#pragma unroll
    for (int k = 0; k < N; ++k) depth[k] = hyp_depth(k, lo, step, dmin, dmax);
    if (tid < 4 * MAXS && tid < 4 * d.S)
        sbox[tid >> 2][tid & 3] = ws_boxes(d.worklist, (int)gridDim.x)[(size_t)tile * (4 * MAXS) + tid];

    const float inv_cg = 1.0f / (float)(C / G);
    float gr[C];             // grad_ref accumulators; the reference features are re-read (L2) in pass B instead of
#pragma unroll                // living in 32 more registers across both passes
    for (int c = 0; c < C; ++c) gr[c] = 0.0f;
    const float4* refp4 = reinterpret_cast<const float4*>(d.ref + pc * C);
    const int Hv = H >> d.vw_shift, Wv = W >> d.vw_shift;
    const long vwi = (long)(yc >> d.vw_shift) * Wv + (xc >> d.vw_shift);
    const float* vwp = d.view_w + (long)b * d.S * Hv * Wv + vwi;
    float wsum = 1e-8f;
    for (int s = 0; s < d.S; ++s) wsum += vwp[(long)s * Hv * Wv];
    float gk[N][G];          // d loss / d (per-view correlation), before the view weight
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int k = 0; k < N; ++k) gk[k][g] = live ? gcost[((long)b * G * N + g * N + k) * hw + yx] / wsum : 0.0f;

    int dcol[SUBS], dch[SUBS];
#pragma unroll
    for (int i = 0; i < SUBS; ++i) {
        const int slot = i * 64 + lane;
        dcol[i] = slot / (NCH + 1);
        dch[i] = slot < SLOTS ? slot - dcol[i] * (NCH + 1) : NCH;
    }

    for (int s = 0; s < d.S; ++s) {
        const float w = vwp[(long)s * Hv * Wv];
        RayW ray;
        ray.init(d.rt + ((long)b * d.S + s) * 12, (float)xc, (float)yc);
        const long voff = ((long)s * d.B + b) * hw * C;
        const float* view = d.src + voff;
        float* gview = gsrc + voff;
        __syncthreads();        // boxes published (s = 0) / the previous view's flush is done with the window
        const int bx0 = sbox[s][0], by0 = sbox[s][1], ncols = sbox[s][2], nrows = sbox[s][3];
        if (ncols <= 0) continue;           // every tap of the tile is padding in this view (workgroup-uniform)
        for (int r = wave; r < nrows; r += DMVS_BLOCK / 64) {
            const float* rowp = view + ((long)(by0 + r) * W + bx0) * C;
#pragma unroll
            for (int i = 0; i < SUBS; ++i) {
                if (dch[i] < NCH && dcol[i] < ncols) {
                    const float* srcp = rowp + dcol[i] * C + dch[i] * 4;
                    float* dstp = win + (r * SLOTS + i * 64) * 4;
                    __builtin_amdgcn_global_load_lds(srcp, DMVS_LDS3(dstp), 16, 0, 0);
                }
            }
        }
        DMVS_DMA_BARRIER();        // source window resident

        // both passes walk the hypotheses once and act per distinct footprint with the accumulated tap weights.
        // A rolled loop with ONE emit site (the emit bodies are 4*C LDS reads / atomics): hypothesis k's depth and
        // gradients are picked by select chains on the wave-uniform k; iteration N only flushes the last footprint.
        auto walk = [&](auto&& emit) {
            int fx = 0, fy = 0;
            bool open = false;
            float Wt[4][G];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int g = 0; g < G; ++g) Wt[t][g] = 0.0f;
#pragma unroll 1
            for (int k = 0; k <= N; ++k) {
                float dk = depth[0], gkk[G];
#pragma unroll
                for (int g = 0; g < G; ++g) gkk[g] = gk[0][g];
#pragma unroll
                for (int kk = 1; kk < N; ++kk) {
                    dk = k == kk ? depth[kk] : dk;
#pragma unroll
                    for (int g = 0; g < G; ++g) gkk[g] = k == kk ? gk[kk][g] : gkk[g];
                }
                float u, v, z;
                bool fin;
                project_uv(ray, dk, u, v, z, fin);
                const SampW sp = make_samp(u, v, fin, H, W);
                const bool change = k == N || !open || sp.x0 != fx || sp.y0 != fy;
                if (open && change) emit(fx, fy, Wt);
                if (k == N) break;
                if (change) {
                    open = true;
                    fx = sp.x0;
                    fy = sp.y0;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int g = 0; g < G; ++g) Wt[t][g] = 0.0f;
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float gc = gkk[g] * w;
                    Wt[0][g] = fmaf(gc, sp.w00, Wt[0][g]);
                    Wt[1][g] = fmaf(gc, sp.w01, Wt[1][g]);
                    Wt[2][g] = fmaf(gc, sp.w10, Wt[2][g]);
                    Wt[3][g] = fmaf(gc, sp.w11, Wt[3][g]);
                }
            }
        };
        auto tap_offsets = [&](int fx, int fy, int (&off)[4]) {
            const int xa = min(max(fx - bx0, 0), ncols - 1), xb = min(max(fx + 1 - bx0, 0), ncols - 1);
            const int ya = min(max(fy - by0, 0), nrows - 1), yb = min(max(fy + 1 - by0, 0), nrows - 1);
            const int ra = __mul24(ya, WW * TS), rb = __mul24(yb, WW * TS), ca = __mul24(xa, TS), cb = __mul24(xb, TS);
            off[0] = ra + ca; off[1] = ra + cb; off[2] = rb + ca; off[3] = rb + cb;
        };

        // ---- pass A: grad_ref from the source window
        walk([&](int fx, int fy, const float (&Wt)[4][G]) {
            int off[4];
            tap_offsets(fx, fy, off);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const float4 q = *reinterpret_cast<const float4*>(win + off[t] + 4 * j);
                    const float wt = Wt[t][j / CPG];
                    gr[4 * j] = fmaf(wt, q.x, gr[4 * j]);
                    gr[4 * j + 1] = fmaf(wt, q.y, gr[4 * j + 1]);
                    gr[4 * j + 2] = fmaf(wt, q.z, gr[4 * j + 2]);
                    gr[4 * j + 3] = fmaf(wt, q.w, gr[4 * j + 3]);
                }
        });
        __syncthreads();        // every lane is done with the source window
        for (int e = tid * 4; e < nrows * (WW * TS); e += DMVS_BLOCK * 4)
            *reinterpret_cast<float4*>(win + e) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        __syncthreads();

        // ---- pass B: scatter into the gradient window (LDS atomics: neighbouring pixels share texels)
        walk([&](int fx, int fy, const float (&Wt)[4][G]) {
            int off[4];
            tap_offsets(fx, fy, off);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (Wt[t][0] == 0.0f && Wt[t][1] == 0.0f && Wt[t][2] == 0.0f && Wt[t][3] == 0.0f) continue;   // padding tap
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const float4 q = refp4[j];
                    const float wt = Wt[t][j / CPG] * inv_cg;
                    atomicAdd(win + off[t] + 4 * j, wt * q.x);
                    atomicAdd(win + off[t] + 4 * j + 1, wt * q.y);
                    atomicAdd(win + off[t] + 4 * j + 2, wt * q.z);
                    atomicAdd(win + off[t] + 4 * j + 3, wt * q.w);
                }
            }
        });
        __syncthreads();

        // ---- flush: one global atomic per window texel-channel, rows contiguous in grad_src
        for (int r = wave; r < nrows; r += DMVS_BLOCK / 64) {
            float* growp = gview + ((long)(by0 + r) * W + bx0) * C;
#pragma unroll
            for (int i = 0; i < SUBS; ++i) {
                if (dch[i] < NCH && dcol[i] < ncols) {
                    const float4 q = *reinterpret_cast<const float4*>(win + (r * SLOTS + i * 64 + lane) * 4);
                    float* gp = growp + dcol[i] * C + dch[i] * 4;
                    if (q.x != 0.0f) atomicAdd(gp, q.x);
                    if (q.y != 0.0f) atomicAdd(gp + 1, q.y);
                    if (q.z != 0.0f) atomicAdd(gp + 2, q.z);
                    if (q.w != 0.0f) atomicAdd(gp + 3, q.w);
                }
            }
        }
    }
    if (live) {
        float4* gp = reinterpret_cast<float4*>(gref + pc * C);
#pragma unroll
        for (int j = 0; j < NCH; ++j)
            gp[j] = make_float4(gr[4 * j] * inv_cg, gr[4 * j + 1] * inv_cg, gr[4 * j + 2] * inv_cg, gr[4 * j + 3] * inv_cg);
    }
}

template <int C>
int launch_bwd_win(const dmvs_getcost_desc& d, const float* gcost, float* gref, float* gsrc, hipStream_t st) {
    if (int rc = launch_getcost_prepass<C>(d, st)) return rc;
    const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
    dim3 grid((unsigned)(tiles_x * tiles_y * d.B)), block(DMVS_BLOCK);
    if (d.n == 4) hipLaunchKernelGGL((getcost_bwd_win_kernel<C, 4>), grid, block, 0, st, d, gcost, gref, gsrc, tiles_x, tiles_y);
    else if (d.n == 6) hipLaunchKernelGGL((getcost_bwd_win_kernel<C, 6>), grid, block, 0, st, d, gcost, gref, gsrc, tiles_x, tiles_y);
    else return DMVS_EINVAL;
    return dmvs_launch_status();
}

}  // namespace

// called by dmvs_getcost_bwd_f32 (warp_bwd.hip) for C in {32, 16} when a worklist is supplied
int dmvs_getcost_bwd_win_dispatch(const dmvs_getcost_desc& d, const float* gcost, float* gref, float* gsrc, hipStream_t st) {
    if (d.C == 32) return launch_bwd_win<32>(d, gcost, gref, gsrc, st);
    if (d.C == 16) return launch_bwd_win<16>(d, gcost, gref, gsrc, st);
    return DMVS_EINVAL;
}
