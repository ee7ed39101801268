// Stage-1 plane sweep (differentiable_warping + group-wise correlation of InitialCost, reference models/module.py:181-218,
// :514-531) through LDS-staged source windows -- the tile-per-workgroup treatment of warp_win.hip applied to the D uniform
// inverse-depth planes of depth initialisation.
//
// The per-pixel gather kernel (warp.hip: warp_corr_init_kernel) spreads a pixel over 16 lanes (C = 48): projection data is
// broadcast with 6 ds_bpermute per hypothesis, the group reduction costs 2 more, and every lane repeats the address and
// validity arithmetic -- ~16x redundant issue slots for everything that is not the channel blend, which is why it sits at
// 8 % of the HBM roofline.  Here one workgroup = one 16x16 pixel tile x one source view, one lane = one pixel:
//   * the D planes are the same for every pixel, so their depths are a workgroup-uniform LDS table;
//   * the planes are walked in depth chunks whose epipolar extent keeps the tile's source footprint inside a
//     24 x 20 texel window; the chunk count follows from the footprint of the full range (segment property);
//   * per chunk and channel half (2 of the 4 correlation groups: 2 x 54 KB windows per CU) the window is streamed by
//     LDS-DMA, then every lane evaluates its hypotheses: projection, taps from LDS only when the 2x2 footprint moved,
//     per-texel group dots (packed fp32), 4-tap blend, one store per (group, plane) -- no cross-lane traffic at all.
// A chunk whose box still exceeds the window (degenerate geometry) gathers its taps from global memory instead.
#include "warp_tile.h"

namespace {

constexpr int IWH = 20;                  // window rows (the 16-row tile + 2 + up to 2 rows of epipolar drift per chunk)

template <int CH>
__device__ __forceinline__ void half_texel_dots(const float* tex, const f2 (&refp)[CH / 2], float (&Dg)[2]) {
    constexpr int CPG = CH / 8;          // 16-byte chunks per correlation group (CH channels = 2 groups)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
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

template <int C>
__global__ void __launch_bounds__(DMVS_BLOCK, 2)
warp_init_win_kernel(const float* __restrict__ ref, const float* __restrict__ src, const float* __restrict__ rt,
                     const float* __restrict__ disp_min, const float* __restrict__ disp_max, float* __restrict__ out, int B,
                     int S, int D, int H, int W, int Hs, int Ws, int tiles_x, int tiles_y) {
    constexpr int G = 4, CH = C / 2, NCH = CH / 4, TS = CH + 4;       // half-texel stride in floats (padded: bank spread)
    constexpr int SLOTS = WW * (NCH + 1), SUBS = (SLOTS + 63) / 64;
    constexpr int MAXD = 256;
    __shared__ __attribute__((aligned(16))) float win[WW * IWH * TS];
    __shared__ float s_depth[MAXD];
    __shared__ int red[2][DMVS_BLOCK / 64][5];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tq = (int)dmvs_xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int txi = tq % tiles_x; tq /= tiles_x;
    const int tyi = tq % tiles_y;
    const int b = tq / tiles_y;
    const int s = blockIdx.y;
    const int x = txi * TW + (tid & (TW - 1)), y = tyi * TH + (tid >> 4);
    const bool live = x < W && y < H;
    const int xc = min(x, W - 1), yc = min(y, H - 1);
    const long hw = (long)H * W, yx = (long)yc * W + xc;

    // plane depths: identical for every pixel of the batch item (reference diffusion.py:187-192, module.py:220-227)
    const float dmin = disp_min[b], dmax = disp_max[b], dm1 = (float)(D - 1);
    for (int k = tid; k < D; k += DMVS_BLOCK) s_depth[k] = dmvs_disp_to_depth((float)k / dm1, dmin, dmax);

    RayW ray;
    ray.init(rt + ((long)b * S + s) * 12, (float)xc, (float)yc);
    const float* view = src + ((long)s * B + b) * (long)Hs * Ws * C;
    float* op = out + (((long)b * S + s) * G) * D * hw + yx;

    int dcol[SUBS], dch[SUBS];
#pragma unroll
    for (int i = 0; i < SUBS; ++i) {
        const int slot = i * 64 + lane;
        dcol[i] = slot / (NCH + 1);
        dch[i] = slot < SLOTS ? slot - dcol[i] * (NCH + 1) : NCH;
    }
    __syncthreads();           // depth table published

    // source footprint box of the tile between two planes (their projections bound everything in between)
    int nred = 0;
    auto plane_box = [&](int ka, int kb, int& bx0, int& by0, int& ncols, int& nrows) -> bool {
        float u0, v0, z0, u1, v1, z1;
        bool f0, f1;
        project_uv(ray, s_depth[ka], u0, v0, z0, f0);
        project_uv(ray, s_depth[kb], u1, v1, z1, f1);
        int bad = live && (!f0 || !f1 || ((z0 < 0.0f) != (z1 < 0.0f)));
        int ax0 = 0x3fffffff, ay0 = 0x3fffffff, ax1 = -0x3fffffff, ay1 = -0x3fffffff;
        if (live && !bad) {
            const int lx = max((int)floorf(fminf(u0, u1)), 0), hx = min((int)floorf(fmaxf(u0, u1)) + 1, Ws - 1);
            const int ly = max((int)floorf(fminf(v0, v1)), 0), hy = min((int)floorf(fmaxf(v0, v1)) + 1, Hs - 1);
            if (lx <= hx && ly <= hy) {
                ax0 = lx; ax1 = hx; ay0 = ly; ay1 = hy;
            }
        }
        ax0 = wave_min(ax0); ay0 = wave_min(ay0); ax1 = wave_max(ax1); ay1 = wave_max(ay1); bad = wave_max(bad);
        int (*rd)[5] = red[nred & 1];
        ++nred;
        if (lane == 0) {
            rd[wave][0] = ax0; rd[wave][1] = ay0; rd[wave][2] = ax1; rd[wave][3] = ay1; rd[wave][4] = bad;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < DMVS_BLOCK / 64; ++q) {
            ax0 = min(ax0, rd[q][0]); ay0 = min(ay0, rd[q][1]);
            ax1 = max(ax1, rd[q][2]); ay1 = max(ay1, rd[q][3]); bad = max(bad, rd[q][4]);
        }
        bx0 = ax0; by0 = ay0; ncols = ax1 - ax0 + 1; nrows = ay1 - ay0 + 1;
        return !bad;
    };

    // depth chunks: the footprint of the whole range tells how many are needed for the chunk windows to fit
    int fx0, fy0, fnc, fnr;
    const bool seg = plane_box(0, D - 1, fx0, fy0, fnc, fnr);
    int nchunk = 1;
    if (seg && fnc > 0) {
        const int ex = max(fnc - (WW - 5), 0), ey = max(fnr - (IWH - 1), 0);       // extent beyond what one window absorbs
        nchunk = 1 + max((ex + 4) / 5, ey);                                      // ~5 texels of drift per chunk along x, 1 along y
    } else if (!seg) {
        nchunk = max(D / 4, 1);            // not a segment (a pole inside the range): short chunks, most will gather
    }
    nchunk = min(nchunk, D);
    const int dc = (D + nchunk - 1) / nchunk;

    for (int k0 = 0; k0 < D; k0 += dc) {
        const int k1 = min(k0 + dc, D);
        int bx0, by0, ncols, nrows;
        const bool okseg = plane_box(k0, k1 - 1, bx0, by0, ncols, nrows);
        const bool empty = okseg && ncols <= 0;
        const bool fits = okseg && !empty && ncols <= WW && nrows <= IWH;
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            f2 refp[CH / 2];
            {
                const float inv_cg = 1.0f / (float)(C / G);
                const float4* rp = reinterpret_cast<const float4*>(ref + ((long)b * hw + yx) * C + h * CH);
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const float4 q = rp[j];
                    refp[2 * j] = f2{q.x * inv_cg, q.y * inv_cg};
                    refp[2 * j + 1] = f2{q.z * inv_cg, q.w * inv_cg};
                }
            }
            float* o0 = op + (long)(2 * h) * D * hw;          // group 2h; group 2h+1 is D*hw further
            if (empty) {                                      // every tap of every plane of the chunk is padding
                if (live)
                    for (int k = k0; k < k1; ++k) {
                        o0[(long)k * hw] = 0.0f;
                        o0[(long)(D + k) * hw] = 0.0f;
                    }
                continue;
            }
            __syncthreads();        // every lane is done with the previous window
            if (fits) {
                const float* vh = view + h * CH;
                for (int r = wave; r < nrows; r += DMVS_BLOCK / 64) {
                    const float* rowp = vh + ((long)(by0 + r) * Ws + bx0) * C;
#pragma unroll
                    for (int i = 0; i < SUBS; ++i) {
                        if (dch[i] < NCH && dcol[i] < ncols) {
                            const float* srcp = rowp + dcol[i] * C + dch[i] * 4;
                            float* dstp = win + (r * SLOTS + i * 64) * 4;
                            __builtin_amdgcn_global_load_lds(srcp, DMVS_LDS3(dstp), 16, 0, 0);
                        }
                    }
                }
            }
            __syncthreads();        // window resident

            int pfx = -0x40000000, pfy = -0x40000000;
            float Dt[4][2];
#pragma unroll
            for (int t = 0; t < 4; ++t) Dt[t][0] = Dt[t][1] = 0.0f;
#pragma unroll 1
            for (int k = k0; k < k1; ++k) {
                float u, v, z;
                bool fin;
                project_uv(ray, s_depth[k], u, v, z, fin);
                const SampW sp = make_samp(u, v, fin, Hs, Ws);
                if (sp.x0 != pfx || sp.y0 != pfy) {
                    pfx = sp.x0;
                    pfy = sp.y0;
                    if (fits) {
                        const int xa = min(max(sp.x0 - bx0, 0), ncols - 1), xb = min(max(sp.x0 + 1 - bx0, 0), ncols - 1);
                        const int ya = min(max(sp.y0 - by0, 0), nrows - 1), yb = min(max(sp.y0 + 1 - by0, 0), nrows - 1);
                        const int ra = __mul24(ya, WW * TS), rb = __mul24(yb, WW * TS), ca = __mul24(xa, TS), cb = __mul24(xb, TS);
                        half_texel_dots<CH>(win + ra + ca, refp, Dt[0]);
                        half_texel_dots<CH>(win + ra + cb, refp, Dt[1]);
                        half_texel_dots<CH>(win + rb + ca, refp, Dt[2]);
                        half_texel_dots<CH>(win + rb + cb, refp, Dt[3]);
                    } else {
                        const int xa = min(max(sp.x0, 0), Ws - 1), xb = min(max(sp.x0 + 1, 0), Ws - 1);
                        const int ya = min(max(sp.y0, 0), Hs - 1), yb = min(max(sp.y0 + 1, 0), Hs - 1);
                        const float* vh = view + h * CH;
                        half_texel_dots<CH>(vh + ((long)ya * Ws + xa) * C, refp, Dt[0]);
                        half_texel_dots<CH>(vh + ((long)ya * Ws + xb) * C, refp, Dt[1]);
                        half_texel_dots<CH>(vh + ((long)yb * Ws + xa) * C, refp, Dt[2]);
                        half_texel_dots<CH>(vh + ((long)yb * Ws + xb) * C, refp, Dt[3]);
                    }
                }
                if (live) {
                    o0[(long)k * hw] = Dt[0][0] * sp.w00 + Dt[1][0] * sp.w01 + Dt[2][0] * sp.w10 + Dt[3][0] * sp.w11;
                    o0[(long)(D + k) * hw] = Dt[0][1] * sp.w00 + Dt[1][1] * sp.w01 + Dt[2][1] * sp.w10 + Dt[3][1] * sp.w11;
                }
            }
        }
    }
}

}  // namespace

// called by dmvs_warp_corr_init_f32 (warp.hip) for C = 48
int dmvs_warp_init_win_dispatch(const float* ref, const float* src, const float* rt, const float* disp_min, const float* disp_max,
                                float* out, int B, int S, int C, int D, int H, int W, int Hs, int Ws, hipStream_t st) {
    if (C != 48 || D > 256) return DMVS_EINVAL;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    dim3 grid((unsigned)(tiles_x * tiles_y * B), (unsigned)S), block(DMVS_BLOCK);
    hipLaunchKernelGGL((warp_init_win_kernel<48>), grid, block, 0, st, ref, src, rt, disp_min, disp_max, out, B, S, D, H, W, Hs, Ws,
                       tiles_x, tiles_y);
    return dmvs_launch_status();
}
