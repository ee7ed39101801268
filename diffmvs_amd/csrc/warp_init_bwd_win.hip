// Backward of the stage-1 plane sweep (dmvs_warp_corr_init_f32) w.r.t. the image features through LDS windows: the
// depth-chunked windows of warp_init_win.hip combined with the two passes of warp_bwd_win.hip.
//   workgroup = one 16x16 pixel tile x one source view; per depth chunk and channel half:
//     A. stage the source half-window (LDS-DMA), accumulate grad_ref from it;
//     B. zero the window, scatter W_t[g] * ref into it with LDS atomics, flush it row-wise to grad_src with one global
//        atomic per texel-channel.
//   W_t[g] = sum over the planes that share a 2x2 footprint of gcor[g, k] * tapweight_t.
// grad_ref is shared by the S workgroups of a tile: zeroed by the entry point, accumulated with one global atomic per
// (pixel, channel, view).  A chunk whose box exceeds the window gathers / scatters per lane in global memory.
#include "warp_tile.h"

namespace {

constexpr int IWHB = 20;

template <int C>
__global__ void __launch_bounds__(DMVS_BLOCK, 2)
warp_init_bwd_win_kernel(const float* __restrict__ ref, const float* __restrict__ src, const float* __restrict__ rt,
                         const float* __restrict__ disp_min, const float* __restrict__ disp_max, const float* __restrict__ gcor,
                         float* __restrict__ gref, float* __restrict__ gsrc, int B, int S, int D, int H, int W, int Hs, int Ws,
                         int tiles_x, int tiles_y) {
    constexpr int G = 4, CH = C / 2, NCH = CH / 4, CPG = NCH / 2, TS = CH + 4;
    constexpr int SLOTS = WW * (NCH + 1), SUBS = (SLOTS + 63) / 64;
    constexpr int MAXD = 256;
    __shared__ __attribute__((aligned(16))) float win[WW * IWHB * TS];
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

    const float dmin = disp_min[b], dmax = disp_max[b], dm1 = (float)(D - 1);
    for (int k = tid; k < D; k += DMVS_BLOCK) s_depth[k] = dmvs_disp_to_depth((float)k / dm1, dmin, dmax);

    RayW ray;
    ray.init(rt + ((long)b * S + s) * 12, (float)xc, (float)yc);
    const long voff = ((long)s * B + b) * (long)Hs * Ws * C;
    const float* view = src + voff;
    float* gview = gsrc + voff;
    const float* gp = gcor + (((long)b * S + s) * G) * D * hw + yx;        // + (g * D + k) * hw
    const float* refpix = ref + ((long)b * hw + yx) * C;
    const float inv_cg = 1.0f / (float)(C / G);

    int dcol[SUBS], dch[SUBS];
#pragma unroll
    for (int i = 0; i < SUBS; ++i) {
        const int slot = i * 64 + lane;
        dcol[i] = slot / (NCH + 1);
        dch[i] = slot < SLOTS ? slot - dcol[i] * (NCH + 1) : NCH;
    }
    __syncthreads();

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

    int fx0, fy0, fnc, fnr;
    const bool seg = plane_box(0, D - 1, fx0, fy0, fnc, fnr);
    int nchunk = 1;
    if (seg && fnc > 0) {
        const int ex = max(fnc - (WW - 5), 0), ey = max(fnr - (IWHB - 1), 0);
        nchunk = 1 + max((ex + 4) / 5, ey);
    } else if (!seg) {
        nchunk = max(D / 4, 1);
    }
    nchunk = min(nchunk, D);
    const int dc = (D + nchunk - 1) / nchunk;

    float gr[2][CH];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < CH; ++c) gr[h][c] = 0.0f;

    for (int k0 = 0; k0 < D; k0 += dc) {
        const int k1 = min(k0 + dc, D);
        int bx0, by0, ncols, nrows;
        const bool okseg = plane_box(k0, k1 - 1, bx0, by0, ncols, nrows);
        if (okseg && ncols <= 0) continue;          // every tap of every plane of the chunk is padding: no gradient
        const bool fits = okseg && ncols <= WW && nrows <= IWHB;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float* vh = view + h * CH;
            float* gvh = gview + h * CH;
            // walk the planes of the chunk once per pass, acting per distinct footprint with the accumulated tap weights
            auto walk = [&](auto&& emit) {
                int fx = 0, fy = 0;
                bool open = false;
                float Wt[4][2];
#pragma unroll
                for (int t = 0; t < 4; ++t) Wt[t][0] = Wt[t][1] = 0.0f;
#pragma unroll 1
                for (int k = k0; k <= k1; ++k) {
                    const int kk = min(k, k1 - 1);
                    float u, v, z;
                    bool fin;
                    project_uv(ray, s_depth[kk], u, v, z, fin);
                    const SampW sp = make_samp(u, v, fin, Hs, Ws);
                    const bool change = k == k1 || !open || sp.x0 != fx || sp.y0 != fy;
                    if (open && change) emit(fx, fy, Wt);
                    if (k == k1) break;
                    if (change) {
                        open = true;
                        fx = sp.x0;
                        fy = sp.y0;
#pragma unroll
                        for (int t = 0; t < 4; ++t) Wt[t][0] = Wt[t][1] = 0.0f;
                    }
                    const float g0 = live ? gp[((long)(2 * h) * D + k) * hw] : 0.0f;
                    const float g1 = live ? gp[((long)(2 * h + 1) * D + k) * hw] : 0.0f;
                    Wt[0][0] = fmaf(g0, sp.w00, Wt[0][0]); Wt[0][1] = fmaf(g1, sp.w00, Wt[0][1]);
                    Wt[1][0] = fmaf(g0, sp.w01, Wt[1][0]); Wt[1][1] = fmaf(g1, sp.w01, Wt[1][1]);
                    Wt[2][0] = fmaf(g0, sp.w10, Wt[2][0]); Wt[2][1] = fmaf(g1, sp.w10, Wt[2][1]);
                    Wt[3][0] = fmaf(g0, sp.w11, Wt[3][0]); Wt[3][1] = fmaf(g1, sp.w11, Wt[3][1]);
                }
            };
            auto tap_offsets = [&](int fx, int fy, int (&off)[4]) {
                if (fits) {
                    const int xa = min(max(fx - bx0, 0), ncols - 1), xb = min(max(fx + 1 - bx0, 0), ncols - 1);
                    const int ya = min(max(fy - by0, 0), nrows - 1), yb = min(max(fy + 1 - by0, 0), nrows - 1);
                    const int ra = __mul24(ya, WW * TS), rb = __mul24(yb, WW * TS), ca = __mul24(xa, TS), cb = __mul24(xb, TS);
                    off[0] = ra + ca; off[1] = ra + cb; off[2] = rb + ca; off[3] = rb + cb;
                } else {
                    const int xa = min(max(fx, 0), Ws - 1), xb = min(max(fx + 1, 0), Ws - 1);
                    const int ya = min(max(fy, 0), Hs - 1), yb = min(max(fy + 1, 0), Hs - 1);
                    off[0] = (ya * Ws + xa) * C; off[1] = (ya * Ws + xb) * C; off[2] = (yb * Ws + xa) * C; off[3] = (yb * Ws + xb) * C;
                }
            };

            __syncthreads();        // previous window (flush of the last pass) is done
            if (fits) {
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
            DMVS_DMA_BARRIER();        // source half-window resident

            // ---- pass A: grad_ref
            walk([&](int fx, int fy, const float (&Wt)[4][2]) {
                int off[4];
                tap_offsets(fx, fy, off);
                const float* basep = fits ? win : vh;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < NCH; ++j) {
                        const float4 q = *reinterpret_cast<const float4*>(basep + off[t] + 4 * j);
                        const float wt = Wt[t][j / CPG];
                        gr[h][4 * j] = fmaf(wt, q.x, gr[h][4 * j]);
                        gr[h][4 * j + 1] = fmaf(wt, q.y, gr[h][4 * j + 1]);
                        gr[h][4 * j + 2] = fmaf(wt, q.z, gr[h][4 * j + 2]);
                        gr[h][4 * j + 3] = fmaf(wt, q.w, gr[h][4 * j + 3]);
                    }
            });
            if (fits) {
                __syncthreads();    // every lane is done with the source window
                for (int e = tid * 4; e < nrows * (WW * TS); e += DMVS_BLOCK * 4)
                    *reinterpret_cast<float4*>(win + e) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                __syncthreads();
            }

            // ---- pass B: scatter (LDS gradient tile, or straight to grad_src when the chunk does not fit)
            const float4* rp4 = reinterpret_cast<const float4*>(refpix + h * CH);
            walk([&](int fx, int fy, const float (&Wt)[4][2]) {
                int off[4];
                tap_offsets(fx, fy, off);
                float* basep = fits ? win : gvh;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (Wt[t][0] == 0.0f && Wt[t][1] == 0.0f) continue;      // padding tap
#pragma unroll
                    for (int j = 0; j < NCH; ++j) {
                        const float4 q = rp4[j];
                        const float wt = Wt[t][j / CPG] * inv_cg;
                        atomicAdd(basep + off[t] + 4 * j, wt * q.x);
                        atomicAdd(basep + off[t] + 4 * j + 1, wt * q.y);
                        atomicAdd(basep + off[t] + 4 * j + 2, wt * q.z);
                        atomicAdd(basep + off[t] + 4 * j + 3, wt * q.w);
                    }
                }
            });
            if (fits) {
                __syncthreads();
                for (int r = wave; r < nrows; r += DMVS_BLOCK / 64) {
                    float* growp = gvh + ((long)(by0 + r) * Ws + bx0) * C;
#pragma unroll
                    for (int i = 0; i < SUBS; ++i) {
                        if (dch[i] < NCH && dcol[i] < ncols) {
                            const float4 q = *reinterpret_cast<const float4*>(win + (r * SLOTS + i * 64 + lane) * 4);
                            float* gq = growp + dcol[i] * C + dch[i] * 4;
                            if (q.x != 0.0f) atomicAdd(gq, q.x);
                            if (q.y != 0.0f) atomicAdd(gq + 1, q.y);
                            if (q.z != 0.0f) atomicAdd(gq + 2, q.z);
                            if (q.w != 0.0f) atomicAdd(gq + 3, q.w);
                        }
                    }
                }
            }
        }
    }
    if (live) {
        float* go = gref + ((long)b * hw + yx) * C;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (gr[h][c] != 0.0f) atomicAdd(go + h * CH + c, gr[h][c] * inv_cg);
    }
}

}  // namespace

// called by dmvs_warp_corr_init_bwd_f32 (warp_bwd.hip) for C = 48
int dmvs_warp_init_bwd_win_dispatch(const float* ref, const float* src, const float* rt, const float* disp_min, const float* disp_max,
                                    const float* gcor, float* gref, float* gsrc, int B, int S, int C, int D, int H, int W, int Hs,
                                    int Ws, hipStream_t st) {
    if (C != 48 || D > 256) return DMVS_EINVAL;
    hipError_t e = hipMemsetAsync(gref, 0, sizeof(float) * (size_t)B * H * W * C, st);
    if (e != hipSuccess) return (int)e;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    dim3 grid((unsigned)(tiles_x * tiles_y * B), (unsigned)S), block(DMVS_BLOCK);
    hipLaunchKernelGGL((warp_init_bwd_win_kernel<48>), grid, block, 0, st, ref, src, rt, disp_min, disp_max, gcor, gref, gsrc, B, S, D,
                       H, W, Hs, Ws, tiles_x, tiles_y);
    return dmvs_launch_status();
}
