// Homography warp + group-wise correlation, "quad per pixel" kernels: GetCost (reference models/module.py:583-667) and the
// stage-1 plane sweep (module.py:514-531, differentiable_warping :181-218) for ANY geometry -- no tile windows, no pre-pass,
// no worklists.  The mapping, the texel walk and the GetCost kernel body are in warp_quad_core.h (shared with the bench-only
// memory-system probe, csrc/probe/getcost_probe.hip); this file holds the kernels of the path and their entry points.
#include "warp_quad_core.h"

namespace {

template <int C, int N, int TPT, int FT>
__global__ void __launch_bounds__(GC_BLOCK) getcost_quad_kernel(const dmvs_getcost_desc d) {
    getcost_quad_body<QuadProduct, C, N, TPT, FT>(d);
}

// the NB plane values of a chunk -> out[b, s, q, d0 + k, pixel]: a wave-uniform 64-bit plane base (scalar registers, advanced by one
// plane per store) + the lane's 32-bit byte offset, so that a store costs no vector address arithmetic (the per-store 64-bit
// multiply-adds and predicates this replaces were ~80 of a chunk's ~600 vector instructions); whole chunks take the unpredicated path
template <int NB>
__device__ __forceinline__ void store_planes(float* view_out, unsigned lane_off, int d0, int D, long hw, bool live, const float (&acc)[NB]) {
    if (!live) return;
    char* pl = reinterpret_cast<char*>(view_out + (long)d0 * hw);      // wave-uniform
    const long step = hw * 4;
    if (d0 + NB <= D) {
#pragma unroll
        for (int k = 0; k < NB; ++k, pl += step) *reinterpret_cast<float*>(pl + lane_off) = acc[k];
    } else {
#pragma unroll
        for (int k = 0; k < NB; ++k, pl += step)
            if (d0 + k < D) *reinterpret_cast<float*>(pl + lane_off) = acc[k];
    }
}

// ------------------------------------------------------------------------------------------ stage-1 plane sweep
// grid = (pixel blocks, S); planes in chunks of 8 (lane q projects planes d0 + q and d0 + q + 4).  out [B,S,4,D,H,W].
template <int C, int TPT, int FT>
__global__ void __launch_bounds__(DMVS_BLOCK)
warp_init_quad_kernel(const void* __restrict__ ref_f, const void* __restrict__ src, const float* __restrict__ rt,
                      const float* __restrict__ disp_min, const float* __restrict__ disp_max, float* __restrict__ out, int B, int S,
                      int D, int H, int W, int Hs, int Ws) {
    constexpr int NB = 8, HPL = 2, PPB = DMVS_BLOCK / 4;
    const int q = threadIdx.x & 3;
    const int hw = H * W;
    const int b = blockIdx.y / S, s = blockIdx.y - b * S;        // (batch item, view): workgroup-uniform
    const int pix = (int)dmvs_xcd_contiguous_block(blockIdx.x, gridDim.x) * PPB + (threadIdx.x >> 2);
    const bool live = pix < hw;
    const int yx = live ? pix : hw - 1;
    const int y = yx / W, x = yx - y * W;
    const long pq = (long)b * hw + yx;

    float ref[C / 4];
    load_ref<C, FT>(ref_f, pq, q, ref);
    RayQ ray;
    ray.init(rt + ((long)b * S + s) * 12, (float)x, (float)y);
    const char* base = reinterpret_cast<const char*>(src) + ((long)s * B + b) * (long)Hs * Ws * Feat<C, FT>::TEXEL_BYTES;   // this view (scalar)
    const unsigned view_off = Feat<C, FT>::lane_bytes(q);
    const float dmin = disp_min[b], dmax = disp_max[b];
    const float dm1 = (float)(D - 1);
    // the plane depths are the same for every pixel of the batch item: one table per workgroup instead of a division chain
    // per (pixel, plane)
    constexpr int TAB = 256;
    __shared__ float depth_tab[TAB];
    if ((int)threadIdx.x < min(D, TAB)) depth_tab[threadIdx.x] = dmvs_disp_to_depth((float)threadIdx.x / dm1, dmin, dmax);
    __syncthreads();
    float* const view_out = out + (((long)b * S + s) * 4 * D) * (long)hw;                  // [4][D][hw] of this (batch item, view): uniform
    const unsigned lane_off = (unsigned)(((long)q * D * hw + yx) * 4);                      // group plane + pixel (one view's volumes < 4 GiB: entry point)
    for (int d0 = 0; d0 < D; d0 += NB) {
        HypQ own[HPL];
#pragma unroll
        for (int h = 0; h < HPL; ++h) {
            const int dk = d0 + q + 4 * h;
            const int dc = min(dk, D - 1);
            const float depth = dc < TAB ? depth_tab[dc] : dmvs_disp_to_depth((float)dc / dm1, dmin, dmax);
            own[h] = project_q(ray, depth, dk < D, Hs, Ws);
        }
        float acc[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) acc[k] = 0.0f;
        quad_accumulate<QuadProduct, C, FT, NB, TPT>(base, view_off, Ws, own, Hs, Ws, ref, 1.0f, acc);
        store_planes<NB>(view_out, lane_off, d0, D, hw, live, acc);
    }
}

// ------------------------------------------------------------------------------------------ stage-1 plane sweep, LDS band
// The same quad arithmetic with the texels served from LDS.  The plane sweep's taps are the L1's worst case: every distinct
// texel of a (pixel, 8-plane chunk) is a separate 64-byte request per quad (~54 texels x 192 B = 10 KB of L1 traffic per
// pixel and view against 960 B of source actually needed: the launch moved ~25 GB through the 64 B/clk/CU vector L1s, which
// alone is half of its run time), while neighbouring pixels and consecutive chunks walk the SAME few source rows.  Here a
// workgroup owns a 16 x 4 pixel tile of one (batch item, view) and stages the source band its planes touch ONCE:
//   * planes are taken in groups of whole 8-plane chunks; a group's band is the bounding box of every pixel's epipolar segment
//     between the group's first and last plane (the projection of a depth interval is a straight image segment, monotonic in
//     depth while z keeps its sign), +1 texel for the 2x2 footprint, +1 all round for rounding, clipped to the image;
//   * the group size starts at all planes and halves until the band fits BAND_BYTES (typical: the whole sweep or half of it);
//   * the band is copied by LDS-DMA in 16-byte pieces (rows are contiguous in the NHWC image: fully coalesced, each texel
//     leaves L2 once per tile instead of once per pixel and chunk), byte-identical to the image, so the quads read it with
//     the same offsets (ds_read_b128: 64 contiguous bytes per quad, conflict-free for adjacent texels at 192-byte pitch);
//   * a group whose single chunk does not fit (or whose segment has a pole: z changes sign) reads global memory exactly like
//     warp_init_quad_kernel -- same code path, other pointer type.
constexpr int BTW = 16, BTH = 4;        // pixel tile of a workgroup (one quad per pixel, one 16-pixel row per wave)

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// (Round 6, measured and removed: 8 waves per workgroup sharing one staged band, the plane chunks split over two wave sets -- 4 instead of 3
// waves per SIMD behind the same LDS bytes, bit-identical -- 1257-1265 against 1070-1079 us per B=96 launch in the model's step
// (profiles/r6_plane_sweep_split_ab.json): the second wave set doubles the per-group box / staging / barrier work and halves nothing but
// the chunk loop.)
template <int C, int TPT, int FT, int BAND_BYTES>
__global__ void __launch_bounds__(DMVS_BLOCK)
warp_init_band_kernel(const void* __restrict__ ref_f, const void* __restrict__ src, const float* __restrict__ rt,
                      const float* __restrict__ disp_min, const float* __restrict__ disp_max, float* __restrict__ out, int B, int S,
                      int D, int H, int W, int Hs, int Ws, int tiles_x) {
    constexpr int NB = 8, HPL = 2, TB = Feat<C, FT>::TEXEL_BYTES, NTHREADS = DMVS_BLOCK, NWAVE = NTHREADS / 64;
    constexpr int TAB = 256;
    __shared__ __attribute__((aligned(16))) char band[BAND_BYTES];
    __shared__ float depth_tab[TAB];
    __shared__ int red[2][NWAVE][5];

    const int tid = threadIdx.x, q = tid & 3, p = tid >> 2, lane = tid & 63, wave = tid >> 6;
    const int hw = H * W;
    const int b = blockIdx.y / S, s = blockIdx.y - b * S;        // (batch item, view): workgroup-uniform
    const int tile = (int)dmvs_xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int px = txi * BTW + (p & (BTW - 1)), py = tyi * BTH + (p >> 4);
    const bool live = px < W && py < H;
    const int x = min(px, W - 1), y = min(py, H - 1);
    const int yx = y * W + x;
    const long pq = (long)b * hw + yx;

    float ref[C / 4];
    load_ref<C, FT>(ref_f, pq, q, ref);
    RayQ ray;
    ray.init(rt + ((long)b * S + s) * 12, (float)x, (float)y);
    const char* gbase = reinterpret_cast<const char*>(src) + ((long)s * B + b) * (long)Hs * Ws * TB;      // this view (scalar base)
    const unsigned view_off = Feat<C, FT>::lane_bytes(q);
    const float dmin = disp_min[b], dmax = disp_max[b];
    const float dm1 = (float)(D - 1);
    if (tid < min(D, TAB)) depth_tab[tid] = dmvs_disp_to_depth((float)tid / dm1, dmin, dmax);
    __syncthreads();
    auto plane_depth = [&](int k) { return k < TAB ? depth_tab[k] : dmvs_disp_to_depth((float)k / dm1, dmin, dmax); };

    // bounding box (clipped to the image, margins included) of the tile's footprints between planes ka and kb; false: some
    // pixel's segment is not one (non-finite end or a pole between the ends).  Workgroup-collective, one barrier.
    int par = 0;
    auto group_box = [&](int ka, int kb, int& bx0, int& by0, int& ncols, int& nrows) -> bool {
        float ua, va, ub, vb;
        const float za = ray.rz * plane_depth(ka) + ray.tz, zb = ray.rz * plane_depth(kb) + ray.tz;
        project_uv_q(ray, plane_depth(ka), ua, va);
        project_uv_q(ray, plane_depth(kb), ub, vb);
        const bool fin = fabsf(ua) < 1.0e9f && fabsf(va) < 1.0e9f && fabsf(ub) < 1.0e9f && fabsf(vb) < 1.0e9f;      // false for NaN
        int bad = (live && (!fin || ((za < 0.0f) != (zb < 0.0f)))) ? 1 : 0;
        int lx = BIG, ly = BIG, hx = -BIG, hy = -BIG;
        if (live && !bad) {
            lx = (int)floorf(fminf(ua, ub)); hx = (int)floorf(fmaxf(ua, ub));
            ly = (int)floorf(fminf(va, vb)); hy = (int)floorf(fmaxf(va, vb));
        }
        lx = wave_min_i(lx); ly = wave_min_i(ly); hx = wave_max_i(hx); hy = wave_max_i(hy); bad = wave_max_i(bad);
        int (*rd)[5] = red[par];
        par ^= 1;
        if (lane == 0) {
            rd[wave][0] = lx; rd[wave][1] = ly; rd[wave][2] = hx; rd[wave][3] = hy; rd[wave][4] = bad;
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) {
            lx = min(lx, rd[w][0]); ly = min(ly, rd[w][1]); hx = max(hx, rd[w][2]); hy = max(hy, rd[w][3]); bad = max(bad, rd[w][4]);
        }
        bx0 = max(lx - 1, 0); by0 = max(ly - 1, 0);
        const int bx1 = min(hx + 2, Ws - 1), by1 = min(hy + 2, Hs - 1);
        ncols = (hx < lx) ? 0 : max(bx1 - bx0 + 1, 0);        // hx < lx: no live pixel
        nrows = max(by1 - by0 + 1, 0);
        return bad == 0;
    };

    float* const view_out = out + (((long)b * S + s) * 4 * D) * (long)hw;                  // [4][D][hw] of this (batch item, view): uniform
    const unsigned lane_off = (unsigned)(((long)q * D * hw + yx) * 4);
    const int nchunk = (D + NB - 1) / NB;
    int cpg = nchunk;                                          // chunks per group: halved until a group's band fits
    for (int c0 = 0; c0 < nchunk;) {
        int cnt, bx0 = 0, by0 = 0, ncols = 0, nrows = 0;
        bool staged;
        for (;;) {
            cnt = min(cpg, nchunk - c0);
            const bool seg = group_box(c0 * NB, min((c0 + cnt) * NB, D) - 1, bx0, by0, ncols, nrows);
            // an empty box (every tap of the group is padding) needs no band: the texel masks come out empty
            staged = seg && ncols > 0 && nrows > 0 && ncols * nrows * TB <= BAND_BYTES;
            if (staged || cpg == 1 || (seg && (ncols == 0 || nrows == 0))) break;
            cpg = (cpg + 1) >> 1;
        }
        if (staged) {
            // every wave is past group_box's barrier, i.e. done with the previous band
            const int ppr = ncols * (TB / 16), npieces = nrows * ppr;       // 16-byte pieces per band row / in the band
            const float inv_ppr = 1.0f / (float)ppr;
            const unsigned corner = (unsigned)(__mul24(by0, Ws) + bx0) * (unsigned)TB;
            for (int i0 = wave * 64; i0 < npieces; i0 += NTHREADS) {
                const int i = i0 + lane;
                if (i < npieces) {
                    int r = (int)((float)i * inv_ppr);
                    r -= (r * ppr > i) ? 1 : 0;
                    r += ((r + 1) * ppr <= i) ? 1 : 0;
                    const int pc = i - r * ppr;
                    const char* srcp = gbase + (corner + (unsigned)__mul24(r, Ws) * (unsigned)TB + (unsigned)pc * 16u);
                    __builtin_amdgcn_global_load_lds(srcp, (__attribute__((address_space(3))) void*)(band + (size_t)i0 * 16), 16, 0, 0);
                }
            }
            DMVS_DMA_BARRIER();                                   // (waits out this wave's LDS-DMA, then the barrier)
        }
        const unsigned band_off = Feat<C, FT>::lane_bytes(q) - (unsigned)(__mul24(by0, ncols) + bx0) * (unsigned)TB;
        for (int ch = c0; ch < c0 + cnt; ++ch) {
            const int d0 = ch * NB;
            HypQ own[HPL];
#pragma unroll
            for (int h = 0; h < HPL; ++h) {
                const int dk = d0 + q + 4 * h;
                own[h] = project_q(ray, plane_depth(min(dk, D - 1)), dk < D, Hs, Ws);
            }
            float acc[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) acc[k] = 0.0f;
            if (staged) quad_accumulate<QuadProduct, C, FT, NB, TPT>((lds_cptr)band, band_off, ncols, own, Hs, Ws, ref, 1.0f, acc);
            else quad_accumulate<QuadProduct, C, FT, NB, TPT>(gbase, view_off, Ws, own, Hs, Ws, ref, 1.0f, acc);
            store_planes<NB>(view_out, lane_off, d0, D, hw, live, acc);
        }
        c0 += cnt;
    }
}

}  // namespace

template <int FT>
static int launch_getcost_quad(const dmvs_getcost_desc& d, dim3 grid, dim3 block, hipStream_t st) {
#define DMVS_GCQ(CC, NN) hipLaunchKernelGGL((getcost_quad_kernel<CC, NN, QUAD_TPT, FT>), grid, block, 0, st, d)
    if (d.C == 32 && d.n == 6) DMVS_GCQ(32, 6);
    else if (d.C == 32 && d.n == 4) DMVS_GCQ(32, 4);
    else if (d.C == 16 && d.n == 4) DMVS_GCQ(16, 4);
    else if (d.C == 16 && d.n == 6) DMVS_GCQ(16, 6);
    else if (d.C == 48 && d.n == 4) DMVS_GCQ(48, 4);
    else if (d.C == 48 && d.n == 6) DMVS_GCQ(48, 6);
    else return DMVS_EINVAL;
#undef DMVS_GCQ
    return dmvs_launch_status();
}

extern "C" int dmvs_getcost_quad_f32(const dmvs_getcost_desc* dp, void* stream) {
    if (!dp) return DMVS_EINVAL;
    const dmvs_getcost_desc& d = *dp;
    if (!getcost_desc_ok(d)) return DMVS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid = getcost_grid(d), block(GC_BLOCK);
    if (d.feat_dtype == DMVS_DTYPE_BF16) return launch_getcost_quad<DMVS_DTYPE_BF16>(d, grid, block, st);
    if (d.feat_dtype == DMVS_DTYPE_F16) return launch_getcost_quad<DMVS_DTYPE_F16>(d, grid, block, st);
    if (d.feat_dtype == DMVS_DTYPE_F32_PLAIN) return launch_getcost_quad<DMVS_DTYPE_F32_PLAIN>(d, grid, block, st);
    return launch_getcost_quad<DMVS_DTYPE_F32>(d, grid, block, st);
}

template <int FT>
static int launch_warp_init_quad(const void* ref, const void* src, const float* rt, const float* disp_min, const float* disp_max, float* out,
                                 int B, int S, int C, int D, int H, int W, int Hs, int Ws, dim3 grid, dim3 block, hipStream_t st) {
#define DMVS_WIQ(CC) hipLaunchKernelGGL((warp_init_quad_kernel<CC, QUAD_TPT, FT>), grid, block, 0, st, ref, src, rt, disp_min, disp_max, out, B, S, D, H, W, Hs, Ws)
    if (C == 48) DMVS_WIQ(48);
    else if (C == 32) DMVS_WIQ(32);
    else if (C == 16) DMVS_WIQ(16);
    else return DMVS_EINVAL;
#undef DMVS_WIQ
    return dmvs_launch_status();
}

// LDS-band form: 48 KB of band per workgroup = 3 workgroups (12 waves) per CU.  Measured on the MI355X and left as they are:
// 38 KB bands (4 workgroups per CU, more plane groups) 797 vs 809 us, 4 texels per trip (141 VGPRs) 837 vs 813 us per B=96
// launch -- the kernel is bound by the VALU work per texel, not by occupancy or LDS latency.
template <int FT>
static int launch_warp_init_band(const void* ref, const void* src, const float* rt, const float* disp_min, const float* disp_max, float* out,
                                 int B, int S, int C, int D, int H, int W, int Hs, int Ws, hipStream_t st) {
    const int tiles_x = (W + BTW - 1) / BTW, tiles_y = (H + BTH - 1) / BTH;
    dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)(B * S)), block(DMVS_BLOCK);
#define DMVS_WIB(CC) hipLaunchKernelGGL((warp_init_band_kernel<CC, QUAD_TPT, FT, 48 * 1024>), grid, block, 0, st, ref, src, rt, disp_min, disp_max, out, B, S, D, H, W, Hs, Ws, tiles_x)
    if (C == 48) DMVS_WIB(48);
    else if (C == 32) DMVS_WIB(32);
    else if (C == 16) DMVS_WIB(16);
    else return DMVS_EINVAL;
#undef DMVS_WIB
    return dmvs_launch_status();
}

extern "C" int dmvs_warp_corr_init_quad_f32(const void* ref, const void* src, const float* rt, const float* disp_min,
                                            const float* disp_max, float* out, int32_t B, int32_t S, int32_t C, int32_t G,
                                            int32_t D, int32_t H, int32_t W, int32_t Hs, int32_t Ws, int32_t feat_dtype, int32_t tune,
                                            void* stream) {
    if (G != 4 || D < 2 || !ref || !src || !rt || !out) return DMVS_EINVAL;
    if (feat_dtype < DMVS_DTYPE_F32 || feat_dtype > DMVS_DTYPE_F32_PLAIN) return DMVS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if ((long)H * W >= (1L << 24) || (long)Hs * Ws >= (1L << 24) || (long)B * S > 65535) return DMVS_EINVAL;
    if (16L * D * H * W >= (1L << 32)) return DMVS_EINVAL;      // one (batch item, view)'s [4][D][H*W] volumes are addressed with 32-bit byte offsets
    if (!(tune & DMVS_TUNE_SWEEP_GLOBAL)) {      // default: the LDS-band kernel; the flag: the round-2 kernel (every texel from global memory / L1), A/B runs
        if (feat_dtype == DMVS_DTYPE_BF16) return launch_warp_init_band<DMVS_DTYPE_BF16>(ref, src, rt, disp_min, disp_max, out, B, S, C, D, H, W, Hs, Ws, st);
        if (feat_dtype == DMVS_DTYPE_F16) return launch_warp_init_band<DMVS_DTYPE_F16>(ref, src, rt, disp_min, disp_max, out, B, S, C, D, H, W, Hs, Ws, st);
        if (feat_dtype == DMVS_DTYPE_F32_PLAIN) return launch_warp_init_band<DMVS_DTYPE_F32_PLAIN>(ref, src, rt, disp_min, disp_max, out, B, S, C, D, H, W, Hs, Ws, st);
        return launch_warp_init_band<DMVS_DTYPE_F32>(ref, src, rt, disp_min, disp_max, out, B, S, C, D, H, W, Hs, Ws, st);
    }
    dim3 grid(dmvs_ceil_div((long)H * W, DMVS_BLOCK / 4), (unsigned)(B * S)), block(DMVS_BLOCK);
    if (feat_dtype == DMVS_DTYPE_BF16) return launch_warp_init_quad<DMVS_DTYPE_BF16>(ref, src, rt, disp_min, disp_max, out, B, S, C, D, H, W, Hs, Ws, grid, block, st);
    if (feat_dtype == DMVS_DTYPE_F16) return launch_warp_init_quad<DMVS_DTYPE_F16>(ref, src, rt, disp_min, disp_max, out, B, S, C, D, H, W, Hs, Ws, grid, block, st);
    if (feat_dtype == DMVS_DTYPE_F32_PLAIN) return launch_warp_init_quad<DMVS_DTYPE_F32_PLAIN>(ref, src, rt, disp_min, disp_max, out, B, S, C, D, H, W, Hs, Ws, grid, block, st);
    return launch_warp_init_quad<DMVS_DTYPE_F32>(ref, src, rt, disp_min, disp_max, out, B, S, C, D, H, W, Hs, Ws, grid, block, st);
}
