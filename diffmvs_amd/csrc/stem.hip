// FeatureNet stem: conv0.0 (3->8, 3x3) + BN + ReLU + conv0.1 (8->8, 3x3) + BN + ReLU at full resolution in ONE kernel
// (reference models/module.py:364-367, :399).  As two conv2d launches these layers move 3.4 GB per 96-image step
// (write + re-read of the 8-channel full-resolution intermediate) at ~2.3 TB/s; fused, the intermediate lives in LDS:
// 0.38 GB in, 1.0 GB out.
//   * persistent workgroups walk 16x16 output tiles; the next tile's 3 x 20 x 20 input halo streams in by LDS-DMA
//     while this tile computes;
//   * conv0.0 on the 18x18 halo'd intermediate as an implicit GEMM, B operand gathered from the LDS halo through per-lane k
//     offsets, with TWO INTERMEDIATE ROWS per MFMA like conv0.1 below (cout = 8 leaves half of the 16 A rows empty): K = (ci, input
//     row j = 0..3, kx) = 36, A rows 0-7 = W[ky = j] for intermediate row 2q, rows 8-15 = W[ky = j-1] for row 2q+1 -- 9 MFMAs per 16
//     pixel PAIRS (162 pair slots per tile = 11 groups = 99 MFMAs) instead of 7 per 16 pixels (21 groups = 147 MFMAs; round 4).  Each
//     output still sums its 27 products in (ci, ky, kx) order, the interleaved zero-weight products add exact zeros;
//     BN + ReLU, zeroed outside the image (it is conv0.1's zero padding), written to LDS [8][18x18];
//   * conv0.1 from that LDS image, with TWO OUTPUT ROWS per MFMA: cout = 8 would leave half of the 16 A rows as zero padding, so
//     rows 0-7 carry W[ky = j] (output row y) and rows 8-15 W[ky = j-1] (output row y+1) for input row y+j, j = 0..3 -- both
//     outputs read the same input operand: 24 instead of 36 MFMAs per row pair.  The pixels are the A operand and the paired
//     weights B, so a lane ends up with 4 consecutive pixels of one (row, channel): BN + ReLU + one 16-byte NCHW store.
// conv0.0: exact-fp32 MFMA; its 27 products are summed in (ci, tap) order here and in (tap, ci) order by the generic kernel, so the two agree
// to the last bits, not bitwise.
// conv0.1 (two thirds of the matrix work), template SPLIT (round 6, the default): split-bf16 arithmetic with fp32 accuracy (conv2d_tiled.h,
// DESIGN.md 4.5) -- conv0.0's epilogue writes the intermediate as three bf16 planes [plane][18 x 18][8 channels] (it holds the values in
// registers anyway: the split is 5.5 VALU per value, once), conv0.1's K = 8 channels x 12 (input row, kx) slots = 96 is exactly three
// v_mfma_f32_16x16x32_bf16 steps, six partial products each: 18 instructions of 16 cycles per row pair instead of 24 of 32, the lane's paired
// weights pre-split in 36 registers for the life of the persistent workgroup, and the next tile's input DMA proceeds under them (an fp32 MFMA
// stops the CU's vector memory: tools/calib/overlap_probe.hip).  conv0.0 stays fp32: its operand is gathered per k-slot from the 3-channel halo,
// so a split would cost more VALU work than the matrix time it saves.  DMVS_TUNE_STEM_EXACT: the exact-fp32 conv0.1.

#include "dmvs_common.h"
#include "dmvs_bf16.h"
#include "dmvs_lds_poison.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define DMVS_LDS_S(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int TS = 16;                 // output tile
constexpr int MW = TS + 2, MP = MW * MW;          // intermediate tile 18 x 18 = 324
constexpr int IW = TS + 4, IP = IW * IW;          // input tile 20 x 20 = 400
constexpr int MPLANE = 336;                        // 324 padded to 16 mod 32 (bank spread over the 4 k-groups)
constexpr int IN_FLOATS = 3 * IP;                  // 1200
constexpr int K0 = 36, K0S = K0 / 4;               // conv0.0: (ci, input row j, kx) of a row PAIR; MFMA k-steps
// V16 (the default where the alignment allows; 1094 -> 938 us per 96 images on the MI355X): the input halo in 16-byte LDS-DMA pieces (conv2d.hip, template V16).  An LDS row is
// the 16-byte aligned 24-float cover of the 20-float halo row (it starts SLACK = 2 floats further left: tile origins are multiples
// of 16 pixels, the halo starts 2 pixels left of them): 360 pieces instead of 1200 elements per tile, 6 instead of 19 wave-level DMA
// instructions.  Needs rows of 16-byte multiples on a 16-byte aligned tensor; a piece lies wholly inside or outside the image.
constexpr int IWL16 = 24, SLACK16 = 2;
constexpr int W1S = 208;                           // conv0.1 paired weight slab [j 4][kx 3][16 rows] per input channel, padded (192 -> 16 mod 32)

__device__ __attribute__((aligned(16))) const float stem_zero16[4] = {0.0f, 0.0f, 0.0f, 0.0f};

template <bool V16, bool SPLIT>
__global__ void __launch_bounds__(DMVS_BLOCK)
featurenet_stem_kernel(const float* __restrict__ x, const float* __restrict__ w0, const float* __restrict__ scale0,
                       const float* __restrict__ shift0, const float* __restrict__ w1, const float* __restrict__ scale1,
                       const float* __restrict__ shift1, float* __restrict__ y, int N, int H, int W, int tiles_x, int tiles_y, int xgroup) {
    // ONE LDS object: with the input buffers, the intermediate and the weights as separate __shared__ arrays hipcc
    // tags the accesses with alias scopes and then waits vmcnt(0) before the first ds_read of an input buffer while
    // the LDS-DMA into the OTHER buffer (same object) is in flight -- the prefetch this kernel is built around would
    // be waited out immediately.
    constexpr int IWP = V16 ? IWL16 : IW, X0 = V16 ? SLACK16 : 0;      // LDS row pitch of the input halo, halo column 0 inside a row
    constexpr int IPL = IW * IWP, INF = 3 * IPL;                       // one channel plane / one input buffer in LDS
    constexpr int MIDF = SPLIT ? 3 * MP * 4 : 8 * MPLANE;               // intermediate: three bf16 planes [324 positions][8 channels] / eight fp32 planes
    __shared__ __attribute__((aligned(16))) float lds[2 * INF + MIDF + K0 * 16 + (SPLIT ? 0 : 8 * W1S)];
    DMVS_LDS_POISON(lds);
    float* const s_mid = lds + 2 * INF;
    float* const s_w0 = s_mid + MIDF;
    [[maybe_unused]] float* const s_w1 = s_w0 + K0 * 16;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;
    const long plane = (long)H * W;
    const int ntiles = tiles_x * tiles_y * N;

    // weights -> LDS once per workgroup, output channels zero-padded to the 16 MFMA rows
    for (int e = tid; e < K0 * 16; e += DMVS_BLOCK) {      // paired slab: k = (ci, j, kx); rows 0-7 take tap row ky = j, rows 8-15 ky = j - 1
        const int k = e >> 4, row = e & 15;
        const int ci = k / 12, j = (k - ci * 12) / 3, kx = k % 3;
        const int ky = row < 8 ? j : j - 1;
        s_w0[e] = (ky >= 0 && ky <= 2) ? w0[(ci * 9 + ky * 3 + kx) * 8 + (row & 7)] : 0.0f;
    }
    if constexpr (!SPLIT) {
    for (int e = tid; e < 8 * W1S; e += DMVS_BLOCK) {
        const int ci = e / W1S, r = e - ci * W1S, jt = r >> 4, row = r & 15;
        const int j = jt / 3, kx = jt - j * 3;
        const int ky = row < 8 ? j : j - 1;              // rows 0-7: output row y, rows 8-15: output row y + 1
        s_w1[e] = (jt < 12 && ky >= 0 && ky <= 2) ? w1[(ci * 9 + ky * 3 + kx) * 8 + (row & 7)] : 0.0f;
    }
    }
    // (SPLIT) this lane's paired conv0.1 weights as bf16 triples, for the life of the workgroup: matrix column m = (output row of the pair m >> 3,
    // cout m & 7), k-slots = the 8 input channels of slot s = 4g + kq of the 12 (input row j, kx) slots: rows 0-7 take tap row ky = j, rows 8-15 ky = j - 1
    [[maybe_unused]] bf16x8 w1h[3], w1m[3], w1l[3];
    if constexpr (SPLIT) {
        const int mrow = threadIdx.x & 15;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int sl = 4 * g + ((threadIdx.x & 63) >> 4);
            const int j = sl / 3, kx = sl - j * 3;
            const int ky = mrow < 8 ? j : j - 1;
            float wv[8];
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) wv[ci] = (ky >= 0 && ky <= 2) ? w1[(ci * 9 + ky * 3 + kx) * 8 + (mrow & 7)] : 0.0f;
            dmvs_split3_bf16x8(wv, w1h[g], w1m[g], w1l[g]);
        }
    }

    // Input halo staging (3 x 20 x 20, zero padded): 4-byte LDS-DMA, 64 consecutive words per wave.  Which (channel,
    // row, column) a lane's pieces are is tile-independent: decoded once; per tile the border test is one packed compare
    // (guard bit above each 7-bit field: ((f | 128) - lo) keeps it iff f >= lo, ((hi-1 | 128) - f) iff f <= hi-1).
    constexpr int NPIECE = V16 ? INF / 4 : IN_FLOATS;      // staging pieces per tile (16-byte / 4-byte)
    constexpr int S_IT = (NPIECE + DMVS_BLOCK - 1) / DMVS_BLOCK;
    constexpr unsigned kGuard = 0x8080u;
    int e_off[S_IT], e_rc[S_IT];             // ci * plane + r * W + c from the halo origin; r | c << 8, or -1 beyond the tile
#pragma unroll
    for (int i = 0; i < S_IT; ++i) {
        const int e = i * DMVS_BLOCK + tid;
        if constexpr (V16) {                 // piece e = LDS floats 4e .. 4e+3: column c = first float inside the LDS row
            const int ci = e / (IPL / 4), rem = e - ci * (IPL / 4);
            const int r = rem / (IWP / 4), c = (rem - r * (IWP / 4)) * 4;
            e_off[i] = ci * (int)plane + r * W + c - SLACK16;
            e_rc[i] = e < NPIECE ? (r | (c << 8)) : -1;
        } else {
            const int ci = e / IP, rem = e - ci * IP;
            const int r = rem / IW, c = rem - r * IW;
            e_off[i] = ci * (int)plane + r * W + c;
            e_rc[i] = e < IN_FLOATS ? (r | (c << 8)) : -1;
        }
    }
    auto stage = [&](int tile, float* buf) {
        int tq = (int)dmvs_xcd_grouped_block((unsigned)tile, (unsigned)ntiles, (unsigned)xgroup);      // which tiles meet in one XCD's L2: conv2d_tiled.h
        const int tx = tq % tiles_x; tq /= tiles_x;
        const int ty = tq % tiles_y;
        const int n = tq / tiles_y;
        const int gy0 = ty * TS - 2, gx0 = tx * TS - 2;
        const float* origin = x + (long)n * 3 * plane + (long)gy0 * W + gx0;       // may lie outside the tensor: only in-range pieces are read
        const int gxa = gx0 - X0;               // image column of LDS-row column 0
        const unsigned lo = (unsigned)(max(0, -gy0) | (max(0, -gxa) << 8));
        const unsigned him1 = (unsigned)((min(IW, H - gy0) - 1) | ((min(IWP, W - gxa) - 1) << 8)) | kGuard;
#pragma unroll
        for (int i = 0; i < S_IT; ++i) {
            if (e_rc[i] >= 0) {
                const unsigned rc = (unsigned)e_rc[i];
                const bool ok = (((rc | kGuard) - lo) & (him1 - rc) & kGuard) == kGuard;
                const float* srcp = ok ? origin + e_off[i] : stem_zero16;
                if constexpr (V16) {
                    float* dstp = buf + (i * DMVS_BLOCK + wave * 64) * 4;
                    __builtin_amdgcn_global_load_lds(srcp, DMVS_LDS_S(dstp), 16, 0, 0);
                } else {
                    float* dstp = buf + i * DMVS_BLOCK + wave * 64;
                    __builtin_amdgcn_global_load_lds(srcp, DMVS_LDS_S(dstp), 4, 0, 0);
                }
            }
        }
    };

    // this lane's conv0.0 operands: A = paired w0[k = 4s + kq][row m], B offsets of k = (ci, j, kx) inside the input halo
    float a0[K0S];
    int koff[K0S];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < K0S; ++s) {
        const int k = 4 * s + kq;
        a0[s] = s_w0[k * 16 + m];
        const int ci = k / 12, j = (k - ci * 12) / 3, kx = k % 3;
        koff[s] = ci * IPL + j * IWP + kx;
    }
    float sc0[4], sh0[4];       // conv0.0: this lane's D rows 4*kq + r = channel (4*kq + r) & 7 of intermediate row 2q + (kq >> 1)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = (4 * kq + r) & 7;
        sc0[r] = scale0 ? scale0[co] : 1.0f;
        sh0[r] = shift0 ? shift0[co] : 0.0f;
    }

    // BN + ReLU + NCHW store of a finished tile.  Issued one iteration LATE (after the next tile's barrier): the
    // barrier's vmcnt(0) -- needed for the LDS-DMA -- would otherwise also wait out stores issued just before it.
    // (transposed accumulators: conv0.1's MFMAs take the pixels as A and the paired weights as B, so a lane holds row
    // (m >> 3) of the pair, channel m & 7, of the 4 consecutive pixels 4*kq + r: one 16-byte store per row pair)
    const bool vec = (W & 3) == 0 && ((uintptr_t)y & 15) == 0;
    const int co1 = m & 7;
    const float sc1t = scale1 ? scale1[co1] : 1.0f, sh1t = shift1 ? shift1[co1] : 0.0f;
    auto store_tile = [&](const f32x4 (&a)[2], int n, int ox0, int oy0) {
        const int ox = ox0 + 4 * kq;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int oy = oy0 + wave * 4 + 2 * pr + (m >> 3);
            if (ox < W && oy < H) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(fmaf(a[pr][r], sc1t, sh1t), 0.0f);
                float* dst = y + ((long)n * 8 + co1) * plane + (long)oy * W + ox;
                if (vec) {
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ox + r < W) dst[r] = v[r];
                }
            }
        }
    };

    int tile = blockIdx.x, cur = 0;
    f32x4 pend[2];                       // conv0.1 accumulators of the previous tile (two row pairs), not stored yet
    int pn = -1, pox0 = 0, poy0 = 0;
    if (tile < ntiles) stage(tile, lds);
    for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
        int tq = (int)dmvs_xcd_grouped_block((unsigned)tile, (unsigned)ntiles, (unsigned)xgroup);
        const int tx = tq % tiles_x; tq /= tiles_x;
        const int ty = tq % tiles_y;
        const int n = tq / tiles_y;
        const int ox0 = tx * TS, oy0 = ty * TS;
        DMVS_DMA_BARRIER();     // this tile's halo has landed; everyone is done with s_mid and the other input buffer
        if (tile + (int)gridDim.x < ntiles) stage(tile + gridDim.x, lds + (cur ^ 1) * INF);
        if (pn >= 0) store_tile(pend, pn, pox0, poy0);
        const float* in = lds + cur * INF;

        // ---- conv0.0 -> s_mid: 11 groups of 16 pair slots (slot = (row pair q, column), row-major over 9 x 18), waves take groups
        // round-robin
        constexpr int NSLOT = (MW / 2) * MW;
        for (int gidx = wave; gidx < (NSLOT + 15) / 16; gidx += DMVS_BLOCK / 64) {
            const int p = min(gidx * 16 + m, NSLOT - 1);
            const int q = p / MW, px = p - q * MW;
            const float* ip = in + (2 * q) * IWP + X0 + px;
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s = 0; s < K0S; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], ip[koff[s]], acc, 0, 0, 0);
            // D[row = 4*kq + r][col = m]: rows 0-7 = the 8 channels of intermediate pixel (2q, px), rows 8-15 = those of (2q + 1, px)
            const int py = 2 * q + (kq >> 1);
            const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
            const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
            if (gidx * 16 + m < NSLOT) {
                if constexpr (SPLIT) {
                    // the lane's 4 channels 4 * (kq & 1) + r of position (py, px) as bf16 triples: 8 bytes per plane
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = inside ? fmaxf(fmaf(acc[r], sc0[r], sh0[r]), 0.0f) : 0.0f;
                        v[4 + r] = 0.0f;
                    }
                    bf16x8 h8, m8, l8;
                    dmvs_split3_bf16x8(v, h8, m8, l8);
                    typedef uint32_t u32x2s __attribute__((ext_vector_type(2)));
                    typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
                    u32x2s* const q = reinterpret_cast<u32x2s*>(s_mid) + (py * MW + px) * 2 + (kq & 1);
                    const u32x4s hh = __builtin_bit_cast(u32x4s, h8), mm = __builtin_bit_cast(u32x4s, m8), ll = __builtin_bit_cast(u32x4s, l8);
                    q[0] = u32x2s{hh[0], hh[1]};
                    q[MP * 2] = u32x2s{mm[0], mm[1]};
                    q[2 * MP * 2] = u32x2s{ll[0], ll[1]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        s_mid[(4 * (kq & 1) + r) * MPLANE + py * MW + px] = inside ? fmaxf(fmaf(acc[r], sc0[r], sh0[r]), 0.0f) : 0.0f;
                }
            }
        }
        DMVS_LDS_BARRIER();     // s_mid complete (ds_writes only); the next tile's input DMA stays in flight

        // ---- conv0.1 from s_mid: wave = 4 output rows = 2 row pairs, 2 k-groups x 4 input rows x 3 kx
        f32x4 acc[2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) acc[pr] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (SPLIT) {
            typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
            const u32x4s* const q_hi = reinterpret_cast<const u32x4s*>(s_mid);
            const u32x4s* const q_mid = q_hi + MP;
            const u32x4s* const q_lo = q_mid + MP;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const int sl = 4 * g + kq;
                const int j = sl / 3, kx = sl - j * 3;
                const int qb = (wave * 4 + j) * MW + m + kx;
                bf16x8 ph[2], pm[2], pl[2];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int qp = qb + 2 * pr * MW;
                    ph[pr] = __builtin_bit_cast(bf16x8, q_hi[qp]);
                    pm[pr] = __builtin_bit_cast(bf16x8, q_mid[qp]);
                    pl[pr] = __builtin_bit_cast(bf16x8, q_lo[qp]);
                }
                // the partial product outermost (the two row pairs are independent accumulators), smallest products first
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) acc[pr] = dmvs_mfma_bf16(pl[pr], w1h[g], acc[pr]);
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) acc[pr] = dmvs_mfma_bf16(ph[pr], w1l[g], acc[pr]);
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) acc[pr] = dmvs_mfma_bf16(pm[pr], w1m[g], acc[pr]);
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) acc[pr] = dmvs_mfma_bf16(pm[pr], w1h[g], acc[pr]);
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) acc[pr] = dmvs_mfma_bf16(ph[pr], w1m[g], acc[pr]);
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) acc[pr] = dmvs_mfma_bf16(ph[pr], w1h[g], acc[pr]);
            }
        } else {
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4) {
            const int ci = c4 * 4 + kq;
            const float* wp = s_w1 + ci * W1S + m;
            const float* mp = s_mid + ci * MPLANE + (wave * 4) * MW + m;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float av = wp[(j * 3 + kx) * 16];
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr)
                        acc[pr] = __builtin_amdgcn_mfma_f32_16x16x4f32(mp[(2 * pr + j) * MW + kx], av, acc[pr], 0, 0, 0);      // D[pixel][(row, cout)]
                }
        }
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) pend[pr] = acc[pr];
        pn = n; pox0 = ox0; poy0 = oy0;
    }
    if (pn >= 0) store_tile(pend, pn, pox0, poy0);
}

}  // namespace

extern "C" int dmvs_featurenet_stem_f32(const float* x, const float* w0, const float* scale0, const float* shift0,
                                        const float* w1, const float* scale1, const float* shift1, float* y, int32_t N,
                                        int32_t H, int32_t W, int32_t tune, void* stream) {
    if (!x || !w0 || !w1 || !y || N <= 0 || H <= 0 || W <= 0) return DMVS_EINVAL;
    if ((long)3 * H * W >= (1L << 31)) return DMVS_EINVAL;
    const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
    const long ntiles = (long)tiles_x * tiles_y * N;
    if (ntiles >= (1L << 31)) return DMVS_EINVAL;
    // persistent workgroups: exactly as many as are resident at once (the occupancy query: 5 per CU with the 16-byte form's 31 KB of
    // LDS, 7 with the 4-byte form's 22 KB).  Until round 4 this was a fixed 6 per CU -- with 31 KB the sixth workgroup of every CU started
    // when the other five had walked all their tiles, and then walked its own share alone on an otherwise idle CU.
    const bool v16 = !(tune & DMVS_TUNE_PIECES4) && (W & 3) == 0 && ((uintptr_t)x & 15) == 0;
    const bool split = !(tune & DMVS_TUNE_STEM_EXACT);
    // DMVS_TUNE_XCD_GROUP: 0 = groups of 4 x-adjacent tiles per XCD (a 16-pixel tile row is half a cache line), 1 = plain round robin, 2 | 3 | 4 = 2 | 4 | 8
    const int xg = (tune >> 14) & 7;
    const int xgroup = xg == 0 ? 4 : (xg <= 4 ? 1 << (xg - 1) : 4);
    // input halo in 16-byte LDS-DMA pieces wherever rows are 16-byte multiples on a 16-byte aligned tensor (6 instead of 19 wave-level
    // DMA instructions per tile): 1094 -> 938 us per 96 images on the MI355X, bit-identical (profiles/r4_optins_ab.jsonl); DMVS_TUNE_PIECES4
    // forces the 4-byte form
#define DMVS_STEM_LAUNCH(V16V, SPLITV) do { \
        static const int resident = dmvs_resident_workgroups(reinterpret_cast<const void*>(featurenet_stem_kernel<V16V, SPLITV>)); \
        const unsigned grid = (unsigned)(ntiles < resident ? ntiles : resident); \
        hipLaunchKernelGGL((featurenet_stem_kernel<V16V, SPLITV>), dim3(grid), dim3(DMVS_BLOCK), 0, (hipStream_t)stream, x, w0, scale0, shift0, w1, \
                           scale1, shift1, y, N, H, W, tiles_x, tiles_y, xgroup); } while (0)
    if (v16 && split) DMVS_STEM_LAUNCH(true, true);
    else if (v16) DMVS_STEM_LAUNCH(true, false);
    else if (split) DMVS_STEM_LAUNCH(false, true);
    else DMVS_STEM_LAUNCH(false, false);
#undef DMVS_STEM_LAUNCH
    return dmvs_launch_status();
}
