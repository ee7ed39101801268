// fp32 3-D convolutions of PixelViewWeight / CostRegNet_small (reference
// models/module.py:422-463): 3x3x3, padding 1.
//
// Stride-1 layers (all the large volumes) run as an implicit GEMM on the matrix cores,
// the 3-D sibling of conv2d.hip:  workgroup = 16(x) x 4(y) x 4(d) output voxels x NT*16
// channels, wave w = depth slice w (4 pixel-tiles of 16 consecutive x), K loop over chunks of
// input channels staged in LDS as a [CK][6][6][18] halo tile + [CK][27][NT*16] weights,
// v_mfma_f32_16x16x4_f32 with A = weights, B = voxels.
//
// The stride-2 and transposed layers act on 1/8 .. 1/64 of the volume and use the direct
// form: one lane = one output voxel (x fastest), CO accumulators in VGPRs, tap weights
// wave-uniform.  The transposed convolution is evaluated in gather form, one output parity
// class (od&1, oh&1, ow&1) per blockIdx.z so that the live taps stay wave-uniform:
//   o = 2j   : k = 1 reads i = j            o = 2j+1 : k = 0 reads i = j+1, k = 2 reads i = j

#include <type_traits>

#include "dmvs_common.h"
#include "dmvs_lds_poison.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Which tiles meet in one XCD's L2 (round 6, dmvs_common.h dmvs_xcd_grouped_block): a tile row is 16 voxels = 64 bytes, half a cache line, and
// x-adjacent tiles share halo columns.  DMVS_TUNE3D_XCD_GROUP: 0 = groups of 4 x-adjacent tiles, 1 = plain round robin, 2 | 3 | 4 = groups of
// 2 | 4 | 8.  A bijection of the tile indices: bit-identical results.
__device__ __forceinline__ int conv3d_xcd_tile(int tile, int ntiles, int tune) {
    const int xg = (tune >> 4) & 7;
    if (xg == 1) return tile;
    const unsigned g = xg == 0 ? 4u : (xg <= 4 ? 1u << (xg - 1) : 4u);
    return (int)dmvs_xcd_grouped_block((unsigned)tile, (unsigned)ntiles, g);
}

constexpr int pad16mod32_3d(int n) {
    int m = n;
    while (m % 32 != 16) ++m;
    return m;
}

// all-zero source for LDS-DMA lanes that stage padding
__device__ __attribute__((aligned(16))) const float dmvs_zero16_3d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#define DMVS_LDS(p) ((__attribute__((address_space(3))) void*)(p))

// ---- staging and epilogue pieces shared by the three stride-1 kernels ---------------------------------------------
// Halo tiles go HBM/L2 -> LDS by LDS-DMA, 4 bytes per lane (their rows are not 16-byte multiples).  Which (z, y, x) of
// a channel plane of the halo tile a lane's elements are never changes, so it is decoded ONCE per workgroup; per tile
// only the border test remains, done for all three coordinates at once on the packed form:  with a guard bit above
// each 7-bit field, ((f | 128) - lo) keeps the guard iff f >= lo and ((hi-1 | 128) - f) keeps it iff f <= hi-1, and no
// field ever borrows from its neighbour.  (The earlier per-element `&&` chains compiled to four nested branches and
// ~90 instructions per staged element: 560 VALU + 590 SALU per 108 MFMAs, SQ PMC on the 4->8 layer.)
constexpr unsigned kHaloGuard = 0x00808080u;

template <int ID, int IH, int IW, int PLANE>
struct HaloMap {
    static constexpr int P_IT = (PLANE + DMVS_BLOCK - 1) / DMVS_BLOCK;
    int off[P_IT];       // (zz * Hin + yy) * Win + xx, relative to the tile's halo origin
    int zyx[P_IT];       // zz | yy << 8 | xx << 16, or -1: not an element of the plane (row padding / beyond it)
    __device__ __forceinline__ void init(int tid, int Hin, int Win) {
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int rem = it * DMVS_BLOCK + tid;
            const int zz = rem / (IH * IW), rem2 = rem - zz * (IH * IW);
            const int yy = rem2 / IW, xx = rem2 - yy * IW;
            off[it] = (zz * Hin + yy) * Win + xx;
            zyx[it] = rem < ID * IH * IW ? (zz | (yy << 8) | (xx << 16)) : -1;
        }
    }
    // packed bounds of the part of the halo tile at origin (gd0, gy0, gx0) that lies inside the volume
    static __device__ __forceinline__ void bounds(int gd0, int gy0, int gx0, int Din, int Hin, int Win, unsigned& lo, unsigned& him1) {
        const int zl = max(0, -gd0), yl = max(0, -gy0), xl = max(0, -gx0);
        const int zh = min(ID, Din - gd0) - 1, yh = min(IH, Hin - gy0) - 1, xh = min(IW, Win - gx0) - 1;
        lo = (unsigned)(zl | (yl << 8) | (xl << 16));
        him1 = (unsigned)(zh | (yh << 8) | (xh << 16)) | kHaloGuard;
    }
    // one channel plane: `origin` = the halo origin inside this channel (may lie outside the tensor for border tiles --
    // only in-range elements are dereferenced); a dead channel (beyond cin) is staged as zeros
    __device__ __forceinline__ void stage(const float* origin, bool chan_live, unsigned lo, unsigned him1, float* dst_plane, int wave) const {
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            if (zyx[it] >= 0) {
                const unsigned z = (unsigned)zyx[it];
                const bool in = chan_live && ((((z | kHaloGuard) - lo) & (him1 - z) & kHaloGuard) == kHaloGuard);
                const float* srcp = in ? origin + off[it] : dmvs_zero16_3d;
                float* dstp = dst_plane + it * DMVS_BLOCK + wave * 64;
                __builtin_amdgcn_global_load_lds(srcp, DMVS_LDS(dstp), 4, 0, 0);
            }
        }
    }
};

// The same halo tile in 16-BYTE pieces (conv2d.hip, template V16: an LDS-DMA instruction costs the texture path the same whatever
// its width, and 4-byte pieces need one wave-level instruction per 64 halo floats).  A row of the LDS image is the 16-byte aligned
// cover of the halo row: it starts SLACK = 3 floats left of the halo's first column (tile origins are multiples of 16 voxels, the
// padding is 1, so column -4 is 16-byte aligned when rows are multiples of 4 floats and the tensor is 16-byte aligned) and its
// pitch IWL is a whole number of pieces: 24 floats for the 18 of a 16-wide tile, 6 pieces instead of 18 elements per row.  A piece
// lies wholly inside or outside the volume.  The planes grow by a third (the pair kernel then holds 2 instead of 3 workgroups per
// CU) and it still pays: 2-6 % per layer on the MI355X (conv3d_v16_ok), so it is the default wherever the alignment allows.
template <int ID, int IH, int IW>
struct HaloMap16 {
    static constexpr int SLACK = 3;
    static constexpr int IWL = (SLACK + IW + 3) / 4 * 4;
    static constexpr int PIECES = ID * IH * IWL / 4;
    static constexpr int PLANE = pad16mod32_3d(ID * IH * IWL);
    static constexpr int P_IT = (PIECES + DMVS_BLOCK - 1) / DMVS_BLOCK;
    int off[P_IT];       // (zz * Hin + yy) * Win + xx - SLACK of the piece's first float, relative to the tile's halo origin
    int zyx[P_IT];       // zz | yy << 8 | xx << 16 (xx = 4 * piece column: its first float inside the LDS row), or -1: no such piece
    __device__ __forceinline__ void init(int tid, int Hin, int Win) {
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int pe = it * DMVS_BLOCK + tid;
            const int zz = pe / (IH * (IWL / 4)), rem2 = pe - zz * (IH * (IWL / 4));
            const int yy = rem2 / (IWL / 4), xx = (rem2 - yy * (IWL / 4)) * 4;
            off[it] = (zz * Hin + yy) * Win + xx - SLACK;
            zyx[it] = pe < PIECES ? (zz | (yy << 8) | (xx << 16)) : -1;
        }
    }
    // as HaloMap::bounds, x in LDS-row coordinates (the row starts at volume column gx0 - SLACK, a multiple of 4)
    static __device__ __forceinline__ void bounds(int gd0, int gy0, int gx0, int Din, int Hin, int Win, unsigned& lo, unsigned& him1) {
        const int gxa = gx0 - SLACK;
        const int zl = max(0, -gd0), yl = max(0, -gy0), xl = max(0, -gxa);
        const int zh = min(ID, Din - gd0) - 1, yh = min(IH, Hin - gy0) - 1, xh = min(IWL, Win - gxa) - 1;
        lo = (unsigned)(zl | (yl << 8) | (xl << 16));
        him1 = (unsigned)(zh | (yh << 8) | (xh << 16)) | kHaloGuard;
    }
    __device__ __forceinline__ void stage(const float* origin, bool chan_live, unsigned lo, unsigned him1, float* dst_plane, int wave) const {
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            if (zyx[it] >= 0) {
                const unsigned z = (unsigned)zyx[it];
                const bool in = chan_live && ((((z | kHaloGuard) - lo) & (him1 - z) & kHaloGuard) == kHaloGuard);
                const float* srcp = in ? origin + off[it] : dmvs_zero16_3d;
                float* dstp = dst_plane + (it * DMVS_BLOCK + wave * 64) * 4;
                __builtin_amdgcn_global_load_lds(srcp, DMVS_LDS(dstp), 16, 0, 0);
            }
        }
    }
};

// the two forms behind one name: Halo<V16, ID, IH, IW>::{type, PLANE, PITCH (LDS row pitch), X0 (halo column 0 inside an LDS row)}
template <bool V16, int ID, int IH, int IW>
struct HaloSel {
    static constexpr int PLANE = pad16mod32_3d(ID * IH * IW), PITCH = IW, X0 = 0;
    typedef HaloMap<ID, IH, IW, PLANE> type;
};
template <int ID, int IH, int IW>
struct HaloSel<true, ID, IH, IW> {
    typedef HaloMap16<ID, IH, IW> type;
    static constexpr int PLANE = type::PLANE, PITCH = type::IWL, X0 = type::SLACK;
};

// 16-byte halo pieces (HaloMap16) wherever rows are 16-byte multiples on a 16-byte aligned tensor: measured on the MI355X against the
// 4-byte form, bit-identical (profiles/r4_optins_ab.jsonl): PixelViewWeight conv0 at 480 volumes 3318 -> 3231 us, CostRegNet conv0
// 659 -> 646, conv1 1540 -> 1453, conv3 443 -> 415, conv5 268 -> 257.  DMVS_TUNE3D_PIECES4 forces the 4-byte form (A/B runs, tests).
static bool conv3d_v16_ok(const dmvs_conv3d_desc& d) {
    return !(d.tune & DMVS_TUNE3D_PIECES4) && (d.Win & 3) == 0 && ((uintptr_t)d.in & 15) == 0;
}

// weight slab [CK][27][NW] (+ row padding to WPAD), 16 bytes per lane; decoded once per workgroup like the halo
template <int CK, int NW, int WPAD>
struct SlabMap {
    static constexpr int W_IT = (CK * WPAD / 4 + DMVS_BLOCK - 1) / DMVS_BLOCK;
    int ci[W_IT], off[W_IT];             // channel within the chunk; offset inside its [27][cout_pad] block, -1: zero / skip
    __device__ __forceinline__ void init(int tid, int nbase, int cout_pad) {
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
            const int e4 = i * DMVS_BLOCK + tid;
            const int c = e4 / (WPAD / 4), rem4 = e4 - c * (WPAD / 4);
            const int t = rem4 / (NW / 4), n4 = rem4 - t * (NW / 4);
            const bool ok = rem4 < 27 * NW / 4 && nbase + n4 * 4 < cout_pad;
            ci[i] = e4 < CK * WPAD / 4 ? c : -1;
            off[i] = ok ? t * cout_pad + nbase + n4 * 4 : -1;
        }
    }
    __device__ __forceinline__ void stage(const float* weight, int c0, int cin, int cout_pad, float* wbuf, int wave) const {
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
            if (ci[i] >= 0) {
                const int cw = c0 + ci[i];
                const float* srcp = (off[i] >= 0 && cw < cin) ? weight + (cw * 27 * cout_pad + off[i]) : dmvs_zero16_3d;
                float* dstp = wbuf + (i * DMVS_BLOCK + wave * 64) * 4;
                __builtin_amdgcn_global_load_lds(srcp, DMVS_LDS(dstp), 16, 0, 0);
            }
        }
    }
};

// Epilogue constants of an MFMA output tile: this lane holds channels nbase + nt*16 + kq*4 + r.  Folded BN scale /
// shift sit in registers for the whole kernel (they were re-read from global memory, with a wait each, per VALUE).
template <int NT>
struct TileEpi {
    // Transposed accumulators: the kernels issue the MFMA with the operands swapped (A = input pixels, B = weights), so a lane
    // holds output channel nbase + nt*16 + m of the 4 CONSECUTIVE voxels x0 + 4*kq + r of a row: one 16-byte store (and
    // residual read) per (row, n-tile) instead of four 4-byte ones.  Values are those of the other operand order bit for bit.
    float sc[NT], sh[NT];
    int cg[NT];
    bool vec;
    __device__ __forceinline__ void init(const dmvs_conv3d_desc& d, int nbase, int m) {
        vec = (d.Wout & 3) == 0 && (((uintptr_t)d.out | (uintptr_t)d.residual) & 15) == 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            cg[nt] = nbase + nt * 16 + m;
            const bool okc = cg[nt] < d.cout;
            sc[nt] = d.scale ? d.scale[okc ? cg[nt] : 0] : 1.0f;
            sh[nt] = d.shift ? d.shift[okc ? cg[nt] : 0] : 0.0f;
        }
    }
    // rows y0 .. y0+3 of depth slice od, columns oxb .. oxb+3; outb / resb = this batch item's [cout][Dout][Hout][Wout] block,
    // addressed with 32-bit element offsets (the entry point rejects cout * volume >= 2^31)
    __device__ __forceinline__ void store(const dmvs_conv3d_desc& d, const f32x4 (&acc)[4][NT], float* outb, const float* resb, int oxb,
                                          int od, int y0, int ovol) const {
        if (oxb >= d.Wout || od >= d.Dout) return;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int oy = y0 + mt;
            if (oy >= d.Hout) continue;
            const int ovox = (od * d.Hout + oy) * d.Wout + oxb;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (cg[nt] >= d.cout) continue;
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = acc[mt][nt][r] * sc[nt] + sh[nt];
                if (d.act == DMVS_ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = fmaxf(y[r], 0.0f);
                } else if (d.act != DMVS_ACT_NONE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = dmvs_act(y[r], d.act);
                }
                const int o = cg[nt] * ovol + ovox;
                if (vec) {
                    if (resb) y += *reinterpret_cast<const f32x4*>(resb + o);
                    *reinterpret_cast<f32x4*>(outb + o) = y;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (oxb + r < d.Wout) outb[o + r] = resb ? y[r] + resb[o + r] : y[r];
                }
            }
        }
    }
};

// resident workgroups per CU of the streamed kernel (28 KB LDS, 80 VGPRs each).  3 / 4 / 5 / 6 measured 898 / 881 / 880 / 891 us
// on the 80-volume 4->8 layer: the kernel is not occupancy-limited.
constexpr int kStreamWgsPerCu = 5;

// cin <= 4 (PixelViewWeight conv0 and CostReg conv0: 4 -> 8 on the full S-view / aggregated cost volumes): the whole
// K dimension is one LDS chunk, so a workgroup of the generic kernel is load -> wait -> 108 MFMAs -> store with nothing
// to overlap.  This variant keeps workgroups resident and walks 16x4x4 voxel tiles: the next tile's halo streams into
// the other LDS buffer (LDS-DMA) while the matrix cores work on this one; the weight slab is staged once.
// One __shared__ array on purpose: with the two halo buffers and the weights as separate LDS objects hipcc attaches
// alias scopes and then waits vmcnt(0) before the first ds_read of a halo buffer while the DMA into the OTHER half is
// in flight (same object) -- which serialised the pipeline this kernel exists for.
template <int NT, bool V16 = false>
__global__ void __launch_bounds__(DMVS_BLOCK) conv3d_mfma_stream_kernel(const dmvs_conv3d_desc d, int tiles_x, int tiles_y, int tiles_d) {
    constexpr int TX = 16, TY = 4, TD = 4, CK = 4;
    constexpr int IW = TX + 2, IH = TY + 2, ID = TD + 2;
    using HS = HaloSel<V16, ID, IH, IW>;
    constexpr int PLANE = HS::PLANE, IWP = HS::PITCH;      // LDS plane size and row pitch
    constexpr int NW = NT * 16;
    constexpr int WPAD = pad16mod32_3d(27 * NW);
    using Halo = typename HS::type;
    using Slab = SlabMap<CK, NW, WPAD>;
    __shared__ __attribute__((aligned(16))) float lds[2 * CK * PLANE + CK * WPAD];
    DMVS_LDS_POISON(lds);
    float* const s_w = lds + 2 * CK * PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA bases stay in SGPRs
    const int m = lane & 15, kq = lane >> 4;
    const int nbase = blockIdx.y * NW;
    const int vol = d.Din * d.Hin * d.Win, ovol = d.Dout * d.Hout * d.Wout;
    const int ntiles = tiles_x * tiles_y * tiles_d * d.B;

    Halo halo;
    halo.init(tid, d.Hin, d.Win);
    TileEpi<NT> epi;
    epi.init(d, nbase, m);
    {
        Slab slab;
        slab.init(tid, nbase, d.cout_pad);
        slab.stage(d.weight, 0, d.cin, d.cout_pad, s_w, wave);
    }

    auto stage = [&](int b, int td, int ty, int tx, float* buf) {
        const int gx0 = tx * TX - 1, gy0 = ty * TY - 1, gd0 = td * TD - 1;
        unsigned lo, him1;
        Halo::bounds(gd0, gy0, gx0, d.Din, d.Hin, d.Win, lo, him1);
        const float* origin = d.in + (size_t)b * d.cin * vol + ((long)gd0 * d.Hin + gy0) * d.Win + gx0;
#pragma unroll
        for (int ci = 0; ci < CK; ++ci) halo.stage(origin + (long)ci * vol, ci < d.cin, lo, him1, buf + ci * PLANE, wave);
    };
    auto decode = [&](int tile, int& b, int& td, int& ty, int& tx) {
        tile = conv3d_xcd_tile(tile, ntiles, d.tune);
        tx = tile % tiles_x; tile /= tiles_x;
        ty = tile % tiles_y; tile /= tiles_y;
        td = tile % tiles_d;
        b = tile / tiles_d;
    };

    auto store = [&](const f32x4 (&a)[4][NT], int sb, int std_, int sty, int stx) {
        const size_t ob = (size_t)sb * d.cout * ovol;
        epi.store(d, a, d.out + ob, d.residual ? d.residual + ob : nullptr, stx * TX + 4 * kq, std_ * TD + wave, sty * TY, ovol);
    };

    int tile = blockIdx.x, cur = 0;
    int b = 0, td = 0, ty = 0, tx = 0;
    f32x4 pend[4][NT];                  // the previous tile's accumulators: stored one iteration late, after the barrier, so
    int pb = -1, ptd = 0, pty = 0, ptx = 0;     // that the barrier's vmcnt(0) (needed for the LDS-DMA) does not wait out fresh stores
    if (tile < ntiles) {
        decode(tile, b, td, ty, tx);
        stage(b, td, ty, tx, lds);
    }
    for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
        DMVS_DMA_BARRIER();            // this tile's halo (and, first time, the weights) landed; the other buffer is free
        int nb = 0, ntd = 0, nty = 0, ntx = 0;
        if (tile + (int)gridDim.x < ntiles) {
            decode(tile + gridDim.x, nb, ntd, nty, ntx);
            stage(nb, ntd, nty, ntx, lds + (cur ^ 1) * (CK * PLANE));
        }
        if (pb >= 0) store(pend, pb, ptd, pty, ptx);
        const float* s_in = lds + cur * (CK * PLANE);
        f32x4 acc[4][NT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        {
            const int ci = kq;
            const float* wp = s_w + ci * WPAD + m;
            const float* ipb = s_in + ci * PLANE + wave * (IH * IWP) + HS::X0 + m;
#pragma unroll 1
            for (int kd = 0; kd < 3; ++kd) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float av[NT];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) av[nt] = wp[((kd * 3 + ky) * 3 + kx) * NW + nt * 16];
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) {
                            const float bv = ipb[(kd * IH + ky + mt) * IWP + kx];
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av[nt], acc[mt][nt], 0, 0, 0);      // D[voxel][cout] (TileEpi)
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) pend[i][j] = acc[i][j];
        pb = b; ptd = td; pty = ty; ptx = tx;
        b = nb; td = ntd; ty = nty; tx = ntx;
    }
    if (pb >= 0) store(pend, pb, ptd, pty, ptx);
}

// cin <= 4 AND cout <= 8 (PixelViewWeight conv0, CostReg conv0: 4 -> 8 on the largest volumes of the model): with 8 output
// channels half of the 16 A rows of v_mfma_f32_16x16x4_f32 would be zero padding.  This variant gives the spare rows to a
// SECOND OUTPUT DEPTH SLICE: a wave owns output slices 2w and 2w+1 of a 16(x) x 4(y) x 8(d) tile, and for input slice
// j = -1 .. 2 (relative to 2w) and tap (ky, kx) the A operand is  rows 0-7 = W[kd = j+1] (output 2w), rows 8-15 = W[kd = j]
// (output 2w+1), zero where kd falls outside 0..2 -- both outputs read the same B operand (the input slice).  4 x 9 x 4 = 144
// MFMAs per wave produce two slices instead of 2 x 108: 1.5x fewer MFMAs for the same result, bit-identical per output
// (each output still sums its 27 x cin products in (kd, ky, kx) order).  Same resident, tile-pipelined structure as
// conv3d_mfma_stream_kernel.
//
// WREG (the 16-byte form; round 5: 3270 -> 3091 us on PixelViewWeight conv0 x480, 650 -> 611 us on CostRegNet conv0 x96, bit-identical,
// profiles/r5_optins.jsonl): the 36 paired weights a lane multiplies with -- (j, ky, kx) of its
// input channel kq and MFMA row m, the same for every tile of the launch -- live in 36 registers (read from global memory once per
// workgroup) instead of an LDS slab read again for every tile.  The kernel then holds only the two halo buffers in LDS: 46 KB instead
// of 56 KB in the 16-byte form = THREE workgroups per CU instead of two, and 72 instead of 108 LDS reads per 144 MFMAs.  The price is
// ~88 instead of 52 VGPRs (no occupancy cost at 3 waves per SIMD) and a fully unrolled tile body.  Same products, same order.
constexpr int kPairWgsPerCu = 3;       // 45 KB of LDS each
template <bool V16, bool WREG = false>
__global__ void __launch_bounds__(DMVS_BLOCK) conv3d_mfma_stream_pair_kernel(const dmvs_conv3d_desc d, int tiles_x, int tiles_y, int tiles_d) {
    constexpr int TX = 16, TY = 4, TD = 8, CK = 4;
    constexpr int IW = TX + 2, IH = TY + 2, ID = TD + 2;
    using HS = HaloSel<V16, ID, IH, IW>;
    constexpr int PLANE = HS::PLANE, IWP = HS::PITCH;
    constexpr int WP = pad16mod32_3d(36 * 16);             // paired weights of one input channel: [j 4][ky 3][kx 3][16 rows]
    using Halo = typename HS::type;
    __shared__ __attribute__((aligned(16))) float lds[2 * CK * PLANE + (WREG ? 0 : CK * WP)];
    DMVS_LDS_POISON(lds);
    float* const s_w = lds + 2 * CK * PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;
    const int vol = d.Din * d.Hin * d.Win, ovol = d.Dout * d.Hout * d.Wout;
    const int ntiles = tiles_x * tiles_y * tiles_d * d.B;

    Halo halo;
    halo.init(tid, d.Hin, d.Win);
    // paired weights of (input channel ci, slab position jt = (j, ky, kx), MFMA row): rows 0-7 take W[kd = j], rows 8-15 W[kd = j - 1]
    auto paired = [&](int ci, int jt, int row) -> float {
        const int j = jt / 9, t9 = jt - j * 9;
        const int kd = row < 8 ? j : j - 1, co_ = row & 7;
        return (ci < d.cin && kd >= 0 && kd <= 2 && co_ < d.cout) ? d.weight[(ci * 27 + kd * 9 + t9) * d.cout_pad + co_] : 0.0f;
    };
    float wreg[WREG ? 36 : 1];
    if constexpr (WREG) {
#pragma unroll
        for (int jt = 0; jt < 36; ++jt) wreg[jt] = paired(kq, jt, m);      // this lane's A operands: k = input channel kq, row m
    } else {
        // paired weight slab, built once per workgroup from the [cin][27][cout_pad = 8] weights
        for (int e = tid; e < CK * 36 * 16; e += DMVS_BLOCK) {
            const int ci = e / (36 * 16), rem = e - ci * (36 * 16);
            s_w[ci * WP + (rem >> 4) * 16 + (rem & 15)] = paired(ci, rem >> 4, rem & 15);
        }
    }
    // epilogue constants (transposed accumulators: A = input pixels, B = the paired weights): this lane holds output slice
    // 2w + (m >> 3), channel m & 7, of the 4 consecutive voxels tx*16 + 4*kq + r of a row -- 16-byte stores
    const int co = m & 7, sl = m >> 3;
    const float sc = d.scale ? d.scale[co < d.cout ? co : 0] : 1.0f, sh = d.shift ? d.shift[co < d.cout ? co : 0] : 0.0f;
    const bool vec = (d.Wout & 3) == 0 && (((uintptr_t)d.out | (uintptr_t)d.residual) & 15) == 0;

    auto stage = [&](int b, int td, int ty, int tx, float* buf) {
        const int gx0 = tx * TX - 1, gy0 = ty * TY - 1, gd0 = td * TD - 1;
        unsigned lo, him1;
        Halo::bounds(gd0, gy0, gx0, d.Din, d.Hin, d.Win, lo, him1);
        const float* origin = d.in + (size_t)b * d.cin * vol + ((long)gd0 * d.Hin + gy0) * d.Win + gx0;
#pragma unroll
        for (int ci = 0; ci < CK; ++ci) halo.stage(origin + (long)ci * vol, ci < d.cin, lo, him1, buf + ci * PLANE, wave);
    };
    auto decode = [&](int tile, int& b, int& td, int& ty, int& tx) {
        tile = conv3d_xcd_tile(tile, ntiles, d.tune);
        tx = tile % tiles_x; tile /= tiles_x;
        ty = tile % tiles_y; tile /= tiles_y;
        td = tile % tiles_d;
        b = tile / tiles_d;
    };
    auto store = [&](const f32x4 (&a)[4], int sb, int std_, int sty, int stx) {
        const int oxb = stx * TX + 4 * kq, od = std_ * TD + 2 * wave + sl;
        if (oxb >= d.Wout || od >= d.Dout || co >= d.cout) return;
        float* outb = d.out + (size_t)sb * d.cout * ovol;
        const float* resb = d.residual ? d.residual + (size_t)sb * d.cout * ovol : nullptr;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int oy = sty * TY + mt;
            if (oy >= d.Hout) continue;
            const int o = co * ovol + (od * d.Hout + oy) * d.Wout + oxb;
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = dmvs_act(a[mt][r] * sc + sh, d.act);
            if (vec) {
                if (resb) y += *reinterpret_cast<const f32x4*>(resb + o);
                *reinterpret_cast<f32x4*>(outb + o) = y;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (oxb + r < d.Wout) outb[o + r] = resb ? y[r] + resb[o + r] : y[r];
            }
        }
    };

    int tile = blockIdx.x, cur = 0;
    int b = 0, td = 0, ty = 0, tx = 0;
    f32x4 pend[4];                      // the previous tile's accumulators, stored one iteration late (after the barrier)
    int pb = -1, ptd = 0, pty = 0, ptx = 0;
    if (tile < ntiles) {
        decode(tile, b, td, ty, tx);
        stage(b, td, ty, tx, lds);
    }
    for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
        DMVS_DMA_BARRIER();            // this tile's halo (and, first time, the paired weights) landed; the other buffer is free
        int nb = 0, ntd = 0, nty = 0, ntx = 0;
        if (tile + (int)gridDim.x < ntiles) {
            decode(tile + gridDim.x, nb, ntd, nty, ntx);
            stage(nb, ntd, nty, ntx, lds + (cur ^ 1) * (CK * PLANE));
        }
        if (pb >= 0) store(pend, pb, ptd, pty, ptx);
        const float* s_in = lds + cur * (CK * PLANE);
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        {
            const float* wp = s_w + kq * WP + m;                                   // k = input channel kq
            const float* ipb = s_in + kq * PLANE + (2 * wave) * (IH * IWP) + HS::X0 + m;      // halo slice 2w = input slice (2w - 1)
            constexpr int kJUnroll = WREG ? 4 : 1;      // (register-resident weights are indexed by compile-time constants)
#pragma unroll kJUnroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float av = WREG ? wreg[WREG ? (j * 3 + ky) * 3 + kx : 0] : wp[((j * 3 + ky) * 3 + kx) * 16];
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) {
                            const float bv = ipb[(j * IH + ky + mt) * IWP + kx];
                            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[mt], 0, 0, 0);      // D[voxel][(slice, cout)]
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) pend[i] = acc[i];
        pb = b; ptd = td; pty = ty; ptx = tx;
        b = nb; td = ntd; ty = nty; tx = ntx;
    }
    if (pb >= 0) store(pend, pb, ptd, pty, ptx);
}

// 5..8 input channels, cout <= 8 (CostRegNet conv1, 8 -> 8 at full resolution): the paired kernel above over TWO 4-channel chunks per
// tile.  (Default since round 5: 1517 -> 983 us per 96 volumes on the MI355X, profiles/r5_optins.jsonl -- the generic kernel spends half of
// its MFMA rows on padding, 216 MFMAs per 64 voxels against 144 here; DMVS_TUNE3D_NO_PAIR selects it for A/B.)  The pipeline item is a (tile, chunk) unit: while chunk c of a tile computes,
// the next unit's 4-channel halo streams into the other LDS buffer; the accumulators live across a tile's two units and are stored one
// unit late, after the next barrier.  Weights in registers (2 x 36 per lane, see WREG above): the kernel holds only the two halo
// buffers in LDS (46 KB, three workgroups per CU).  Each output still sums its products in (ci, kd, ky, kx) order: bit-identical to the
// generic kernel.
__global__ void __launch_bounds__(DMVS_BLOCK) conv3d_mfma_stream_pair8_kernel(const dmvs_conv3d_desc d, int tiles_x, int tiles_y, int tiles_d) {
    constexpr int TX = 16, TY = 4, TD = 8, CK = 4, NCH = 2;
    constexpr int IW = TX + 2, IH = TY + 2, ID = TD + 2;
    using HS = HaloSel<true, ID, IH, IW>;
    constexpr int PLANE = HS::PLANE, IWP = HS::PITCH;
    using Halo = typename HS::type;
    __shared__ __attribute__((aligned(16))) float lds[2 * CK * PLANE];
    DMVS_LDS_POISON(lds);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;
    const int vol = d.Din * d.Hin * d.Win, ovol = d.Dout * d.Hout * d.Wout;
    const int ntiles = tiles_x * tiles_y * tiles_d * d.B;

    Halo halo;
    halo.init(tid, d.Hin, d.Win);
    // this lane's A operands: input channel 4c + kq, MFMA row m: rows 0-7 take W[kd = j] (output slice 2w), rows 8-15 W[kd = j - 1] (2w + 1)
    float wreg[NCH][36];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int jt = 0; jt < 36; ++jt) {
            const int ci = 4 * c + kq, j = jt / 9, t9 = jt - j * 9;
            const int kd = m < 8 ? j : j - 1, co_ = m & 7;
            wreg[c][jt] = (ci < d.cin && kd >= 0 && kd <= 2 && co_ < d.cout) ? d.weight[(ci * 27 + kd * 9 + t9) * d.cout_pad + co_] : 0.0f;
        }
    const int co = m & 7, sl = m >> 3;
    const float sc = d.scale ? d.scale[co < d.cout ? co : 0] : 1.0f, sh = d.shift ? d.shift[co < d.cout ? co : 0] : 0.0f;
    const bool vec = (d.Wout & 3) == 0 && (((uintptr_t)d.out | (uintptr_t)d.residual) & 15) == 0;

    auto stage = [&](int b, int td, int ty, int tx, int c, float* buf) {
        const int gx0 = tx * TX - 1, gy0 = ty * TY - 1, gd0 = td * TD - 1;
        unsigned lo, him1;
        Halo::bounds(gd0, gy0, gx0, d.Din, d.Hin, d.Win, lo, him1);
        const float* origin = d.in + ((size_t)b * d.cin + 4 * c) * vol + ((long)gd0 * d.Hin + gy0) * d.Win + gx0;
#pragma unroll
        for (int ci = 0; ci < CK; ++ci) halo.stage(origin + (long)ci * vol, 4 * c + ci < d.cin, lo, him1, buf + ci * PLANE, wave);
    };
    auto decode = [&](int tile, int& b, int& td, int& ty, int& tx) {
        tile = conv3d_xcd_tile(tile, ntiles, d.tune);
        tx = tile % tiles_x; tile /= tiles_x;
        ty = tile % tiles_y; tile /= tiles_y;
        td = tile % tiles_d;
        b = tile / tiles_d;
    };
    auto store = [&](const f32x4 (&a)[4], int sb, int std_, int sty, int stx) {
        const int oxb = stx * TX + 4 * kq, od = std_ * TD + 2 * wave + sl;
        if (oxb >= d.Wout || od >= d.Dout || co >= d.cout) return;
        float* outb = d.out + (size_t)sb * d.cout * ovol;
        const float* resb = d.residual ? d.residual + (size_t)sb * d.cout * ovol : nullptr;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int oy = sty * TY + mt;
            if (oy >= d.Hout) continue;
            const int o = co * ovol + (od * d.Hout + oy) * d.Wout + oxb;
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = dmvs_act(a[mt][r] * sc + sh, d.act);
            if (vec) {
                if (resb) y += *reinterpret_cast<const f32x4*>(resb + o);
                *reinterpret_cast<f32x4*>(outb + o) = y;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (oxb + r < d.Wout) outb[o + r] = resb ? y[r] + resb[o + r] : y[r];
            }
        }
    };
    auto chunk_mfmas = [&](const float* s_in, const float (&w)[36], f32x4 (&acc)[4]) __attribute__((always_inline)) {
        const float* ipb = s_in + kq * PLANE + (2 * wave) * (IH * IWP) + HS::X0 + m;      // halo slice 2w = input slice (2w - 1)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ipb[(j * IH + ky + mt) * IWP + kx], w[(j * 3 + ky) * 3 + kx], acc[mt], 0, 0, 0);
    };

    int tile = blockIdx.x, cur = 0;
    int b = 0, td = 0, ty = 0, tx = 0;
    f32x4 acc[4], pend[4];              // this tile's accumulators (live across its two units); the previous tile's, not stored yet
    int pb = -1, ptd = 0, pty = 0, ptx = 0;
    if (tile < ntiles) {
        decode(tile, b, td, ty, tx);
        stage(b, td, ty, tx, 0, lds);
    }
    for (; tile < ntiles; tile += gridDim.x) {
        // ---- unit (tile, chunk 0): chunk 1 of the same tile streams in meanwhile
        DMVS_DMA_BARRIER();
        stage(b, td, ty, tx, 1, lds + (cur ^ 1) * (CK * PLANE));
        if (pb >= 0) store(pend, pb, ptd, pty, ptx);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        chunk_mfmas(lds + cur * (CK * PLANE), wreg[0], acc);
        cur ^= 1;
        // ---- unit (tile, chunk 1): chunk 0 of the workgroup's next tile streams in meanwhile
        DMVS_DMA_BARRIER();
        int nb = 0, ntd = 0, nty = 0, ntx = 0;
        if (tile + (int)gridDim.x < ntiles) {
            decode(tile + gridDim.x, nb, ntd, nty, ntx);
            stage(nb, ntd, nty, ntx, 0, lds + (cur ^ 1) * (CK * PLANE));
        }
        chunk_mfmas(lds + cur * (CK * PLANE), wreg[1], acc);
        cur ^= 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) pend[i] = acc[i];
        pb = b; ptd = td; pty = ty; ptx = tx;
        b = nb; td = ntd; ty = nty; tx = ntx;
    }
    if (pb >= 0) store(pend, pb, ptd, pty, ptx);
}

// S = 2: the stride-2 layers of CostRegNet_small (conv2 8 -> 16, conv4 16 -> 32, reference module.py:428-433) as the same
// implicit GEMM: a 16 x 4 x 4 tile of OUTPUT voxels reads a 33 x 9 x 9 input halo, the B operand walks it with stride 2.
template <int NT, int S = 1, bool V16 = false>
__global__ void __launch_bounds__(DMVS_BLOCK) conv3d_mfma_kernel(const dmvs_conv3d_desc d, int tiles_x, int tiles_y, int tiles_d) {
    static_assert(!(V16 && S != 1), "16-byte halo pieces: the stride-1 form");
    constexpr int TX = 16, TY = 4, TD = 4;
    constexpr int IW = (TX - 1) * S + 3, IH = (TY - 1) * S + 3, ID = (TD - 1) * S + 3;
    using HS = HaloSel<V16, ID, IH, IW>;
    constexpr int IWP = HS::PITCH;
    // channel pitch: the 16 lanes of a k-group read 16 consecutive words (S = 1) or every second word (S = 2); the two k-groups
    // of a 32-lane half must land on disjoint banks: pitch = 16 mod 32 (S = 1), odd (17 mod 32: S = 2)
    constexpr int PLANE = S == 1 ? HS::PLANE : pad16mod32_3d(ID * IH * IW - 1) + 1;
    constexpr int NW = NT * 16;
    constexpr int WPAD = pad16mod32_3d(27 * NW);
    // input channels per LDS chunk (double buffered): 8 if that stays within 48 KB, else 4
    constexpr int kCK = (2 * 8 * (PLANE + WPAD) * 4 > 49152) ? 4 : 8;
    constexpr int BUF = kCK * (PLANE + WPAD);
    using Halo = typename std::conditional<V16, typename HS::type, HaloMap<ID, IH, IW, PLANE>>::type;
    using Slab = SlabMap<kCK, NW, WPAD>;
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
    DMVS_LDS_POISON(lds);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;
    int tile = conv3d_xcd_tile((int)blockIdx.x, (int)gridDim.x, d.tune);
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y; tile /= tiles_y;
    const int td = tile % tiles_d;
    const int b = tile / tiles_d;
    const int x0 = tx * TX, y0 = ty * TY, d0 = td * TD;
    const int nbase = blockIdx.y * NW;
    const int vol = d.Din * d.Hin * d.Win, ovol = d.Dout * d.Hout * d.Wout;

    Halo halo;
    halo.init(tid, d.Hin, d.Win);
    Slab slab;
    slab.init(tid, nbase, d.cout_pad);
    unsigned lo, him1;
    Halo::bounds(d0 * S - 1, y0 * S - 1, x0 * S - 1, d.Din, d.Hin, d.Win, lo, him1);
    const float* origin = d.in + (size_t)b * d.cin * vol + ((long)(d0 * S - 1) * d.Hin + (y0 * S - 1)) * d.Win + (x0 * S - 1);

    // LDS-DMA staging (global_load_lds): halo tile 4 bytes per lane, weight slab 16 bytes per lane
    auto stage = [&](int c0, float* buf) {
#pragma unroll
        for (int ci = 0; ci < kCK; ++ci) halo.stage(origin + (long)(c0 + ci) * vol, c0 + ci < d.cin, lo, him1, buf + ci * PLANE, wave);
        slab.stage(d.weight, c0, d.cin, d.cout_pad, buf + kCK * PLANE, wave);
    };

    f32x4 acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    stage(0, lds);
    int cur = 0;
    for (int c0 = 0; c0 < d.cin; c0 += kCK, cur ^= 1) {
        const float* s_in = lds + cur * BUF;
        const float* s_w = s_in + kCK * PLANE;
        DMVS_DMA_BARRIER();     // drains this wave's LDS-DMA; chunk c0 complete, other buffer free
        if (c0 + kCK < d.cin) stage(c0 + kCK, lds + (cur ^ 1) * BUF);
        const int live_c = d.cin - c0 < kCK ? d.cin - c0 : kCK;
        const int nc4 = (live_c + 3) >> 2;
#pragma unroll 1
        for (int c4 = 0; c4 < nc4; ++c4) {
            const int ci = c4 * 4 + kq;
            const float* wp = s_w + ci * WPAD + m;
            const float* ipb = s_in + ci * PLANE + wave * S * (IH * IWP) + HS::X0 + m * S;
#pragma unroll 1
            for (int kd = 0; kd < 3; ++kd) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float av[NT];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) av[nt] = wp[((kd * 3 + ky) * 3 + kx) * NW + nt * 16];
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) {
                            const float bv = ipb[(kd * IH + ky + mt * S) * IWP + kx];
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av[nt], acc[mt][nt], 0, 0, 0);      // D[voxel][cout] (TileEpi)
                        }
                    }
                }
            }
        }
    }

    TileEpi<NT> epi;
    epi.init(d, nbase, m);
    const size_t ob = (size_t)b * d.cout * ovol;
    epi.store(d, acc, d.out + ob, d.residual ? d.residual + ob : nullptr, x0 + 4 * kq, d0 + wave, y0, ovol);
}


// ------------------------------------------------------------------------------------------
// cout == 1 (PixelViewWeight's second conv, CostRegNet's prob head, reference :439,:456): a
// 16-wide MFMA N-tile would be 94 % padding, so this is a direct form, register-blocked so that the LDS is not the
// limit: workgroup = 16(x) x 16(y) x 16(d) voxels, one lane = 4 consecutive x of 2 rows of 2 depth slices.  Per input
// channel the lane walks the 4 x 4 (slice, row) halo rows it touches, reads each 6-float row ONCE (3 x ds_read_b64) and
// slides the 3 x-taps of every (kd, ky) that maps it to one of its 2 x 2 output rows over it: 48 LDS reads per 432 FMAs
// (the one-row-per-lane form needed 54 reads per 108).  The 27 weights of a channel are wave-uniform: scalar loads, used
// as SGPR operands of the FMAs.  One channel per chunk, double-buffered: channel c+1 streams in by LDS-DMA while c is
// consumed.  Each output's products are summed in (ci, kd, ky, kx = 2,1,0) order.
__global__ void __launch_bounds__(DMVS_BLOCK) conv3d_c1_kernel(const dmvs_conv3d_desc d, int tiles_x, int tiles_y, int tiles_d) {
    constexpr int TX = 16, TY = 16, TD = 16;
    constexpr int IW = TX + 4, IH = TY + 2, ID = TD + 2;      // row pitch 20: x halo 1 left, 3 right (8-byte aligned rows)
    constexpr int PLANE = ID * IH * IW;
    using Halo = HaloMap<ID, IH, IW, PLANE>;
    __shared__ __attribute__((aligned(16))) float lds[2 * PLANE];
    DMVS_LDS_POISON(lds);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = conv3d_xcd_tile((int)blockIdx.x, (int)gridDim.x, d.tune);
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y; tile /= tiles_y;
    const int td = tile % tiles_d;
    const int b = tile / tiles_d;
    const int x0 = tx * TX, y0 = ty * TY, d0 = td * TD;
    const int vol = d.Din * d.Hin * d.Win;
    const int lx = (tid & 3) * 4, ly = ((tid >> 2) & 7) * 2, ld = (tid >> 5) * 2;

    Halo halo;
    halo.init(tid, d.Hin, d.Win);
    unsigned lo, him1;
    Halo::bounds(d0 - 1, y0 - 1, x0 - 1, d.Din, d.Hin, d.Win, lo, him1);
    const float* origin = d.in + (size_t)b * d.cin * vol + ((long)(d0 - 1) * d.Hin + (y0 - 1)) * d.Win + (x0 - 1);

    float acc[2][2][4];
#pragma unroll
    for (int i = 0; i < 16; ++i) (&acc[0][0][0])[i] = 0.0f;
    halo.stage(origin, true, lo, him1, lds, wave);
    int cur = 0;
    for (int ci = 0; ci < d.cin; ++ci, cur ^= 1) {
        const float* s_in = lds + cur * PLANE + (ld * IH + ly) * IW + lx;
        DMVS_DMA_BARRIER();     // channel ci landed (DMA drained); the other buffer is free
        if (ci + 1 < d.cin) halo.stage(origin + (long)(ci + 1) * vol, true, lo, him1, lds + (cur ^ 1) * PLANE, wave);
        // constant address space + wave-uniform index = s_load (lgkmcnt): a vector load here would put a vmcnt(0) -- and with it
        // the wait for the NEXT channel's LDS-DMA -- in front of this channel's arithmetic
        typedef const __attribute__((address_space(4))) float* cfloat_p;
        cfloat_p wc = (cfloat_p)(uintptr_t)(d.weight + ci * 27 * d.cout_pad);
        float w[27];
#pragma unroll
        for (int t = 0; t < 27; ++t) w[t] = wc[t * d.cout_pad];
#pragma unroll
        for (int dz = 0; dz < 4; ++dz) {
#pragma unroll
            for (int yy = 0; yy < 4; ++yy) {
                const float* row = s_in + (dz * IH + yy) * IW;
                // (native vector type on purpose: its loads keep their TBAA tag, and hipcc waits out every LDS-DMA in flight --
                // here the next channel's prefetch -- before an LDS read that carries no alias metadata at all, which is what
                // the struct copy of a float2 compiles to)
                const f32x2 r01 = *reinterpret_cast<const f32x2*>(row);
                const f32x2 r23 = *reinterpret_cast<const f32x2*>(row + 2);
                const f32x2 r45 = *reinterpret_cast<const f32x2*>(row + 4);
                const float in6[6] = {r01[0], r01[1], r23[0], r23[1], r45[0], r45[1]};
#pragma unroll
                for (int od = 0; od < 2; ++od) {
#pragma unroll
                    for (int oy = 0; oy < 2; ++oy) {
                        const int kd = dz - od, ky = yy - oy;
                        if (kd < 0 || kd > 2 || ky < 0 || ky > 2) continue;
                        const float w0 = w[(kd * 3 + ky) * 3], w1 = w[(kd * 3 + ky) * 3 + 1], w2 = w[(kd * 3 + ky) * 3 + 2];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[od][oy][j] = fmaf(in6[j], w0, fmaf(in6[j + 1], w1, fmaf(in6[j + 2], w2, acc[od][oy][j])));
                    }
                }
            }
        }
    }
    const int ovol = d.Dout * d.Hout * d.Wout;
    const float sc = d.scale ? d.scale[0] : 1.0f, sh = d.shift ? d.shift[0] : 0.0f;
    float* outb = d.out + (size_t)b * ovol;
    const float* resb = d.residual ? d.residual + (size_t)b * ovol : nullptr;
    const bool vec = (d.Wout & 3) == 0;       // x0 + lx is a multiple of 4: rows of 16-byte pieces
#pragma unroll
    for (int od = 0; od < 2; ++od) {
#pragma unroll
        for (int oy = 0; oy < 2; ++oy) {
            const int gd = d0 + ld + od, gy = y0 + ly + oy;
            if (gd >= d.Dout || gy >= d.Hout || x0 + lx >= d.Wout) continue;
            const int o0 = (gd * d.Hout + gy) * d.Wout + x0 + lx;
            float y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = dmvs_act(acc[od][oy][j] * sc + sh, d.act);
            if (vec) {
                if (resb) {
                    const float4 r = *reinterpret_cast<const float4*>(resb + o0);
                    y[0] += r.x; y[1] += r.y; y[2] += r.z; y[3] += r.w;
                }
                *reinterpret_cast<float4*>(outb + o0) = make_float4(y[0], y[1], y[2], y[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (x0 + lx + j >= d.Wout) continue;
                    outb[o0 + j] = resb ? y[j] + resb[o0 + j] : y[j];
                }
            }
        }
    }
}


// cout == 1 on volumes whose rows are 16-byte multiples and at most 256 floats wide (every CostRegNet / PixelViewWeight
// volume of the reference configurations): FULL-ROW tiles.  An LDS-DMA instruction costs the texture path the same
// ~60 cycles whether its lanes move 4 or 16 bytes, and the 16-wide tiles above can only use 4-byte pieces (their halo
// starts one column left of the tile): at 26 instructions per lane and channel that kernel is bound by DMA ISSUE, 4 B/clk/CU.
// Here a tile spans the whole row, so the halo slab of a depth slice is one contiguous run of 16-byte pieces; the x padding
// is a zero piece between consecutive rows (the right pad of row r is the left pad of row r + 1).  XT = W/4 lanes cover a
// row, the 256/XT lane rows split into YT (y) x DT (d), every lane again on 2 x 2 x 4 outputs; the tile shape is chosen by the
// host per volume (c1_rows_shape).  Arithmetic and summation order are those of conv3d_c1_kernel.
constexpr int kC1RowsLds = 13448;          // floats: two channel buffers of (8 x 10 rows) x (80 + 4) + 4 -- three workgroups per CU
constexpr int kC1RowsPieces = 8;           // 16-byte pieces per lane and channel

__global__ void __launch_bounds__(DMVS_BLOCK) conv3d_c1_rows_kernel(const dmvs_conv3d_desc d, int XT, int YT, int DT, int tiles_y, int tiles_d) {
    __shared__ __attribute__((aligned(16))) float lds[kC1RowsLds];
    DMVS_LDS_POISON(lds);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = blockIdx.x;
    const int ty = tile % tiles_y; tile /= tiles_y;
    const int td = tile % tiles_d;
    const int b = tile / tiles_d;
    const int IH = 2 * YT + 2, ID = 2 * DT + 2, PW = d.Win + 4, NP = XT + 1;
    const int plane = ID * IH * PW + 4, npieces = ID * IH * NP + 1;
    const int y0 = ty * 2 * YT, d0 = td * 2 * DT;
    const int vol = d.Din * d.Hin * d.Win;

    // piece p of a channel buffer: p = 0 the leading zero piece; then per halo row XT data pieces and one zero piece
    int off[kC1RowsPieces];               // float offset from the channel's first voxel; -1: zero piece; -2: beyond the buffer
#pragma unroll
    for (int it = 0; it < kC1RowsPieces; ++it) {
        const int p = it * DMVS_BLOCK + tid;
        const int q = p - 1, row = q / NP, xi = q - row * NP;
        const int zz = row / IH, rr = row - zz * IH;
        const int gd = d0 - 1 + zz, gy = y0 - 1 + rr;
        const bool data = p >= 1 && xi < XT && gd >= 0 && gd < d.Din && gy >= 0 && gy < d.Hin;
        off[it] = p >= npieces ? -2 : data ? (gd * d.Hin + gy) * d.Win + 4 * xi : -1;
    }
    const float* chan0 = d.in + (size_t)b * d.cin * vol;
    auto stage = [&](int ci, float* buf) {
        const float* base = chan0 + (long)ci * vol;
#pragma unroll
        for (int it = 0; it < kC1RowsPieces; ++it) {
            if (off[it] != -2) {
                const float* srcp = off[it] >= 0 ? base + off[it] : dmvs_zero16_3d;
                float* dstp = buf + (it * DMVS_BLOCK + wave * 64) * 4;
                __builtin_amdgcn_global_load_lds(srcp, DMVS_LDS(dstp), 16, 0, 0);
            }
        }
    };

    const int xr = tid / XT, lx = (tid - xr * XT) * 4;
    const int rd = xr / YT, ry = xr - rd * YT;
    const bool active = rd < DT;
    const int lbase = 4 + ((2 * rd) * IH + 2 * ry) * PW + lx;      // this lane's (slice 0, row 0, column lx) of the halo slab
    float acc[2][2][4];
#pragma unroll
    for (int i = 0; i < 16; ++i) (&acc[0][0][0])[i] = 0.0f;
    stage(0, lds);
    int cur = 0;
    for (int ci = 0; ci < d.cin; ++ci, cur ^= 1) {
        const float* s_in = lds + cur * plane + lbase;
        DMVS_DMA_BARRIER();     // channel ci landed (DMA drained); the other buffer is free
        if (ci + 1 < d.cin) stage(ci + 1, lds + (cur ^ 1) * plane);
        if (!active) continue;
        typedef const __attribute__((address_space(4))) float* cfloat_p;      // wave-uniform weights: scalar loads (see conv3d_c1_kernel)
        cfloat_p wc = (cfloat_p)(uintptr_t)(d.weight + ci * 27 * d.cout_pad);
        float w[27];
#pragma unroll
        for (int t = 0; t < 27; ++t) w[t] = wc[t * d.cout_pad];
#pragma unroll
        for (int dz = 0; dz < 4; ++dz) {
#pragma unroll
            for (int yy = 0; yy < 4; ++yy) {
                const float* row = s_in + (dz * IH + yy) * PW;
                const f32x4 mid = *reinterpret_cast<const f32x4*>(row);      // native vector type: keeps its TBAA tag (see conv3d_c1_kernel)
                const float in6[6] = {row[-1], mid[0], mid[1], mid[2], mid[3], row[4]};
#pragma unroll
                for (int od = 0; od < 2; ++od) {
#pragma unroll
                    for (int oy = 0; oy < 2; ++oy) {
                        const int kd = dz - od, ky = yy - oy;
                        if (kd < 0 || kd > 2 || ky < 0 || ky > 2) continue;
                        const float w0 = w[(kd * 3 + ky) * 3], w1 = w[(kd * 3 + ky) * 3 + 1], w2 = w[(kd * 3 + ky) * 3 + 2];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[od][oy][j] = fmaf(in6[j], w0, fmaf(in6[j + 1], w1, fmaf(in6[j + 2], w2, acc[od][oy][j])));
                    }
                }
            }
        }
    }
    if (!active) return;
    const int ovol = d.Dout * d.Hout * d.Wout;
    const float sc = d.scale ? d.scale[0] : 1.0f, sh = d.shift ? d.shift[0] : 0.0f;
    float* outb = d.out + (size_t)b * ovol;
    const float* resb = d.residual ? d.residual + (size_t)b * ovol : nullptr;
#pragma unroll
    for (int od = 0; od < 2; ++od) {
#pragma unroll
        for (int oy = 0; oy < 2; ++oy) {
            const int gd = d0 + 2 * rd + od, gy = y0 + 2 * ry + oy;
            if (gd >= d.Dout || gy >= d.Hout) continue;
            const int o0 = (gd * d.Hout + gy) * d.Wout + lx;
            f32x4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = dmvs_act(acc[od][oy][j] * sc + sh, d.act);
            if (resb) y += *reinterpret_cast<const f32x4*>(resb + o0);
            *reinterpret_cast<f32x4*>(outb + o0) = y;
        }
    }
}

// tile shape of conv3d_c1_rows_kernel for a volume: fewest tiles first, then fewest staged halo rows; false: not applicable
static bool c1_rows_shape(const dmvs_conv3d_desc& d, int& XT, int& YT, int& DT) {
    if (d.Win % 4 || d.Win < 16 || d.Win > 256) return false;
    if (((uintptr_t)d.in | (uintptr_t)d.out | (uintptr_t)d.residual) & 15) return false;       // 16-byte pieces, loads and stores
    XT = d.Win / 4;
    const int RT = DMVS_BLOCK / XT;
    long best_tiles = -1, best_rows = 0;
    for (int yt = 1; yt <= 4 && yt <= RT; ++yt)
        for (int dt = 1; dt <= 4 && yt * dt <= RT; ++dt) {
            const long rows = (long)(2 * yt + 2) * (2 * dt + 2);
            if (2 * (rows * (d.Win + 4) + 4) > kC1RowsLds || rows * (XT + 1) + 1 > kC1RowsPieces * DMVS_BLOCK) continue;
            const long tiles = (long)((d.Hout + 2 * yt - 1) / (2 * yt)) * ((d.Dout + 2 * dt - 1) / (2 * dt));
            if (best_tiles < 0 || tiles < best_tiles || (tiles == best_tiles && tiles * rows < best_rows)) {
                best_tiles = tiles; best_rows = tiles * rows; YT = yt; DT = dt;
            }
        }
    return best_tiles > 0;
}


template <int CO>
__device__ __forceinline__ void conv3d_epilogue(const dmvs_conv3d_desc& d, const float (&acc)[CO], int co0, int b,
                                                size_t ovox) {
    const size_t ovol = (size_t)d.Dout * d.Hout * d.Wout;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const int cg = co0 + co;
        if (cg >= d.cout) break;
        float y = acc[co];
        if (d.scale) y *= d.scale[cg];
        if (d.shift) y += d.shift[cg];
        y = dmvs_act(y, d.act);
        const size_t oi = ((size_t)b * d.cout + cg) * ovol + ovox;
        if (d.residual) y += d.residual[oi];
        d.out[oi] = y;
    }
}

template <int CO, int STRIDE>
__global__ void __launch_bounds__(DMVS_BLOCK) conv3d_kernel(const dmvs_conv3d_desc d) {
    const long total = (long)d.B * d.Dout * d.Hout * d.Wout;
    const long p = (long)blockIdx.x * DMVS_BLOCK + threadIdx.x;
    const int co0 = blockIdx.y * CO;
    const bool live = p < total;
    long q = live ? p : total - 1;
    const int ox = (int)(q % d.Wout); q /= d.Wout;
    const int oy = (int)(q % d.Hout); q /= d.Hout;
    const int od = (int)(q % d.Dout);
    const int b = (int)(q / d.Dout);

    float acc[CO];
#pragma unroll
    for (int i = 0; i < CO; ++i) acc[i] = 0.0f;
    const size_t ivol = (size_t)d.Din * d.Hin * d.Win;
    const int id0 = od * STRIDE - 1, iy0 = oy * STRIDE - 1, ix0 = ox * STRIDE - 1;
    for (int ci = 0; ci < d.cin; ++ci) {
        const float* vol = d.in + ((size_t)b * d.cin + ci) * ivol;
        const float* wrow = d.weight + (size_t)ci * 27 * d.cout_pad + co0;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const int id = id0 + kd;
            const bool din = id >= 0 && id < d.Din;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = iy0 + ky;
                const bool yin = din && iy >= 0 && iy < d.Hin;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = ix0 + kx;
                    float v = 0.0f;
                    if (yin && ix >= 0 && ix < d.Win) v = vol[((size_t)id * d.Hin + iy) * d.Win + ix];
                    const float* wt = wrow + ((kd * 3 + ky) * 3 + kx) * d.cout_pad;
#pragma unroll
                    for (int co = 0; co < CO; ++co) acc[co] = fmaf(v, wt[co], acc[co]);
                }
            }
        }
    }
    if (!live) return;
    conv3d_epilogue<CO>(d, acc, co0, b, ((size_t)od * d.Hout + oy) * d.Wout + ox);
}

template <int CO>
__global__ void __launch_bounds__(DMVS_BLOCK) deconv3d_kernel(const dmvs_conv3d_desc d) {
    // thread = one input-grid position j; this block's parity class gives output o = 2j + par
    const long total = (long)d.B * d.Din * d.Hin * d.Win;
    const long p = (long)blockIdx.x * DMVS_BLOCK + threadIdx.x;
    const int co0 = blockIdx.y * CO;
    const int pd = (blockIdx.z >> 2) & 1, py = (blockIdx.z >> 1) & 1, px = blockIdx.z & 1;
    const bool live = p < total;
    long q = live ? p : total - 1;
    const int jx = (int)(q % d.Win); q /= d.Win;
    const int jy = (int)(q % d.Hin); q /= d.Hin;
    const int jd = (int)(q % d.Din);
    const int b = (int)(q / d.Din);

    float acc[CO];
#pragma unroll
    for (int i = 0; i < CO; ++i) acc[i] = 0.0f;
    const size_t ivol = (size_t)d.Din * d.Hin * d.Win;
    // per dimension: parity 0 -> taps {(k=1, +0)}; parity 1 -> taps {(k=0, +1), (k=2, +0)}
    const int nd = pd ? 2 : 1, ny = py ? 2 : 1, nx = px ? 2 : 1;
    for (int ci = 0; ci < d.cin; ++ci) {
        const float* vol = d.in + ((size_t)b * d.cin + ci) * ivol;
        const float* wrow = d.weight + (size_t)ci * 27 * d.cout_pad + co0;
        for (int td = 0; td < nd; ++td) {
            const int kd = pd ? (td == 0 ? 0 : 2) : 1;
            const int id = jd + ((pd && td == 0) ? 1 : 0);
            for (int ty = 0; ty < ny; ++ty) {
                const int ky = py ? (ty == 0 ? 0 : 2) : 1;
                const int iy = jy + ((py && ty == 0) ? 1 : 0);
                for (int tx = 0; tx < nx; ++tx) {
                    const int kx = px ? (tx == 0 ? 0 : 2) : 1;
                    const int ix = jx + ((px && tx == 0) ? 1 : 0);
                    float v = 0.0f;
                    if (id < d.Din && iy < d.Hin && ix < d.Win) v = vol[((size_t)id * d.Hin + iy) * d.Win + ix];
                    const float* wt = wrow + ((kd * 3 + ky) * 3 + kx) * d.cout_pad;
#pragma unroll
                    for (int co = 0; co < CO; ++co) acc[co] = fmaf(v, wt[co], acc[co]);
                }
            }
        }
    }
    if (!live) return;
    const int od = 2 * jd + pd, oy = 2 * jy + py, ox = 2 * jx + px;
    conv3d_epilogue<CO>(d, acc, co0, b, ((size_t)od * d.Hout + oy) * d.Wout + ox);
}

// Transposed 3x3x3 convolution, stride 2, output_padding 1 (Deconv3d, module.py:110-144; CostRegNet conv7: 16 -> 8 onto the
// full cost volume) on the matrix cores.  In gather form an output o = 2j + p reads, per axis,
//     p = 0: tap k = 1 at input j            p = 1: tap k = 2 at input j  and  tap k = 0 at input j + 1,
// so every input offset (dz, dy, dx) in {0,1}^3 of a 16(x) x 4(y) x 4(d) tile of INPUT positions j feeds a fixed set of
// output parity classes.  With cout <= 8 the two x-parities share one MFMA: A rows 0-7 = W[.., kx] of class px = 0, rows 8-15 =
// class px = 1 (dx = 0: kx = 1 | 2; dx = 1: zero | kx = 0), B = the input at that offset, 4 input channels per MFMA.  Per
// axis pair (z, y) there are 3 x 3 (offset, parity, tap) combinations -> 18 A slabs per input channel, 18 x cin/4 x 4 rows =
// 288 MFMAs per wave for 16 x 4 x 8 x cout outputs (the direct kernel issued 27 x cin FMAs per output LANE and ran at ~10
// TFLOP/s).  Every output still sums its products in a fixed order (ci groups of 4, then the slab order).
constexpr int kDeconvCin = 16;
__global__ void __launch_bounds__(DMVS_BLOCK) deconv3d_mfma_kernel(const dmvs_conv3d_desc d, int tiles_x, int tiles_y, int tiles_d) {
    constexpr int TX = 16, TY = 4, TD = 4, CK = kDeconvCin;
    constexpr int IW = TX + 1, IH = TY + 1, ID = TD + 1;
    constexpr int PLANE = pad16mod32_3d(ID * IH * IW);
    constexpr int NS = 18;                                   // A slabs per input channel: (cz, cy, dx)
    constexpr int WP = pad16mod32_3d(NS * 16);
    using Halo = HaloMap<ID, IH, IW, PLANE>;
    __shared__ __attribute__((aligned(16))) float lds[CK * PLANE + CK * WP];
    DMVS_LDS_POISON(lds);
    float* const s_w = lds + CK * PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;
    int tile = conv3d_xcd_tile((int)blockIdx.x, (int)gridDim.x, d.tune);
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y; tile /= tiles_y;
    const int td = tile % tiles_d;
    const int b = tile / tiles_d;
    const int vol = d.Din * d.Hin * d.Win, ovol = d.Dout * d.Hout * d.Wout;
    const int jx0 = tx * TX, jy0 = ty * TY, jd0 = td * TD;

    // output channels co_base .. co_base + 7 (blockIdx.y: wider layers -- CostRegNet conv6, 32 -> 16 -- take 8 channels per
    // workgroup); input channels in chunks of CK (single-buffered: 2 chunks at most in the reference's networks)
    const int co_base = blockIdx.y * 8;
    Halo halo;
    halo.init(tid, d.Hin, d.Win);
    f32x4 acc[4][4];                    // [pz * 2 + py][input row mt]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[c][i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
    for (int c0 = 0; c0 < d.cin; c0 += CK) {
    if (c0 > 0) __syncthreads();        // every wave is done with the previous chunk's halo and slabs
    {
        unsigned lo, him1;
        Halo::bounds(jd0, jy0, jx0, d.Din, d.Hin, d.Win, lo, him1);
        const float* origin = d.in + ((size_t)b * d.cin + c0) * vol + ((long)jd0 * d.Hin + jy0) * d.Win + jx0;
#pragma unroll 4
        for (int ci = 0; ci < CK; ++ci) halo.stage(origin + (long)ci * vol, c0 + ci < d.cin, lo, him1, lds + ci * PLANE, wave);
    }
    // A slabs from the gather-form weights [cin][27][cout_pad]: combination c of an axis = (offset, parity, tap):
    // c = 0: (0, 0, k=1), c = 1: (0, 1, k=2), c = 2: (1, 1, k=0)
    for (int e = tid; e < CK * NS * 16; e += DMVS_BLOCK) {
        const int ci = e / (NS * 16), rem = e - ci * (NS * 16);
        const int sidx = rem >> 4, row = rem & 15;
        const int dx = sidx & 1, cy = (sidx >> 1) % 3, cz = (sidx >> 1) / 3;
        const int kd = cz == 0 ? 1 : (cz == 1 ? 2 : 0), ky = cy == 0 ? 1 : (cy == 1 ? 2 : 0);
        const int co = co_base + (row & 7);
        int kx = -1;
        if (row < 8) kx = dx == 0 ? 1 : -1;
        else kx = dx == 0 ? 2 : 0;
        float v = 0.0f;
        if (c0 + ci < d.cin && kx >= 0 && co < d.cout) v = d.weight[((c0 + ci) * 27 + (kd * 3 + ky) * 3 + kx) * d.cout_pad + co];
        s_w[ci * WP + sidx * 16 + row] = v;
    }
    DMVS_DMA_BARRIER();                    // halo (LDS-DMA) and slabs (ds_write) landed

    const int live_c = d.cin - c0 < CK ? d.cin - c0 : CK;
    const int ngroups = (live_c + 3) >> 2;
#pragma unroll 1
    for (int g = 0; g < ngroups; ++g) {
        const int ci = g * 4 + kq;
        const float* wp = s_w + ci * WP + m;
        const float* ipb = lds + ci * PLANE + wave * (IH * IW) + m;
#pragma unroll
        for (int cz = 0; cz < 3; ++cz) {
            const int dz = cz == 2 ? 1 : 0, pz = cz == 0 ? 0 : 1;
#pragma unroll
            for (int cy = 0; cy < 3; ++cy) {
                const int dy = cy == 2 ? 1 : 0, py = cy == 0 ? 0 : 1;
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const float av = wp[((cz * 3 + cy) * 2 + dx) * 16];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const float bv = ipb[(dz * IH + mt + dy) * IW + dx];
                        acc[pz * 2 + py][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[pz * 2 + py][mt], 0, 0, 0);
                    }
                }
            }
        }
    }
    }      // input-channel chunks

    // epilogue: this lane holds x-parity kq >> 1, channels co_base + (kq & 1) * 4 + r of input position (jd0 + wave, jy0 + mt, jx0 + m).
    // The two x-parities of a channel sit 32 lanes apart; stored as they are, every store instruction would write 4-byte
    // elements at an 8-byte stride.  Lanes 0-31 therefore take the (pz, py) classes 0, 1 and lanes 32-63 the classes 2, 3 of
    // BOTH parities (one cross-half exchange per value) and write / read 8-byte pairs: 128 contiguous bytes per 16 lanes.
    const bool lowhalf = kq < 2;
    const int cg0 = co_base + (kq & 1) * 4;
    const int jx = jx0 + m, jd = jd0 + wave;
    const bool live = jx < d.Win && jd < d.Din;
    float* outb = d.out + (size_t)b * d.cout * ovol;
    const float* resb = d.residual ? d.residual + (size_t)b * d.cout * ovol : nullptr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cg = cg0 + r;
        const bool okc = cg < d.cout;
        const float sc = d.scale ? d.scale[okc ? cg : 0] : 1.0f, sh = d.shift ? d.shift[okc ? cg : 0] : 0.0f;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float keep = lowhalf ? acc[c2][mt][r] : acc[c2 + 2][mt][r];
                const float give = lowhalf ? acc[c2 + 2][mt][r] : acc[c2][mt][r];
                const float got = __shfl_xor(give, 32, 64);
                const int c = lowhalf ? c2 : c2 + 2;
                const int jy = jy0 + mt;
                if (!live || !okc || jy >= d.Hin) continue;
                const int o = cg * ovol + ((2 * jd + (c >> 1)) * d.Hout + 2 * jy + (c & 1)) * d.Wout + 2 * jx;      // even: 8-byte aligned
                float v0 = dmvs_act((lowhalf ? keep : got) * sc + sh, d.act);       // x-parity 0
                float v1 = dmvs_act((lowhalf ? got : keep) * sc + sh, d.act);       // x-parity 1
                if (resb) {
                    const float2 rv = *reinterpret_cast<const float2*>(resb + o);
                    v0 += rv.x;
                    v1 += rv.y;
                }
                *reinterpret_cast<float2*>(outb + o) = make_float2(v0, v1);
            }
        }
    }
}

// DMVS_TUNE3D_S2_DIRECT: the round-1 direct VALU kernels for the stride-2 layers (A/B runs)
static bool conv3d_s2_mfma(const dmvs_conv3d_desc& d) { return !(d.tune & DMVS_TUNE3D_S2_DIRECT); }

extern "C" int dmvs_conv3d_f32(const dmvs_conv3d_desc* dp, void* stream) {
    if (!dp) return DMVS_EINVAL;
    const dmvs_conv3d_desc& d = *dp;
    hipStream_t st = (hipStream_t)stream;
    if (d.cout_pad % 8 || d.cout > d.cout_pad || !d.in || !d.weight || !d.out) return DMVS_EINVAL;
    if ((uintptr_t)d.weight & 15) return DMVS_EINVAL;      // the weight slab is staged in 16-byte LDS-DMA pieces
    const int co = (d.cout_pad % 16 == 0) ? 16 : 8;
    dim3 block(DMVS_BLOCK);
    if (d.transposed) {
        if (d.stride != 2 || d.Dout != 2 * d.Din || d.Hout != 2 * d.Hin || d.Wout != 2 * d.Win) return DMVS_EINVAL;
        const long total = (long)d.B * d.Din * d.Hin * d.Win;
        if ((long)d.cout * d.Dout * d.Hout * d.Wout < (1L << 31) && (long)d.cin * d.Din * d.Hin * d.Win < (1L << 31)) {
            // matrix-core form (CostRegNet conv7 16 -> 8, conv6 32 -> 16): 8 output channels per workgroup
            const int tiles_x = (d.Win + 15) / 16, tiles_y = (d.Hin + 3) / 4, tiles_d = (d.Din + 3) / 4;
            dim3 g((unsigned)(tiles_x * tiles_y * tiles_d * d.B), (unsigned)((d.cout + 7) / 8));
            hipLaunchKernelGGL(deconv3d_mfma_kernel, g, block, 0, st, d, tiles_x, tiles_y, tiles_d);
            return dmvs_launch_status();
        }
        dim3 grid(dmvs_ceil_div(total, DMVS_BLOCK), d.cout_pad / co, 8);
        if (co == 16) hipLaunchKernelGGL((deconv3d_kernel<16>), grid, block, 0, st, d);
        else hipLaunchKernelGGL((deconv3d_kernel<8>), grid, block, 0, st, d);
        return dmvs_launch_status();
    }
    if (d.stride != 1 && d.stride != 2) return DMVS_EINVAL;
    const int ed = (d.Din - 1) / d.stride + 1, eh = (d.Hin - 1) / d.stride + 1, ew = (d.Win - 1) / d.stride + 1;
    if (ed != d.Dout || eh != d.Hout || ew != d.Wout) return DMVS_EINVAL;
    const long total = (long)d.B * d.Dout * d.Hout * d.Wout;
    dim3 grid(dmvs_ceil_div(total, DMVS_BLOCK), d.cout_pad / co);
    // the stride-1 kernels address one batch item's input / output block with 32-bit element offsets
    if (d.stride == 1 && ((long)d.cin * d.Din * d.Hin * d.Win >= (1L << 31) || (long)d.cout * d.Dout * d.Hout * d.Wout >= (1L << 31)))
        return DMVS_EINVAL;
    if (d.stride == 1 && d.cout == 1) {
        int XT, YT, DT;
        if (c1_rows_shape(d, XT, YT, DT)) {
            const int tiles_y = (d.Hout + 2 * YT - 1) / (2 * YT), tiles_d = (d.Dout + 2 * DT - 1) / (2 * DT);
            dim3 g((unsigned)(tiles_y * tiles_d * d.B));
            hipLaunchKernelGGL(conv3d_c1_rows_kernel, g, block, 0, st, d, XT, YT, DT, tiles_y, tiles_d);
            return dmvs_launch_status();
        }
        const int tiles_x = (d.Wout + 15) / 16, tiles_y = (d.Hout + 15) / 16, tiles_d = (d.Dout + 15) / 16;
        dim3 g((unsigned)(tiles_x * tiles_y * tiles_d * d.B));
        hipLaunchKernelGGL(conv3d_c1_kernel, g, block, 0, st, d, tiles_x, tiles_y, tiles_d);
        return dmvs_launch_status();
    }
    if (d.stride == 1) {
        const int tiles_x = (d.Wout + 15) / 16, tiles_y = (d.Hout + 3) / 4, tiles_d = (d.Dout + 3) / 4;
        const int ntiles = (d.cout_pad + 15) / 16;
        dim3 g((unsigned)(tiles_x * tiles_y * tiles_d * d.B), (unsigned)((ntiles + 1) / 2));
        const long vtiles = (long)tiles_x * tiles_y * tiles_d * d.B;
        const bool v16 = conv3d_v16_ok(d);
        if (!(d.tune & DMVS_TUNE3D_NO_PAIR) && d.cin <= 4 && d.cout <= 8 && d.cout_pad == 8) {      // 4 -> 8 layers: two output depth slices share the 16 MFMA rows
            const int tiles_d8 = (d.Dout + 7) / 8;
            if ((long)tiles_x * tiles_y * tiles_d8 * d.B >= 512) {
                if (v16) {      // 16-byte halo pieces, the lane's 36 paired weights in registers: 46 KB of LDS, resident count from the occupancy query
                    static const int resident = dmvs_resident_workgroups(reinterpret_cast<const void*>(conv3d_mfma_stream_pair_kernel<true, true>));
                    hipLaunchKernelGGL((conv3d_mfma_stream_pair_kernel<true, true>), dim3((unsigned)resident, 1), block, 0, st, d, tiles_x, tiles_y, tiles_d8);
                } else {
                    hipLaunchKernelGGL((conv3d_mfma_stream_pair_kernel<false, false>), dim3((unsigned)(256 * kPairWgsPerCu), 1), block, 0, st, d, tiles_x, tiles_y, tiles_d8);
                }
                return dmvs_launch_status();
            }
        }
        if (!(d.tune & DMVS_TUNE3D_NO_PAIR) && v16 && d.cin > 4 && d.cin <= 8 && d.cout <= 8 && d.cout_pad == 8) {      // 5..8 -> <= 8 channels (CostRegNet conv1): the paired form over two chunks
            const int tiles_d8 = (d.Dout + 7) / 8;
            if ((long)tiles_x * tiles_y * tiles_d8 * d.B >= 512) {
                static const int resident = dmvs_resident_workgroups(reinterpret_cast<const void*>(conv3d_mfma_stream_pair8_kernel));
                hipLaunchKernelGGL(conv3d_mfma_stream_pair8_kernel, dim3((unsigned)resident, 1), block, 0, st, d, tiles_x, tiles_y, tiles_d8);
                return dmvs_launch_status();
            }
        }
        if (d.cin <= 4 && ntiles == 1 && vtiles >= 256 * kStreamWgsPerCu) {      // one K chunk, many tiles: resident workgroups, pipelined tiles
            dim3 gs((unsigned)(256 * kStreamWgsPerCu), 1);
            if (v16) hipLaunchKernelGGL((conv3d_mfma_stream_kernel<1, true>), dim3(256 * 4, 1), block, 0, st, d, tiles_x, tiles_y, tiles_d);      // 35 KB of LDS each
            else hipLaunchKernelGGL((conv3d_mfma_stream_kernel<1>), gs, block, 0, st, d, tiles_x, tiles_y, tiles_d);
            return dmvs_launch_status();
        }
        if (v16) {
            if (ntiles == 1) hipLaunchKernelGGL((conv3d_mfma_kernel<1, 1, true>), g, block, 0, st, d, tiles_x, tiles_y, tiles_d);
            else hipLaunchKernelGGL((conv3d_mfma_kernel<2, 1, true>), g, block, 0, st, d, tiles_x, tiles_y, tiles_d);
        } else if (ntiles == 1) hipLaunchKernelGGL((conv3d_mfma_kernel<1>), g, block, 0, st, d, tiles_x, tiles_y, tiles_d);
        else hipLaunchKernelGGL((conv3d_mfma_kernel<2>), g, block, 0, st, d, tiles_x, tiles_y, tiles_d);
    } else {
        // stride 2 on the matrix cores (32-bit element offsets inside a batch item, like the stride-1 kernels)
        const bool fits32 = (long)d.cin * d.Din * d.Hin * d.Win < (1L << 31) && (long)d.cout * d.Dout * d.Hout * d.Wout < (1L << 31);
        const int ntiles = (d.cout_pad + 15) / 16;
        if (fits32 && ntiles <= 2 && conv3d_s2_mfma(d)) {
            const int tiles_x = (d.Wout + 15) / 16, tiles_y = (d.Hout + 3) / 4, tiles_d = (d.Dout + 3) / 4;
            dim3 g((unsigned)(tiles_x * tiles_y * tiles_d * d.B), 1);
            if (ntiles == 1) hipLaunchKernelGGL((conv3d_mfma_kernel<1, 2>), g, block, 0, st, d, tiles_x, tiles_y, tiles_d);
            else hipLaunchKernelGGL((conv3d_mfma_kernel<2, 2>), g, block, 0, st, d, tiles_x, tiles_y, tiles_d);
            return dmvs_launch_status();
        }
        if (co == 16) hipLaunchKernelGGL((conv3d_kernel<16, 2>), grid, block, 0, st, d);
        else hipLaunchKernelGGL((conv3d_kernel<8, 2>), grid, block, 0, st, d);
    }
    return dmvs_launch_status();
}

// ------------------------------------------------------------------------------------------
// Weight gradient of the 3x3x3 convolutions (stride 1|2), the 3-D sibling of conv2d_wgrad_kernel:
//   gw[ci][t][co] += sum_voxels dY[co][v] * X[ci][v*S - 1 + tap t]
// MFMA GEMM with the reduction over voxels: A = dY [co = lane&15][k = voxel], B = X [k][j = (ci,t) = lane&15].
// Workgroup = CK input channels x 16 output channels, grid-stride over 16(x) x 4(y) x 4(d) voxel tiles; wave w
// reduces depth slice w.  The transposed layers' weight gradient is the same reduction with the roles of
// input and output gradient swapped (caller-side).
template <int S>
__global__ void __launch_bounds__(DMVS_BLOCK) conv3d_wgrad_kernel(const dmvs_conv3d_desc d, const float* __restrict__ gout,
                                                                  float* __restrict__ ws, int want_bias, int tiles_x, int tiles_y,
                                                                  int tiles_d) {
    constexpr int CK = S == 1 ? 8 : 2;
    constexpr int IW = 15 * S + 3, IH = 3 * S + 3, ID = 3 * S + 3;
    constexpr int PLANE = pad16mod32_3d(ID * IH * IW);
    constexpr int GROW = 257;
    constexpr int NTN = (CK * 27 + 1 + 15) / 16;     // (ci, tap) columns of the chunk + the bias column of ones
    constexpr int IN_IT = (CK * PLANE + DMVS_BLOCK - 1) / DMVS_BLOCK, G_IT = (16 * GROW + DMVS_BLOCK - 1) / DMVS_BLOCK;
    __shared__ float lds[CK * PLANE + 16 * GROW];
    DMVS_LDS_POISON(lds);
    float* s_in = lds;
    float* s_g = lds + CK * PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int c0 = blockIdx.y * CK, cobase = blockIdx.z * 16;
    const size_t ivol = (size_t)d.Din * d.Hin * d.Win, ovol = (size_t)d.Dout * d.Hout * d.Wout;
    const int vol = (int)ivol;
    int boff[NTN];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
        const int jj = nt * 16 + m;
        const int ci = jj / 27, t = jj - ci * 27;
        boff[nt] = jj < CK * 27 ? ci * PLANE + ((t / 9) * IH + (t / 3) % 3) * IW + t % 3 : 0;
    }
    f32x4 acc[NTN];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) acc[nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const float bias_one = (want_bias && blockIdx.y == 0) ? 1.0f : 0.0f;
    const int ntiles = tiles_x * tiles_y * tiles_d * d.B;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int tq = tile;
        const int tx = tq % tiles_x; tq /= tiles_x;
        const int ty = tq % tiles_y; tq /= tiles_y;
        const int td = tq % tiles_d;
        const int b = tq / tiles_d;
        const int x0 = tx * 16, y0 = ty * 4, d0 = td * 4;
        const float* inb = d.in + (size_t)b * d.cin * ivol;
        const float* gb = gout + (size_t)b * d.cout * ovol;
        __syncthreads();
#pragma unroll 2
        for (int i = 0; i < IN_IT; ++i) {
            const int e = i * DMVS_BLOCK + tid;
            if (e < CK * PLANE) {
                const int ci = e / PLANE, rem = e - ci * PLANE;
                const int zz = rem / (IH * IW), rem2 = rem - zz * (IH * IW);
                const int yy = rem2 / IW, xx = rem2 - yy * IW;
                const int gd = d0 * S - 1 + zz, gy = y0 * S - 1 + yy, gx = x0 * S - 1 + xx;
                const bool ok = rem < ID * IH * IW && c0 + ci < d.cin && gd >= 0 && gd < d.Din && gy >= 0 && gy < d.Hin &&
                                gx >= 0 && gx < d.Win;
                const float* src = ok ? inb + ((c0 + ci) * vol + (gd * d.Hin + gy) * d.Win + gx) : dmvs_zero16_3d;
                __builtin_amdgcn_global_load_lds(src, DMVS_LDS(s_in + i * DMVS_BLOCK + wave * 64), 4, 0, 0);
            }
        }
#pragma unroll 2
        for (int i = 0; i < G_IT; ++i) {
            const int e = i * DMVS_BLOCK + tid;
            if (e < 16 * GROW) {
                const int co = e / GROW, p = e - co * GROW;
                const int od = d0 + (p >> 6), oy = y0 + ((p >> 4) & 3), ox = x0 + (p & 15);
                const bool ok = p < 256 && cobase + co < d.cout && od < d.Dout && oy < d.Hout && ox < d.Wout;
                const float* src = ok ? gb + ((size_t)(cobase + co) * ovol + ((size_t)od * d.Hout + oy) * d.Wout + ox) : dmvs_zero16_3d;
                __builtin_amdgcn_global_load_lds(src, DMVS_LDS(s_g + i * DMVS_BLOCK + wave * 64), 4, 0, 0);
            }
        }
        DMVS_DMA_BARRIER();
#pragma unroll 1
        for (int yy = 0; yy < 4; ++yy) {
#pragma unroll
            for (int xg = 0; xg < 4; ++xg) {
                const float av = s_g[m * GROW + wave * 64 + yy * 16 + xg * 4 + kq];
                const float* ip = s_in + ((wave * S) * IH + yy * S) * IW + (xg * 4 + kq) * S;
#pragma unroll
                for (int nt = 0; nt < NTN; ++nt) {
                    float bv = ip[boff[nt]];
                    if (nt == (CK * 27) / 16) bv = m == (CK * 27) % 16 ? bias_one : bv;
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[nt], 0, 0, 0);
                }
            }
        }
    }
    // deterministic block total -> one workspace slot per workgroup (see conv2d_wgrad_kernel)
    __syncthreads();
    float* red = lds;
    static_assert(NTN * 256 <= CK * PLANE + 16 * GROW, "block partial must fit the tile buffers");
#pragma unroll 1
    for (int w = 0; w < DMVS_BLOCK / 64; ++w) {
        if (wave == w) {
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* q = red + (nt * 16 + m) * 16 + kq * 4 + r;
                    *q = w == 0 ? acc[nt][r] : *q + acc[nt][r];
                }
        }
        __syncthreads();
    }
    float* slot = ws + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (NTN * 256);
    for (int e = tid; e < NTN * 256; e += DMVS_BLOCK) slot[e] = red[e];
}

template <int CK>
__global__ void __launch_bounds__(DMVS_BLOCK) conv3d_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ gw,
                                                                         float* __restrict__ gb, int gx, int gy, int cin,
                                                                         int cout) {
    constexpr int T = 27, NTN = (CK * T + 1 + 15) / 16, PER = NTN * 256, SL = 16;
    __shared__ float red[SL][17];
    DMVS_LDS_POISON(red);
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    const int by = blockIdx.y, bz = blockIdx.z;
    const float* base = ws + (size_t)(bz * gy + by) * gx * PER + e;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int x = sl;
    for (; x + 3 * SL < gx; x += 4 * SL) {
        a0 += base[(size_t)x * PER];
        a1 += base[(size_t)(x + SL) * PER];
        a2 += base[(size_t)(x + 2 * SL) * PER];
        a3 += base[(size_t)(x + 3 * SL) * PER];
    }
    for (; x < gx; x += SL) a0 += base[(size_t)x * PER];
    red[sl][el] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0) {
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < SL; ++i) t += red[i][el];
        const int jj = e >> 4, co = bz * 16 + (e & 15);
        const int ci = jj / T, tap = jj - ci * T;
        if (co < cout) {
            if (jj < CK * T && by * CK + ci < cin) gw[((size_t)co * cin + by * CK + ci) * T + tap] = t;
            else if (jj == CK * T && by == 0 && gb) gb[co] = t;
        }
    }
}

struct Wgrad3dGrid {
    int gx, gy, gz, ntn, tiles_x, tiles_y, tiles_d;
    long floats;
};
static Wgrad3dGrid wgrad3d_grid(const dmvs_conv3d_desc& d) {
    Wgrad3dGrid g;
    g.tiles_x = (d.Wout + 15) / 16, g.tiles_y = (d.Hout + 3) / 4, g.tiles_d = (d.Dout + 3) / 4;
    const long ntiles = (long)g.tiles_x * g.tiles_y * g.tiles_d * d.B;
    const int ck = d.stride == 1 ? 8 : 2;
    g.gy = (d.cin + ck - 1) / ck;
    g.gz = (d.cout + 15) / 16;
    long gx = (1024 + g.gy * g.gz - 1) / (g.gy * g.gz);
    if (gx > ntiles) gx = ntiles;
    g.gx = gx < 1 ? 1 : (int)gx;
    g.ntn = (ck * 27 + 1 + 15) / 16;
    g.floats = (long)g.gx * g.gy * g.gz * g.ntn * 256;
    return g;
}
static int wgrad3d_check(const dmvs_conv3d_desc& d) {
    if (d.transposed || (d.stride != 1 && d.stride != 2) || !d.in || d.cout > d.cout_pad || d.B <= 0) return DMVS_EINVAL;
    const int ed = (d.Din - 1) / d.stride + 1, eh = (d.Hin - 1) / d.stride + 1, ew = (d.Win - 1) / d.stride + 1;
    if (ed != d.Dout || eh != d.Hout || ew != d.Wout) return DMVS_EINVAL;
    return 0;
}

extern "C" int dmvs_conv3d_wgrad_workspace_f32(const dmvs_conv3d_desc* dp, int64_t* bytes) {
    if (!dp || !bytes) return DMVS_EINVAL;
    if (int rc = wgrad3d_check(*dp)) return rc;
    *bytes = (int64_t)wgrad3d_grid(*dp).floats * 4;
    return 0;
}

extern "C" int dmvs_conv3d_wgrad_f32(const dmvs_conv3d_desc* dp, const float* grad_out, float* gw, float* gb, float* workspace,
                                     int64_t workspace_bytes, void* stream) {
    if (!dp || !grad_out || !gw || !workspace) return DMVS_EINVAL;
    const dmvs_conv3d_desc& d = *dp;
    if (int rc = wgrad3d_check(d)) return rc;
    const Wgrad3dGrid g = wgrad3d_grid(d);
    if (workspace_bytes < (int64_t)g.floats * 4) return DMVS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(g.gx, g.gy, g.gz), block(DMVS_BLOCK), rgrid(g.ntn * 16, g.gy, g.gz);
    if (d.stride == 1) {
        hipLaunchKernelGGL((conv3d_wgrad_kernel<1>), grid, block, 0, st, d, grad_out, workspace, gb ? 1 : 0, g.tiles_x, g.tiles_y, g.tiles_d);
        hipLaunchKernelGGL((conv3d_wgrad_reduce_kernel<8>), rgrid, block, 0, st, workspace, gw, gb, g.gx, g.gy, d.cin, d.cout);
    } else {
        hipLaunchKernelGGL((conv3d_wgrad_kernel<2>), grid, block, 0, st, d, grad_out, workspace, gb ? 1 : 0, g.tiles_x, g.tiles_y, g.tiles_d);
        hipLaunchKernelGGL((conv3d_wgrad_reduce_kernel<2>), rgrid, block, 0, st, workspace, gw, gb, g.gx, g.gy, d.cin, d.cout);
    }
    return dmvs_launch_status();
}
