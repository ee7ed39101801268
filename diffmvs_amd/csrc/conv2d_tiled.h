// The tiled implicit-GEMM kernel of conv2d.hip and its dispatch, as a header: the instantiations of one (KH, KW, stride) family are
// compiled in their own translation unit (conv2d_k*.hip) so that the ~500 kernel instantiations build in parallel -- as one file
// they were 6 of the 7 minutes of a library build.  Everything here has internal linkage; a translation unit exports plain
// functions dmvs_detail::launch_conv2d_<kh><kw><stride>() (declared at the end of this file) for the entry point in conv2d.hip.
#pragma once
#include <type_traits>

#include "dmvs_common.h"
#include "dmvs_lds_poison.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#include "dmvs_bf16.h"

namespace {


constexpr int pad16mod32(int n) {   // smallest m >= n with m % 32 == 16
    int m = n;
    while (m % 32 != 16) ++m;
    return m;
}

// all-zero source for LDS-DMA lanes that stage padding (an LDS-DMA lane cannot write a literal)
__device__ __attribute__((aligned(16))) const float dmvs_zero16[4] = {0.0f, 0.0f, 0.0f, 0.0f};

#define DMVS_LDS(p) ((__attribute__((address_space(3))) void*)(p))

template <int KH, int KW, int S, int NT, int MT, int AR = DMVS_ARITH_F32, int WX = 1, bool V16 = false>
struct ConvCfg {
    static constexpr int T = KH * KW;
    static constexpr int COLS = 16 * WX;                      // pixel tile of the workgroup: the 4 waves sit WX across, 4 / WX down
    static constexpr int ROWS = (4 / WX) * MT;
    static constexpr int TW = (COLS - 1) * S + KW, TH = (ROWS - 1) * S + KH;
    // V16 (16-byte staging pieces, see the kernel): an LDS row starts SLACK floats left of the halo's first column, on a 16-byte
    // boundary of the image row (tile origins are multiples of 16 pixels, the padding is (KW - 1) / 2), and its pitch TWL is a
    // whole number of pieces
    static constexpr int SLACK = V16 ? (4 - ((KW - 1) / 2) % 4) % 4 : 0;
    static constexpr int TWL = V16 ? (SLACK + TW + 3) / 4 * 4 : TW;
    static constexpr int PLANE = pad16mod32(TH * TWL);
    static constexpr int NW = NT * 16;
    static constexpr int WPAD = pad16mod32(T * NW);
    // input channels per LDS chunk: 8 when the double-buffered chunk stays within 40 KB, else 4 (the bf16 form: always 8,
    // its matrix instruction spans 8 channels); decided on the 4-byte form's plane so that both staging forms chunk alike
    // (4-channel chunks for the one-n-tile 3x3 layers -- half the LDS, 8 instead of 5 workgroups per CU -- measured 4-6 % slower)
    static constexpr int CK = AR != DMVS_ARITH_F32 ? 8 : ((2 * 8 * (pad16mod32(TH * TW) + WPAD) * 4 > 40960) ? 4 : 8);
    // (DMVS_ARITH_SPLIT) the chunk's halo tile as three bf16 planes [plane][TH * TW positions][8 channels]: 16 bytes per position and plane
    static constexpr int QPOS = TH * TW;
    static constexpr int NG = (T + 3) / 4;                     // tap groups of the bf16 forms: K = 32 = 8 channels x 4 taps
    static constexpr bool SPLIT = AR == DMVS_ARITH_SPLIT;
    // floats per pipeline stage (the split form keeps no weight slab in LDS: its pre-split weights go from global memory to registers)
    static constexpr int BUF = CK * (PLANE + (SPLIT ? 0 : WPAD));
    static constexpr int NBUF = SPLIT ? 1 : 2;                 // the split form converts a staged chunk at once and then re-uses the one buffer
    static constexpr int LDS_FLOATS = NBUF * BUF + 32 + (SPLIT ? 3 * TH * TW * 4 : 0);
    static constexpr int IN_IT = (CK * PLANE + DMVS_BLOCK - 1) / DMVS_BLOCK;           // 4-byte DMA pieces per thread
    static constexpr int W_IT = (CK * WPAD / 4 + DMVS_BLOCK - 1) / DMVS_BLOCK;         // 16-byte DMA pieces per thread
};

// ZI = the zero-insert input mode (training only: input gradient of a stride-2 layer).  It is a separate instantiation
// because its extra predicate in the staging path costs the inference kernels scalar-register spills (measured: +30 %
// on the 3x3 NT=1 MT=4 kernel when it was a run-time branch of the same code).
// minimum waves per SIMD the register allocator must leave room for: the 32->32 (NT=2, MT=4) and 64->64 (NT=4, MT=2)
// shapes otherwise settle at 160 / 212 VGPRs = 3 / 2 waves, too few to cover the per-chunk barrier + DMA latency
constexpr int conv_min_waves(int nt, int mt, int ar = DMVS_ARITH_F32) {
    if (ar == DMVS_ARITH_SPLIT) return nt <= 2 ? 4 : 2;      // accumulators + the tap group's weights (and the next group's) in registers
    return (nt == 2 && mt == 4) ? 4 : ((nt == 4 && mt == 2) ? 3 : 1);
}

// OT = element type of a channel-last output (DMVS_DTYPE_*): 16-bit feature storage is its own instantiation so that the
// fp32 kernels keep their register allocation.
//
// WALK = resident, tile-walking workgroups: the grid holds only as many workgroups as fit the chip at once and workgroup w
// takes tiles w, w + gridDim.x, ...; while the matrix cores sweep the LAST channel chunk of a tile, the FIRST chunk of the
// workgroup's next tile already streams into the other LDS buffer, and that tile's index decode happens under those MFMAs.
// A one-tile-per-workgroup launch pays, per tile, the index decode + the full HBM/L2 latency of its first chunk + the
// epilogue with the matrix pipe idle; with 2-4 chunks per tile (the 16- and 32-channel layers) that is half of a tile's
// life, and it does not average out over co-resident workgroups because a launch starts them all in the same phase
// (SQ PMC on the 16 -> 16 layer: matrix pipe busy 41 % of the cycles, waves waiting 27 %).  Same arithmetic, same order.
// First attempt (every NCHW layer walked, all fused paths in the loop): correct but SLOWER -- the B=96 step 83.9 -> 90.9 ms
// (profiles/r3_conv_walk_ab.txt): 112 instead of 63 VGPRs (3 instead of 5 waves per SIMD) and 217 SGPR-spill reads on the
// <3,3,1,1,4> instantiation.  Second form (this one): compiled for the "lean" layers only (kLean below: the other fused paths are
// compiled out) with the lane coordinates redefined opaquely per tile so that hipcc does not hoist the chunk loop's and the
// epilogue's lane-dependent addresses out of the tile loop: 59 VGPRs, 46 spill accesses.
//
// AR = DMVS_ARITH_BF16: the same kernel -- same fp32 tensors, same LDS-DMA staging of fp32 tiles, same epilogue -- with the
// operands rounded to bf16 (nearest even, v_cvt_pk_bf16_f32) as they leave LDS and v_mfma_f32_16x16x32_bf16 (fp32
// accumulation) in place of the fp32 MFMA.  One bf16 MFMA spans K = 32 = the 8 input channels of an LDS chunk x 4 TAPS: lane
// group kq carries tap 4g + kq (taps beyond KH*KW meet zero weights), its 8 k-slots are the 8 channels.  3 MFMAs replace the
// 18 fp32 ones of a 3x3 chunk, at half the cycles each; the loop is then bound by its (unchanged) 8 LDS reads per operand.
//
// V16 = the input halo is staged in 16-BYTE pieces (global_load_lds_dwordx4) instead of 4-byte ones.  An LDS-DMA instruction costs
// the texture path the same ~55-60 cycles whatever its width (measured on the cout = 1 3-D kernel), and the 4-byte
// form needs one wave-level instruction per 64 halo floats: 96 + 10 per 16 -> 16 tile, ~6.0 k cycles of a CU's DMA issue against 4.6 k
// matrix cycles per SIMD -- which is why those layers moved with NONE of: bf16 matrix arithmetic, register staging, tile width,
// chunk size, tile walking (section 4.0).  Here an LDS row is the 16-byte aligned cover of the halo row (SLACK extra floats on the
// left, pitch TWL: 24 instead of 18 floats for a 3x3 tile), so the image is a plain sequence of pieces: 2 wave instructions per
// channel instead of 6.  A piece lies entirely inside or entirely outside the image (rows are multiples of 4 floats; the
// dispatcher checks that, the 16-byte alignment of the bases, a PLAIN input and the usual "same" padding), so zero padding stays
// all-or-nothing per piece.  The slack columns receive neighbouring pixels that no MFMA reads.  Same arithmetic, same order.
//
// WX = waves side by side in a workgroup's pixel tile (1: 16 x 16*MT pixels, 2: 32 x 8*MT): 128-byte instead of 64-byte runs
// per channel row in the stores and the halo reads.  Worth 6-9 % on the two-n-tile layers of the large planes, a loss on the
// others (conv_tile_waves_x); it is NOT what holds the 16-channel layers at ~0.5 (unchanged by it, as by a 6x cut of the
// matrix time and by register staging: DESIGN.md section 4).
//
// LEAN (round 4) = the "lean" specialisation WITHOUT the tile walk: the generic instantiation resolves every fused path (second
// input, gating, GRU blend, GroupNorm statistics, residual modes, five activations, post-scale) with wave-uniform run-time branches,
// per (row, n-tile) of the epilogue and per chunk of the staging -- the 16 -> 16 generic kernel is 2517 vector + 2481 scalar static
// instructions for 36 MFMAs, and its SQ counters show 747 scalar + 571 vector instructions per 144 MFMAs per wave at run time.  The
// plain layers (FeatureNet / ContextNet trunks, encoder, heads: ~65 % of the conv2d time) need none of it.
template <int KH, int KW, int S, int NT, int MT, bool ZI, int OT = DMVS_DTYPE_F32, bool TR = false, bool WALK = false, int AR = DMVS_ARITH_F32,
          int WX = 1, bool V16 = false, bool LEAN = false>
__global__ void __launch_bounds__(DMVS_BLOCK, conv_min_waves(NT, MT, AR)) conv2d_mfma_kernel(const dmvs_conv2d_desc d, int tiles_x, int tiles_y) {
    using Cfg = ConvCfg<KH, KW, S, NT, MT, AR, WX, V16>;
    static_assert(!(V16 && ZI), "16-byte staging pieces: PLAIN inputs only");
    constexpr int T = Cfg::T, TW = Cfg::TW, TH = Cfg::TH, PLANE = Cfg::PLANE, NW = Cfg::NW, WPAD = Cfg::WPAD;
    constexpr int TWL = Cfg::TWL, SLACK = Cfg::SLACK;      // LDS row pitch and the halo's first column inside an LDS row
    constexpr int CK = Cfg::CK, BUF = Cfg::BUF, IN_IT = Cfg::IN_IT, W_IT = Cfg::W_IT;
    // one LDS object on purpose (tile buffers + the 32-float GroupNorm scratch): with separate objects hipcc orders reads of
    // one against LDS-DMA into another with vmcnt(0) waits (conv3d.hip, conv3d_mfma_stream_kernel)
    __shared__ __attribute__((aligned(16))) float lds[Cfg::LDS_FLOATS];
    constexpr bool kSplit = AR == DMVS_ARITH_SPLIT;
    constexpr int NBUF = Cfg::NBUF;
    DMVS_LDS_POISON(lds);

    int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA bases in SGPRs
    int m = lane & 15, kq = lane >> 4;       // (tid, m, kq not const: the tile-walking form redefines them per tile, see the tile loop)
    const int wx = wave % WX, wy = wave / WX;      // this wave's 16-pixel column block and row block inside the tile
    int tile = blockIdx.x;   // round-robin over XCDs: an XCD-contiguous remap measured 6-8 % SLOWER here (HBM channel spread)
    const int ntiles = tiles_x * tiles_y * d.B;
    // Which tiles meet in one XCD's L2 (round 6; dmvs_common.h dmvs_xcd_grouped_block): a 16-pixel fp32 tile row is 64 bytes -- half a cache line
    // -- and x-adjacent tiles share halo columns, so under the plain round-robin dispatch every line of the input is fetched, and every line
    // of the output written, by two XCDs.  Groups of 4 (one-n-tile 3x3 layers: 8) x-adjacent tiles per XCD, measured over plain round
    // robin / 2 / 4 / 8 / a tile row / two rows / an image (profiles/r6_conv_xcd_group_ab.json): the 16 -> 16 layers -3 ... -10 %, the
    // convolutions of a B = 96 step 52.9 -> 52.2 ms; whole rows or images per XCD give half of that.  `tile` stays the dispatch index
    // (the walking form strides it); the remap happens at decode.  A bijection of the tile indices: bit-identical results.
    const int xg = (d.tune >> 14) & 7;
    const unsigned xgroup = xg == 0 ? ((NT == 1 && KH * KW == 9) ? 8u : 4u)
                                    : (xg <= 4 ? 1u << (xg - 1) : (unsigned)(xg == 5 ? tiles_x : (xg == 6 ? 2 * tiles_x : tiles_x * tiles_y)));
    // s_* / gy0 / gx0: the tile whose input is being STAGED; b / ox0 / oy0 (set at the top of the tile loop): the tile being
    // computed and stored.  They differ only while WALK prefetches the next tile under the last chunk of the current one.
    int s_b, s_ox0, s_oy0, gy0, gx0;
    auto decode_tile = [&](int t) {
        t = (int)dmvs_xcd_grouped_block((unsigned)t, (unsigned)ntiles, xgroup);
        const int tx = t % tiles_x;
        t /= tiles_x;
        const int ty = t % tiles_y;
        s_b = t / tiles_y;
        s_ox0 = tx * Cfg::COLS;
        s_oy0 = ty * Cfg::ROWS;
        gy0 = s_oy0 * S - d.pad_h;
        gx0 = s_ox0 * S - d.pad_w;
    };
    decode_tile(tile);
    const int nbase = blockIdx.y * NW;
    // The tile-walking form is built for the PLAIN layers only (FeatureNet / ContextNet trunks, the plain Unet layers): one input
    // tensor, no gating / GRU blend / GroupNorm statistics, ReLU or no activation, optional same-size residual, 16-byte stores.
    // Compiling the other paths out is what lets two tiles' state fit the register file (the dispatcher checks the conditions).
    constexpr bool kLean = WALK || LEAN;
    // (LEAN keeps two things the walking form compiles out: a second PLAIN concat input and the GroupNorm statistics -- the Unet's
    // weight-standardised convolutions are "plain" otherwise)
    const int cin = WALK ? d.c0 : d.c0 + d.c1;

    // ---- addressing of the logical input, all in 32-bit element offsets from per-batch bases
    const int mode = ZI ? DMVS_IN_UPSAMPLE2 : (kLean ? DMVS_IN_PLAIN : d.in_mode);      // zero-insert addresses like nearest-x2 (plus a parity predicate)
    const int pW = mode == DMVS_IN_UPSAMPLE2 ? (d.Win >> 1) : (mode == DMVS_IN_UNSHUFFLE2 ? (d.Win << 1) : d.Win);
    const int pH = mode == DMVS_IN_UPSAMPLE2 ? (d.Hin >> 1) : (mode == DMVS_IN_UNSHUFFLE2 ? (d.Hin << 1) : d.Hin);
    const int plane0 = pH * pW, plane1 = d.Hin * d.Win;
    const int pc0 = mode == DMVS_IN_UNSHUFFLE2 ? (d.c0 >> 2) : d.c0;
    const float *in0b, *mul0b, *in1b;      // per-batch-item input bases of the tile being staged
    auto set_bases = [&]() {
        in0b = d.in0 + (size_t)s_b * (d.in0_cstride ? d.in0_cstride : pc0) * plane0;
        mul0b = (!kLean && d.mul0) ? d.mul0 + (size_t)s_b * (d.gate_cstride ? d.gate_cstride : pc0) * plane0 : nullptr;
        in1b = (!WALK && d.in1) ? d.in1 + (size_t)s_b * d.c1 * plane1 : d.in0;
    };
    set_bases();
    static_assert(!((WALK || LEAN) && ZI), "the lean forms are inference kernels");

    // element e of the padded LDS input image of chunk c0 -> global source (or nullptr for padding)
    auto in_src = [&](int c0, int e, int& off_out) -> const float* {
        const int ci = e / PLANE, rem = e - ci * PLANE;
        const int r = rem / TWL, c = rem - r * TWL - SLACK;
        const int cig = c0 + ci, iy = gy0 + r, ix = gx0 + c;
        const bool ok = rem < TH * TWL && cig < cin && iy >= 0 && iy < d.Hin && ix >= 0 && ix < d.Win && !(ZI && ((iy | ix) & 1));
        int off;
        if (mode == DMVS_IN_PLAIN) off = cig * plane0 + iy * pW + ix;
        else if (mode == DMVS_IN_UPSAMPLE2) off = cig * plane0 + (iy >> 1) * pW + (ix >> 1);
        else off = (cig >> 2) * plane0 + (iy * 2 + ((cig >> 1) & 1)) * pW + ix * 2 + (cig & 1);
        off_out = (ok && cig < d.c0) ? off : -1;
        if (!ok) return nullptr;
        return cig < d.c0 ? in0b + off : in1b + ((cig - d.c0) * plane1 + iy * d.Win + ix);
    };

    // Staging map.  The LDS image of a chunk is [CK][PLANE]; a lane stages the SAME positions of every channel plane
    // (plane iteration it -> position it*256 + tid), so the (row, column) decode, the image-border test and the spatial
    // offset are computed ONCE per tile for P_IT positions -- not for every (channel, position) element of a chunk.
    // (Decoding e -> (ci, row, col) with its integer divisions in every chunk was ~80 % of the non-MFMA instructions;
    // decoding it once per tile for all CK*PLANE/256 elements was still ~600 VALU per workgroup: a third of the issue
    // slots of the 16-channel layers.)  The channel is then a scalar loop variable: the DMA address is an SGPR base
    // plus a 32-bit lane offset.
    constexpr int PIECES = TH * TWL / 4;                  // (V16) 16-byte pieces of a channel plane; piece p = LDS floats 4p .. 4p+3
    constexpr int P_IT = V16 ? (PIECES + DMVS_BLOCK - 1) / DMVS_BLOCK : (PLANE + DMVS_BLOCK - 1) / DMVS_BLOCK;
    int p_sp[P_IT];                        // spatial source offset of plane position (V16: piece) it*256 + tid, -1: padding / not staged
    auto map_tile = [&]() {                // for the tile (gy0, gx0) about to be staged
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            if constexpr (V16) {
                const int pe = it * DMVS_BLOCK + tid;
                const int r = pe / (TWL / 4), pc = pe - r * (TWL / 4);
                const int iy = gy0 + r, ix = gx0 - SLACK + 4 * pc;      // a multiple of 4: the piece is inside or outside as a whole
                const bool ok = pe < PIECES && iy >= 0 && iy < d.Hin && ix >= 0 && ix < d.Win;
                p_sp[it] = ok ? iy * pW + ix : -1;
                continue;
            }
            const int rem = it * DMVS_BLOCK + tid;
            const int r = rem / TW, c = rem - r * TW;
            const int iy = gy0 + r, ix = gx0 + c;
            const bool ok = rem < TH * TW && iy >= 0 && iy < d.Hin && ix >= 0 && ix < d.Win && !(ZI && ((iy | ix) & 1));
            int sp;
            if (mode == DMVS_IN_PLAIN) sp = iy * pW + ix;
            else if (mode == DMVS_IN_UPSAMPLE2) sp = (iy >> 1) * pW + (ix >> 1);
            else sp = iy * 2 * pW + ix * 2;
            p_sp[it] = ok ? sp : -1;
        }
    };
    map_tile();
    int w_ci[W_IT], w_off[W_IT];           // weight slab piece -> (channel within the chunk, offset inside its [T][cout_pad] block)
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        const int e4 = i * DMVS_BLOCK + tid;
        const int ci = e4 / (WPAD / 4), rem4 = e4 - ci * (WPAD / 4);
        const int t = rem4 / (NW / 4), n4 = rem4 - t * (NW / 4);
        const bool ok = e4 < CK * WPAD / 4 && rem4 < T * NW / 4 && nbase + n4 * 4 < d.cout_pad;
        w_ci[i] = ci;
        w_off[i] = ok ? t * d.cout_pad + nbase + n4 * 4 : -1;
    }

    // Stage chunk c0 into `buf` with LDS-DMA (global_load_lds): no VGPR round trip, fully asynchronous.
    // A wave-instruction fills 64 consecutive LDS words (4-byte form, input halo tile -- its rows are
    // not 16-byte multiples) or 64 consecutive 16-byte slots (weight slab) from per-lane sources.
    // Padding is written ONCE per workgroup, not per chunk: spatial padding stays padding in every chunk, and channels
    // beyond cin only ever meet zero weights, so they just must not hold non-finite LDS garbage on their first use.
    // The per-chunk DMA then touches valid elements only (exec-masked), with no zero-source pointer to select.
    // An interior tile of a layer whose channel count is a multiple of the chunk has no such position that an MFMA reads: the pass
    // (2 x CK predicated LDS writes per plane position: ~40 VALU + ~170 SALU per wave, a quarter of the scalar work of a 16 -> 16
    // tile) is skipped then -- workgroup-uniform branch.  (A walking workgroup's later border tiles clear their own padding.)
    // Measured on the MI355X (profiles/r4_optins_ab.jsonl, bit-identical results): 16 -> 16 at 96 x 256 x 320 532 -> 471 us, the other
    // layer shapes within 2 %.
    const bool pad_pass = ZI || (cin & (CK - 1)) != 0 || gy0 < 0 || gx0 < 0 || gy0 + TH > d.Hin || gx0 + TW > d.Win;
    if (pad_pass) {
#pragma unroll
    for (int it = 0; it < P_IT; ++it) {
        const int rem = it * DMVS_BLOCK + tid;
        if constexpr (V16) {
            if (rem < PIECES) {
                const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int ci = 0; ci < CK; ++ci) {
                    if (p_sp[it] < 0 || ci >= cin) *reinterpret_cast<f32x4*>(&lds[ci * PLANE + 4 * rem]) = z4;
                    if (NBUF == 2 && (p_sp[it] < 0 || CK + ci >= cin)) *reinterpret_cast<f32x4*>(&lds[BUF + ci * PLANE + 4 * rem]) = z4;
                }
            }
        } else if (rem < PLANE) {
#pragma unroll
            for (int ci = 0; ci < CK; ++ci) {
                if (p_sp[it] < 0 || ci >= cin) lds[ci * PLANE + rem] = 0.0f;
                if (NBUF == 2 && (p_sp[it] < 0 || CK + ci >= cin)) lds[BUF + ci * PLANE + rem] = 0.0f;
            }
        }
    }
    }
    const bool simple = WALK || (d.c1 == 0 && mode != DMVS_IN_UNSHUFFLE2);      // one input tensor: the channel base just advances by a plane
    auto stage_as = [&](auto simple_tag, int c0, float* buf) __attribute__((always_inline)) {
        constexpr bool kSimple = decltype(simple_tag)::value;          // two instantiations: no mode decisions inside the simple one
        const float* cb = in0b + (size_t)c0 * plane0;                  // wave-uniform base of the channel being staged
#pragma unroll
        for (int ci = 0; ci < CK; ++ci) {
            const int cig = c0 + ci;
            if (cig < cin) {
                if constexpr (!kSimple) {
                    if (cig >= d.c0) cb = in1b + (size_t)(cig - d.c0) * plane1;        // second concat input: always PLAIN
                    else if (mode == DMVS_IN_UNSHUFFLE2) cb = in0b + ((size_t)(cig >> 2) * plane0 + ((cig >> 1) & 1) * pW + (cig & 1));
                    else cb = in0b + (size_t)cig * plane0;
                }
#pragma unroll
                for (int it = 0; it < P_IT; ++it) {
                    if (p_sp[it] >= 0) {
                        const float* srcp = cb + (unsigned)p_sp[it];
                        if constexpr (V16) {
                            float* dstp = buf + ci * PLANE + (it * DMVS_BLOCK + wave * 64) * 4;
                            __builtin_amdgcn_global_load_lds(srcp, DMVS_LDS(dstp), 16, 0, 0);
                        } else {
                            float* dstp = buf + ci * PLANE + it * DMVS_BLOCK + wave * 64;
                            __builtin_amdgcn_global_load_lds(srcp, DMVS_LDS(dstp), 4, 0, 0);
                        }
                    }
                }
            }
            cb += plane0;
        }
        if constexpr (kSplit) return;                               // (no weight slab in LDS)
        float* wbuf = buf + CK * PLANE;
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
            if (i * DMVS_BLOCK + tid < CK * WPAD / 4) {
                const int cw = c0 + w_ci[i];
                const float* srcp = (w_off[i] >= 0 && cw < cin) ? d.weight + (cw * T * d.cout_pad + w_off[i]) : dmvs_zero16;
                float* dstp = wbuf + (i * DMVS_BLOCK + wave * 64) * 4;
                __builtin_amdgcn_global_load_lds(srcp, DMVS_LDS(dstp), 16, 0, 0);
            }
        }
    };
    auto stage = [&](int c0, float* buf) __attribute__((always_inline)) {
        if (simple) stage_as(std::true_type{}, c0, buf);
        else stage_as(std::false_type{}, c0, buf);
    };

    // padding positions of a BORDER tile in a buffer about to be re-staged: a walking workgroup's previous tile left data there
    auto zero_padding = [&](float* buf) {
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int rem = it * DMVS_BLOCK + tid;
            if constexpr (V16) {
                if (rem < PIECES && p_sp[it] < 0) {
#pragma unroll
                    for (int ci = 0; ci < CK; ++ci) *reinterpret_cast<f32x4*>(&buf[ci * PLANE + 4 * rem]) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
            } else if (rem < TH * TW && p_sp[it] < 0) {
#pragma unroll
                for (int ci = 0; ci < CK; ++ci) buf[ci * PLANE + rem] = 0.0f;
            }
        }
    };
    float* const gn_scratch = lds + NBUF * BUF;      // (not the tile buffers: a walking workgroup's next tile is streaming into them)

    stage(0, lds);
    int cur = 0;
    bool border = false;                   // the tile being staged has padding positions (workgroup-uniform)
    for (;;) {      // tiles of this workgroup (one unless WALK)
    if constexpr (WALK) {
        // Opaque redefinition of the lane coordinates: without it hipcc hoists every lane-dependent address of the chunk loop and
        // the epilogue out of the tile loop and keeps them all live (112 instead of 63 VGPRs, 3 instead of 5 waves per SIMD)
#ifndef DMVS_HOST_EMULATION
        asm volatile("" : "+v"(m), "+v"(kq), "+v"(tid));
#endif
    }
    const int b = s_b, ox0 = s_ox0, oy0 = s_oy0;
    f32x4 acc[MT][NT];
    // (split arithmetic: ONE accumulator per output.  A separate one for the five small partial products would keep their roundings relative to a
    // sum 2^-8 the size; but the matrix instruction rounds once per K = 32, not once per product as an fma chain does, so six roundings per 32
    // products are already fewer than the chain's 32 -- measured on the MI355X against fp64 the single accumulator is as close as the fp32 kernel,
    // profiles/r6_split_accuracy.txt -- and the 32 registers were the difference between 3 and 4 waves per SIMD: built, measured, removed.)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    // (split arithmetic) this lane's pre-split weights of one tap group: 16 bytes = the 8 channels of the chunk for (plane, tap 4g + kq, cout)
    typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
    [[maybe_unused]] u32x4s a_cur[kSplit ? 3 : 1][kSplit ? NT : 1], a_next[kSplit ? 3 : 1][kSplit ? NT : 1];
    [[maybe_unused]] const u32x4s* const wsp = reinterpret_cast<const u32x4s*>(d.weight_split);
    [[maybe_unused]] int a_co[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) a_co[nt] = min(nbase + nt * 16 + m, d.cout_pad - 1) + kq * d.cout_pad;      // (a channel tile beyond cout_pad: discarded by the epilogue)
    [[maybe_unused]] auto load_a = [&](u32x4s (&a)[kSplit ? 3 : 1][kSplit ? NT : 1], int chunk, int g) __attribute__((always_inline)) {
        if constexpr (kSplit) {
            const u32x4s* base = wsp + (size_t)((chunk * Cfg::NG + g) * 3) * 4 * d.cout_pad;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) a[pl][nt] = base[pl * 4 * d.cout_pad + a_co[nt]];
        }
    };
    constexpr bool kPrefetchA = kSplit && NT <= 2;      // the next tap group's weights in a second register set (48 registers at two n-tiles)
    if constexpr (kSplit) load_a(a_cur, 0, 0);
    for (int c0 = 0; c0 < cin; c0 += CK, cur ^= 1) {
        float* s_in = lds + (kSplit ? 0 : cur) * BUF;
        float* s_w = s_in + CK * PLANE;
        // drains this wave's LDS-DMA (explicit vmcnt(0), dmvs_common.h) and orders it against everyone's ds_reads:
        // after it, chunk c0 is complete in `cur` and the other buffer is free for the next chunk
        DMVS_DMA_BARRIER();
        if (!kLean && mul0b) {   // r*h gating of the GRU candidate conv: scale the staged in0 channels in place
            for (int i = 0; i < IN_IT; ++i) {
                const int e = i * DMVS_BLOCK + tid;
                if (e < CK * PLANE) {
                    int off;
                    in_src(c0, e, off);
                    if (off >= 0) s_in[e] *= mul0b[off];
                }
            }
            __syncthreads();
        }
        if constexpr (!kSplit) {
        float* other = lds + (cur ^ 1) * BUF;
        if (c0 + CK < cin) {
            if (WALK && border) zero_padding(other);
            stage(c0 + CK, other);   // lands while the matrix cores chew on chunk c0
        } else if (WALK && tile + (int)gridDim.x < ntiles) {
            // last chunk of this tile: the workgroup's next tile starts streaming now
            decode_tile(tile + (int)gridDim.x);
            set_bases();
            map_tile();
            border = gy0 < 0 || gx0 < 0 || gy0 + TH > d.Hin || gx0 + TW > d.Win;
            if (border) zero_padding(other);
            stage(0, other);
        }
        }
        if constexpr (AR == DMVS_ARITH_SPLIT) {
            // Split-bf16 arithmetic with fp32 accuracy (round 6).  The fp32 MFMA runs at the vector ALU's rate and -- measured with
            // tools/calib/overlap_probe.hip -- a CU executing it makes NO progress on vector-memory instructions (fp32 MFMAs and HBM streaming
            // in one CU take the SUM of their times; bf16 MFMAs and VALU work overlap with memory), which is why every fp32 layer's time is
            // its matrix time plus its memory time.  Here every fp32 operand is split into three bf16 values (hi + mid + lo = x to 2^-27) and the
            // product a * b is formed as the six partial products down to 2^-18 (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi; the three
            // dropped ones are below 2^-26 of the product) on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: 6 x 16 cycles per K = 32
            // instead of 8 x 32, and the staging DMA proceeds underneath.  K = 32 = the chunk's 8 channels x 4 taps, as in the bf16 form.
            //   pass 1: the staged fp32 halo tile -> three bf16 planes in LDS [plane][position][8 channels], ONCE per element (not once per
            //           tap that reads it); the one fp32 buffer is then free, and the next chunk streams into it under this chunk's MFMAs;
            //   pass 2: per tap group the lane's weights arrive PRE-SPLIT from global memory (d.weight_split, packed once per layer: 3 x NT
            //           16-byte loads, the next group's in flight), the pixel operand is three 16-byte LDS reads: no VALU work in the loop.
            static_assert(CK == 8, "the split form maps the 8 channels of an LDS chunk onto the 8 k-slots of a lane");
            u32x4s* const q_hi = reinterpret_cast<u32x4s*>(lds + NBUF * BUF + 32);
            u32x4s* const q_mid = q_hi + Cfg::QPOS;
            u32x4s* const q_lo = q_mid + Cfg::QPOS;
            for (int e = tid; e < Cfg::QPOS; e += DMVS_BLOCK) {
                const int r = e / TW, c = e - r * TW;
                const float* sp = s_in + r * TWL + SLACK + c;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = sp[j * PLANE];
                bf16x8 h8, m8, l8;
                dmvs_split3_bf16x8(v, h8, m8, l8);
                q_hi[e] = __builtin_bit_cast(u32x4s, h8);
                q_mid[e] = __builtin_bit_cast(u32x4s, m8);
                q_lo[e] = __builtin_bit_cast(u32x4s, l8);
            }
            DMVS_LDS_BARRIER();        // the planes are complete and the fp32 buffer is free (ds accesses only: nothing is in flight)
            if (c0 + CK < cin) stage(c0 + CK, lds);      // lands while the matrix cores chew on chunk c0
            const int chunk = c0 / CK;
            auto tap_group = [&](int g, const u32x4s (&au)[kSplit ? 3 : 1][kSplit ? NT : 1]) __attribute__((always_inline)) {
                const int t = 4 * g + kq;
                const int tc = t < T ? t : T - 1;      // (a tap slot beyond the kernel meets zero weights)
                const int ky = tc / KW, kx = tc - ky * KW;
                const int qbase = ((wy * MT * S) + ky) * TW + (wx * 16 + m) * S + kx;
                // RB rows at a time, the partial product outermost: the RB * NT MFMAs of one product are independent, so an accumulator is
                // re-used RB * NT instructions later
                constexpr int RB = MT < 2 ? 1 : ((NT >= 2 || MT < 4) ? 2 : 4);
                bf16x8 ah[NT], am[NT], al[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    ah[nt] = __builtin_bit_cast(bf16x8, au[0][nt]);
                    am[nt] = __builtin_bit_cast(bf16x8, au[1][nt]);
                    al[nt] = __builtin_bit_cast(bf16x8, au[2][nt]);
                }
                auto mf = [&](const bf16x8& bq, const bf16x8& aq, f32x4 c) __attribute__((always_inline)) -> f32x4 {
                    return TR ? dmvs_mfma_bf16(bq, aq, c) : dmvs_mfma_bf16(aq, bq, c);
                };
#pragma unroll
                for (int mt0 = 0; mt0 < MT; mt0 += RB) {
                    bf16x8 bh[RB], bm[RB], bl[RB];
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        const int qp = qbase + ((mt0 + r) * S) * TW;
                        bh[r] = __builtin_bit_cast(bf16x8, q_hi[qp]);
                        bm[r] = __builtin_bit_cast(bf16x8, q_mid[qp]);
                        bl[r] = __builtin_bit_cast(bf16x8, q_lo[qp]);
                    }
#define DMVS_SPLIT_PRODUCT(BQ, AQ, ACC) \
                    _Pragma("unroll") for (int r = 0; r < RB; ++r) \
                        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) ACC[mt0 + r][nt] = mf(BQ[r], AQ[nt], ACC[mt0 + r][nt]);
                    DMVS_SPLIT_PRODUCT(bl, ah, acc)      // smallest partial products first
                    DMVS_SPLIT_PRODUCT(bh, al, acc)
                    DMVS_SPLIT_PRODUCT(bm, am, acc)
                    DMVS_SPLIT_PRODUCT(bm, ah, acc)
                    DMVS_SPLIT_PRODUCT(bh, am, acc)
                    DMVS_SPLIT_PRODUCT(bh, ah, acc)
#undef DMVS_SPLIT_PRODUCT
                }
            };
            const bool more = c0 + CK < cin;
            if constexpr (kPrefetchA) {
                // two register sets, used alternately WITHOUT copies (a rotating copy at the end of the loop body made hipcc wait for the
                // prefetch at the top of the next trip: the L2 latency of the weights was exposed once per tap group -- 17 k cycles per
                // chunk on the 32 -> 32 layer where the matrix work is 2.3 k); invariant: a_cur holds (chunk, 0) when a chunk starts
                auto groups = [&]() __attribute__((always_inline)) {
                    int g = 0;
                    for (; g + 1 < Cfg::NG; g += 2) {
                        load_a(a_next, chunk, g + 1);
                        tap_group(g, a_cur);
                        if (g + 2 < Cfg::NG) load_a(a_cur, chunk, g + 2);
                        else if (more) load_a(a_cur, chunk + 1, 0);
                        tap_group(g + 1, a_next);
                    }
                    if (g < Cfg::NG) {      // odd number of groups: the last one's weights are in a_cur
                        if (more) load_a(a_next, chunk + 1, 0);
                        tap_group(g, a_cur);
                        if (more) {
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) a_cur[pl][nt] = a_next[pl][nt];
                        }
                    }
                };
                groups();
            } else {
#pragma unroll 1
                for (int g = 0; g < Cfg::NG; ++g) {
                    if (g > 0) load_a(a_cur, chunk, g);
                    tap_group(g, a_cur);
                }
                if (more) load_a(a_cur, chunk + 1, 0);
            }
        } else if constexpr (AR == DMVS_ARITH_BF16) {
            static_assert(CK == 8, "the bf16 form maps the 8 channels of an LDS chunk onto the 8 k-slots of a lane");
            constexpr int NG = (T + 3) / 4;          // tap groups: K = 32 = 4 taps x 8 channels
#pragma unroll 1
            for (int g = 0; g < NG; ++g) {
                const int t = 4 * g + kq;
                const bool tv = t < T;
                const int tc = tv ? t : T - 1;
                const int ky = tc / KW, kx = tc - ky * KW;
                const float* wp = s_w + tc * NW + m;
                const float* ip = s_in + ((wy * MT * S) + ky) * TWL + SLACK + (wx * 16 + m) * S + kx;
                bf16x8 av[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float a[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) a[j] = tv ? wp[j * WPAD + nt * 16] : 0.0f;      // tap beyond the kernel: zero weights
                    av[nt] = dmvs_pack_bf16x8(a);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float bb[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) bb[j] = ip[j * PLANE + (mt * S) * TWL];
                    const bf16x8 bv = dmvs_pack_bf16x8(bb);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = TR ? dmvs_mfma_bf16(bv, av[nt], acc[mt][nt]) : dmvs_mfma_bf16(av[nt], bv, acc[mt][nt]);
                }
            }
        } else {
        const int live_c = cin - c0 < CK ? cin - c0 : CK;
        // all-zero 4-channel groups of the last chunk are skipped.  (A 4-channel chunk has one group: said at compile time for the
        // tall-tile form, whose accumulators hipcc otherwise moves VGPR -> AGPR -> VGPR around the one-trip loop, 64 moves per chunk)
        const int nc4 = (CK == 4 && MT >= 8) ? 1 : (live_c + 3) >> 2;
#pragma unroll 1
        for (int c4 = 0; c4 < nc4; ++c4) {
            const int ci = c4 * 4 + kq;
            const float* wp = s_w + ci * WPAD + m;
            const float* ip = s_in + ci * PLANE + (wy * MT * S) * TWL + SLACK + (wx * 16 + m) * S;
            // All KH rows of taps of a 3x3 / 1x1 / 5x1 layer in one loop trip (36-72 MFMAs between two branches instead of 12-24): a loop
            // of 4 MFMAs per trip runs the matrix pipe at 0.81 of the rate of 16 per trip (tools/calib/issue_probe.hip).  Measured per
            // layer (profiles/r3_conv_ky_unroll_ab.txt): -1...-9 % on the >= 32-channel layers, 16 -> 16 unchanged; the 5x5 / 7x7 layers
            // already have 10-40 per trip.  Same order of operations.
            // Round 4 (profiles/r4_conv_tall_s2_ky_ab.jsonl, both removed again): all five rows of the one-n-tile 5x5 layers in one trip (50
            // MFMAs instead of 10) -1.4 % on 8 -> 16 stride 2, nothing elsewhere; the stride-2 B operand read as 8-byte pairs (ds_read2_b64:
            // a lane's taps kx, kx + 1 -- the 4-byte reads of lanes 2 floats apart meet two by two in the banks) +-1 %: LDS read
            // bandwidth is not what separates the stride-2 layers from their stride-1 peers.
            constexpr int kKyUnroll = (KH * KW <= 9) ? KH : 1;
#pragma unroll kKyUnroll
            for (int ky = 0; ky < KH; ++ky) {
#pragma unroll
                for (int kx = 0; kx < KW; ++kx) {
                    float av[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) av[nt] = wp[(ky * KW + kx) * NW + nt * 16];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float bv = ip[(mt * S + ky) * TWL + kx];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = TR ? __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av[nt], acc[mt][nt], 0, 0, 0)       // D[pixel][cout]
                                             : __builtin_amdgcn_mfma_f32_16x16x4f32(av[nt], bv, acc[mt][nt], 0, 0, 0);     // D[cout][pixel]
                    }
                }
            }
        }
        }
    }

    // ---- epilogue: this lane holds couts nbase + nt*16 + 4*kq + r of pixels (oy0 + MT*wave + mt, ox0 + m)
    const int ox = ox0 + wx * 16 + m;
    const int oplane = d.Hout * d.Wout;
    const bool rup = !kLean && d.res_mode == DMVS_IN_UPSAMPLE2;
    const bool do_gn = !WALK && d.gn_stats, do_gru = !kLean && d.gru_z;
    const int act = kLean ? (d.act == DMVS_ACT_RELU ? DMVS_ACT_RELU : DMVS_ACT_NONE) : d.act;
    const int rW = rup ? (d.Wout >> 1) : d.Wout, rH = rup ? (d.Hout >> 1) : d.Hout;
    // per-batch-item bases (wave-uniform, 64-bit) + 32-bit element offsets `channel * plane + pixel` (one full-rate
    // v_mad_u32_u24 per value; the 64-bit multiply-adds this replaces are quarter rate and were ~30 % of the VALU time
    // of the 16-channel layers).  The entry point rejects planes >= 2^24 pixels and tensors >= 2^31 elements per item.
    const int rplane = rH * rW;
    float* const outb = d.out_layout == DMVS_LAYOUT_NCHW ? d.out + ((size_t)b * d.out_cstride + d.out_coffset) * oplane
                                                         : d.out + (size_t)b * oplane * d.out_cstride + d.out_coffset;
    const float* const resb = d.residual ? d.residual + (size_t)b * d.cout * rplane : nullptr;
    const float* const gzb = do_gru ? d.gru_z + (size_t)b * (d.gate_cstride ? d.gate_cstride : d.cout) * oplane : nullptr;
    const float* const ghb = do_gru ? d.gru_h + (size_t)b * d.cout * oplane : nullptr;
    // GroupNorm statistics of the pre-activation output (4 groups), reduced lane -> wave -> workgroup
    float gs[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int gn_cg = do_gn ? d.cout / d.gn_groups : 1;
    if constexpr (TR) {
        // Transposed accumulators (the MFMA was issued with the operands swapped): this lane holds cout nbase + nt*16 + m of
        // the 4 CONSECUTIVE pixels ox0 + 4*kq + r of row oy0 + MT*wave + mt -- one 16-byte NCHW store (and one 16-byte
        // residual / GRU-gate read) per (row, n-tile) instead of four 4-byte ones, one bounds predicate and one offset
        // per four values.  NCHW fp32 outputs only; the values are those of the other form bit for bit.
        const int oxb = ox0 + wx * 16 + 4 * kq;
        const bool vec = kLean || ((d.Wout & 3) == 0 && (((uintptr_t)d.out | (uintptr_t)d.residual | (uintptr_t)d.gru_z | (uintptr_t)d.gru_h | (uintptr_t)d.out_mul) & 15) == 0 &&
                                   ((oplane * d.out_coffset) & 3) == 0);
        float sc[NT], sh[NT];
        int cgs[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            cgs[nt] = nbase + nt * 16 + m;
            const bool okc = cgs[nt] < d.cout;
            sc[nt] = d.scale ? d.scale[okc ? cgs[nt] : 0] : 1.0f;
            sh[nt] = d.shift ? d.shift[okc ? cgs[nt] : 0] : 0.0f;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int oy = oy0 + wy * MT + mt;
            const int opix = oy * d.Wout + oxb;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int cg = cgs[nt];
                const bool okl = oy < d.Hout && cg < d.cout;          // this lane's (row, channel) exists
                bool ok[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) ok[r] = okl && oxb + r < d.Wout;
                const bool fast = vec && ok[0];                        // rows of 16-byte multiples: the four pixels exist together
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = acc[mt][nt][r] * sc[nt] + sh[nt];
                if (do_gn) {
                    const int g = cg / gn_cg;
                    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = ok[r] ? y[r] : 0.0f;
                        s1 += v;
                        s2 += v * v;
                    }
#pragma unroll
                    for (int gi = 0; gi < 4; ++gi) {
                        gs[gi] += g == gi ? s1 : 0.0f;
                        gq[gi] += g == gi ? s2 : 0.0f;
                    }
                }
                const unsigned o0 = okl ? (unsigned)(__mul24(cg, oplane) + opix) : 0u;
                f32x4 res = {0.0f, 0.0f, 0.0f, 0.0f};
                if (d.residual) {
                    if (!rup && fast) {
                        res = *reinterpret_cast<const f32x4*>(resb + o0);
                    } else if (rup && fast) {      // nearest-x2 source: pixels 4*kq..4*kq+3 read source pixels 2*kq, 2*kq+1 (8 bytes)
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        const f32x2 rv = *reinterpret_cast<const f32x2*>(resb + (unsigned)(__mul24(cg, rplane) + (oy >> 1) * rW + (oxb >> 1)));
                        res = f32x4{rv[0], rv[0], rv[1], rv[1]};
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int rpix = rup ? (oy >> 1) * rW + ((oxb + r) >> 1) : opix + r;
                            const float rv = resb[ok[r] ? (unsigned)(__mul24(cg, rplane) + rpix) : 0u];
                            res[r] = ok[r] ? rv : 0.0f;
                        }
                    }
                    if (kLean || !d.res_after_act) y += res;
                }
                if (act == DMVS_ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = fmaxf(y[r], 0.0f);
                } else if (!kLean && act != DMVS_ACT_NONE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = dmvs_act(y[r], act);
                }
                if (!kLean) y *= d.post_scale;
                if (!kLean && d.residual && d.res_after_act) y += res;
                if (do_gru) {
                    f32x4 z, h;
                    if (fast) {
                        z = *reinterpret_cast<const f32x4*>(gzb + o0);
                        h = *reinterpret_cast<const f32x4*>(ghb + o0);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            z[r] = gzb[ok[r] ? o0 + r : 0u];
                            h[r] = ghb[ok[r] ? o0 + r : 0u];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = (1.0f - z[r]) * h[r] + z[r] * y[r];
                }
                if (!kLean && d.out_mul && cg >= d.out_mul_c0 && okl) {      // SepConvGRU: the r half of the merged gate conv leaves as r * h
                    const float* mb = d.out_mul + ((size_t)b * (d.cout - d.out_mul_c0) + (cg - d.out_mul_c0)) * oplane + opix;
                    if (fast) {
                        y *= *reinterpret_cast<const f32x4*>(mb);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (ok[r]) y[r] *= mb[r];      // (mb already holds the pixel offset: a lane whose pixels lie beyond the row must not touch it --
                                                            //  mb[0] of the last row's last lanes is past the end of the tensor; found by the host emulation under ASan, round 6)
                    }
                }
                if (fast) {
                    *reinterpret_cast<f32x4*>(outb + o0) = y;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ok[r]) outb[o0 + r] = y[r];
                }
            }
        }
    } else {
        float sc[NT][4], sh[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cg = nbase + nt * 16 + kq * 4 + r;
                const bool okc = cg < d.cout;
                sc[nt][r] = d.scale ? d.scale[okc ? cg : 0] : 1.0f;
                sh[nt][r] = d.shift ? d.shift[okc ? cg : 0] : 0.0f;
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int oy = oy0 + wy * MT + mt;
            const bool okp = ox < d.Wout && oy < d.Hout;
            const int opix = oy * d.Wout + ox;
            const int rpix = rup ? (oy >> 1) * rW + (ox >> 1) : opix;
            float y[NT][4];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) y[nt][r] = acc[mt][nt][r] * sc[nt][r] + sh[nt][r];
            if (do_gn) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cg = nbase + nt * 16 + kq * 4 + r;
                        const float v = (okp && cg < d.cout) ? y[nt][r] : 0.0f;
                        const int g = cg / gn_cg;
#pragma unroll
                        for (int gi = 0; gi < 4; ++gi) {
                            gs[gi] += g == gi ? v : 0.0f;
                            gq[gi] += g == gi ? v * v : 0.0f;
                        }
                    }
            }
            float res[NT][4];
            if (d.residual) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cg = nbase + nt * 16 + kq * 4 + r;
                        const bool okr = okp && cg < d.cout;
                        const float rv = resb[okr ? (unsigned)(__mul24(cg, rplane) + rpix) : 0u];
                        res[nt][r] = okr ? rv : 0.0f;
                        if (!d.res_after_act) y[nt][r] += res[nt][r];
                    }
            }
            if (act == DMVS_ACT_RELU) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[nt][r] = fmaxf(y[nt][r], 0.0f);
            } else if (!kLean && act != DMVS_ACT_NONE) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[nt][r] = dmvs_act(y[nt][r], act);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) y[nt][r] *= d.post_scale;
            if (d.residual && d.res_after_act) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[nt][r] += res[nt][r];
            }
            if (do_gru) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cg = nbase + nt * 16 + kq * 4 + r;
                        const unsigned gi = (okp && cg < d.cout) ? (unsigned)(__mul24(cg, oplane) + opix) : 0u;
                        const float z = gzb[gi];
                        y[nt][r] = (1.0f - z) * ghb[gi] + z * y[nt][r];
                    }
            }
            if (d.out_mul) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cg = nbase + nt * 16 + kq * 4 + r;
                        if (okp && cg < d.cout && cg >= d.out_mul_c0)
                            y[nt][r] *= d.out_mul[((size_t)b * (d.cout - d.out_mul_c0) + (cg - d.out_mul_c0)) * oplane + opix];
                    }
            }
            if constexpr (OT != DMVS_DTYPE_F32) {      // channel-last, 16-bit elements
                uint16_t* const ob16 = reinterpret_cast<uint16_t*>(d.out) + (size_t)b * oplane * d.out_cstride + d.out_coffset;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cg = nbase + nt * 16 + kq * 4 + r;
                        if (okp && cg < d.cout) ob16[(unsigned)(__mul24(opix, d.out_cstride) + cg)] = dmvs_to_x16<OT>(y[nt][r]);
                    }
            } else if (d.out_layout == DMVS_LAYOUT_NCHW) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cg = nbase + nt * 16 + kq * 4 + r;
                        if (okp && cg < d.cout) outb[(unsigned)(__mul24(cg, oplane) + opix)] = y[nt][r];
                    }
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cg = nbase + nt * 16 + kq * 4 + r;
                        if (okp && cg < d.cout) outb[(unsigned)(__mul24(opix, d.out_cstride) + cg)] = y[nt][r];
                    }
            }
        }
    }
    if (do_gn) {
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                gs[gi] += __shfl_down(gs[gi], o, 64);
                gq[gi] += __shfl_down(gq[gi], o, 64);
            }
        }
        DMVS_LDS_BARRIER();               // the previous tile's readers of gn_scratch are done (LDS-only: a prefetch DMA stays in flight)
        if (lane == 0) {
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                gn_scratch[wave * 8 + gi] = gs[gi];
                gn_scratch[wave * 8 + 4 + gi] = gq[gi];
            }
        }
        DMVS_LDS_BARRIER();
        if (tid < 8) {
            const float tot = gn_scratch[tid] + gn_scratch[8 + tid] + gn_scratch[16 + tid] + gn_scratch[24 + tid];
            const int gi = tid & 3, which = tid >> 2;
            if (tot != 0.0f) dmvs_gn_accumulate(&d.gn_stats[((size_t)b * 4 + gi) * 2], which, (double)tot);
        }
    }
    if constexpr (!WALK) break;
    tile += (int)gridDim.x;
    if (tile >= ntiles) break;
    }      // tiles
}

// Which layers compute in bf16 when the caller asks for DMVS_ARITH_BF16: stride-1 layers with more than one tap, at least 24
// input channels and a planar output.  Measured at B = 96 (profiles/r3_conv_bf16_ab.txt): those run 1.4-1.9x faster (32 -> 32
// 4.88 -> 3.08 ms, 64 -> 64 3.63 -> 1.88); the 16-channel layers do not move at all (5.09 -> 5.06: they are not bound by the
// matrix cores) and the stride-2 / 3-channel layers LOSE with the 8-channel chunks the bf16 form needs (3.80 -> 5.09), so
// those keep the exact-fp32 kernels in either mode.  (dmvs.h documents this as the contract of `arith`.)
static bool conv_bf16_honoured(const dmvs_conv2d_desc& d) {
    return d.arith == DMVS_ARITH_BF16 && d.out_layout == DMVS_LAYOUT_NCHW && d.stride == 1 && d.kh * d.kw > 1 && d.c0 + d.c1 >= 24;
}

// Which layers compute in split-bf16 (fp32-accurate) arithmetic when the caller asks for DMVS_ARITH_SPLIT: multi-tap layers with a planar fp32
// output whose input can be staged in 16-byte pieces (every layer of the reference configurations); 1x1 layers (one tap per K = 32 group would
// waste three quarters of the matrix work), channel-last outputs and the training-only zero-insert form keep the exact-fp32 kernels.
template <int KW>
static bool conv_v16_ok(const dmvs_conv2d_desc& d);
// Measured per layer at B = 96 (profiles/r6_conv_split_ab.txt, ms per step, exact fp32 -> split): 7x7 32 -> 16 4.11 -> 2.43, 3x3 64 -> 64 at
// 576 x 64 x 80 3.31 -> 2.50, 32 -> 32 at 576 x 128 x 160 3.50 -> 2.84 and at 96 x 128 x 160 (x 13) 4.02 -> 3.33, 64 -> 31 2.24 -> 1.71, 5x5 stride 2
// 32 -> 64 2.51 -> 1.98, 24 -> 32 0.95 -> 0.81, 1x5 / 5x1 64 -> 64 0.91 -> 0.76; the one-n-tile 16 -> 16 / 32 -> 16 layers gain 4-6 %.  It loses where
// a tile's conversion pass and the two barriers per chunk outweigh the halved matrix time: few input channels (3 -> 8 0.49 -> 0.74, 6 -> 32 even)
// and the stride-2 layers below 32 input channels, whose tile holds four times the positions per output (8 -> 16 5x5 3.52 -> 4.65, 16 -> 32 5x5
// 2.79 -> 3.10, 3x3 stride 2 all).  Honoured: stride 1 with >= 16 input channels; stride 2 with >= 25 taps and >= 32 input channels.
// DMVS_TUNE_SPLIT_ALL: wherever the form applies (A/B runs, tests).
template <int KW>
static bool conv_split_honoured(const dmvs_conv2d_desc& d) {
    if (d.arith != DMVS_ARITH_SPLIT || !d.weight_split || ((uintptr_t)d.weight_split & 15) || d.kh * d.kw <= 1 || d.in_mode == DMVS_IN_ZEROINSERT2 ||
        (d.out_layout != DMVS_LAYOUT_NCHW && d.out_layout != DMVS_LAYOUT_NHWC) || !conv_v16_ok<KW>(d))
        return false;
    if (d.tune & DMVS_TUNE_SPLIT_ALL) return true;
    const int cin = d.c0 + d.c1;
    // (The rule must not look at the batch size: which arithmetic a layer runs in decides its bits, and the forward is independent of the batch
    // composition -- what makes the scene cache exact.  A "small launches stay fp32" rule was measured -- batch 1 graphed 3.15 -> 3.00 ms per
    // map -- and removed for that reason: tests/test_scene.py caught the scene store's FeatureNet pass differing from the per-sample one.)
    return d.stride == 1 ? cin >= 16 : (cin >= 32 && d.kh * d.kw >= 25);
}

// Waves side by side in a workgroup's tile (template WX) for a 3x3 / 5x5 layer with planar output.  Measured at B = 96
// (profiles/r3_conv_wx_ab.txt; WX = 1 / 2 / 4, ms per step): two n-tiles on planes of >= 128 x 160 pixels gain with 32-wide tiles
// -- 32 -> 32 3x3 4.79 / 4.46 / 5.32 and 4.21 / 3.96 / 4.73, 16 -> 32 5x5 stride 2 3.14 / 2.95 / 4.25, 24 -> 32 1.18 / 1.08 /
// 1.26, 6 -> 32 0.57 / 0.47 / 0.56 -- one n-tile and the small planes lose (16 -> 16 2.44 / 2.56 / 2.89, 32 -> 32 at 64 x 80
// 1.48 / 1.64 / 2.36), 64-wide tiles lose nearly everywhere.  DMVS_TUNE_TILE_WX(1 | 2) forces it for A/B runs.
static int conv_tile_waves_x(const dmvs_conv2d_desc& d, long out_pixels, int nt) {
    const int forced = d.tune & 3;
    if (forced) return forced >= 2 ? 2 : 1;
    return (nt == 2 && out_pixels >= 128L * 160) ? 2 : 1;
}

// The tile-walking kernels are compiled for "lean" layers only (kLean in the kernel): one plain input tensor, exact fp32, ReLU or
// no activation, no gating / GRU blend / GroupNorm statistics / post-scale, an optional same-size residual added before the
// activation, rows of 16-byte multiples on 16-byte aligned tensors.  DMVS_TUNE_NO_WALK: one tile per workgroup everywhere (A/B).
static bool conv_lean_ok(const dmvs_conv2d_desc& d) {
    if ((d.tune & DMVS_TUNE_NO_LEAN) || d.arith == DMVS_ARITH_BF16 || d.in_mode != DMVS_IN_PLAIN || d.mul0 || d.gru_z || d.out_mul) return false;
    if ((d.act != DMVS_ACT_NONE && d.act != DMVS_ACT_RELU) || d.post_scale != 1.0f) return false;
    if (d.residual && (d.res_after_act || d.res_mode != DMVS_IN_PLAIN)) return false;
    if ((d.Wout & 3) || ((((uintptr_t)d.out | (uintptr_t)d.residual) & 15) != 0) || (((long)d.Hout * d.Wout * d.out_coffset) & 3)) return false;
    return true;
}
static bool conv_walk_ok(const dmvs_conv2d_desc& d) {
    if ((d.tune & DMVS_TUNE_NO_WALK) || !conv_lean_ok(d) || d.c1 != 0 || d.gn_stats) return false;
    // Measured at B = 96 (profiles/r3_conv_walk2_ab.txt, ms per step, one tile per workgroup -> walking): the lean layers gain
    // (3 -> 8 at 512x640 0.76 -> 0.64, 8 -> 16 stride 2 0.92 -> 0.85, 16 -> 32 stride 2 0.55 -> 0.48, 16 -> 32 at 64x80 0.28 -> 0.22,
    // 16 -> 32 5x5 stride 2 3.05 -> 2.94) EXCEPT the stride-1 one-n-tile layers with >= 16 input channels (16 -> 16 at 256x320:
    // 5.01 -> 5.49 and 1.41 -> 1.50) -- whatever holds those at ~0.5 is not the per-tile prologue either.
    if (d.stride == 1 && d.cout_pad <= 16 && d.c0 >= 16) return false;
    return true;
}

// 16-byte staging pieces (template V16) need: a PLAIN input (and second concat input), image rows of 16-byte multiples on 16-byte
// aligned tensors, and the "same" padding the LDS row alignment is built for.  DMVS_TUNE_PIECES4: 4-byte pieces everywhere (A/B).
template <int KW>
static bool conv_v16_ok(const dmvs_conv2d_desc& d) {
    if ((d.tune & DMVS_TUNE_PIECES4) || d.in_mode != DMVS_IN_PLAIN || (d.Win & 3) || d.pad_w != (KW - 1) / 2) return false;
    if (((uintptr_t)d.in0 & 15) || (d.c1 > 0 && ((uintptr_t)d.in1 & 15))) return false;
    return true;
}

// 16-bit channel-last outputs (FeatureNet's out1 / out2 / out3 in the reduced-precision configurations): 1x1 and 3x3 stride 1
template <int KH, int KW, int MT, int OT>
int launch_conv2d_x16(const dmvs_conv2d_desc& d, hipStream_t st, int nt, int ngroups) {
    const int tiles_x = (d.Wout + 15) / 16, tiles_y = (d.Hout + 4 * MT - 1) / (4 * MT);
    dim3 grid((unsigned)(tiles_x * tiles_y * d.B), (unsigned)ngroups), block(DMVS_BLOCK);
    switch (nt) {
        case 1: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, 1, 1, MT, false, OT>), grid, block, 0, st, d, tiles_x, tiles_y); break;
        case 2: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, 1, 2, MT, false, OT>), grid, block, 0, st, d, tiles_x, tiles_y); break;
        case 3: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, 1, 3, MT, false, OT>), grid, block, 0, st, d, tiles_x, tiles_y); break;
        default: return DMVS_EINVAL;
    }
    return dmvs_launch_status();
}

template <int KH, int KW, int S, int MT, bool ZI>
int launch_conv2d_mt(const dmvs_conv2d_desc& d, hipStream_t st, int nt, int ngroups) {
    if (d.out_layout == DMVS_LAYOUT_NHWC_BF16 || d.out_layout == DMVS_LAYOUT_NHWC_F16) {
        if constexpr (!ZI && S == 1 && ((KH == 1 && KW == 1) || (KH == 3 && KW == 3)) && MT == 2) {
            return d.out_layout == DMVS_LAYOUT_NHWC_BF16 ? launch_conv2d_x16<KH, KW, MT, DMVS_DTYPE_BF16>(d, st, nt, ngroups)
                                                         : launch_conv2d_x16<KH, KW, MT, DMVS_DTYPE_F16>(d, st, nt, ngroups);
        }
        return DMVS_EINVAL;
    }
    const int tiles_x = (d.Wout + 15) / 16, tiles_y = (d.Hout + 4 * MT - 1) / (4 * MT);
    dim3 grid((unsigned)(tiles_x * tiles_y * d.B), (unsigned)ngroups), block(DMVS_BLOCK);
    const bool v16 = !ZI && conv_v16_ok<KW>(d);
    if (d.out_layout == DMVS_LAYOUT_NCHW) {      // transposed accumulators: 16-byte NCHW stores
        if constexpr (!ZI && KH * KW > 1) {      // split-bf16 arithmetic (fp32-accurate): 16-byte staging pieces, lean or generic epilogue
            if (conv_split_honoured<KW>(d)) {
                const bool lean = conv_lean_ok(d);
#define DMVS_SP(NTV) do { \
                    using SCfg = ConvCfg<KH, KW, S, NTV, MT, DMVS_ARITH_SPLIT, 1, true>; \
                    if constexpr (SCfg::LDS_FLOATS * 4 <= 150 * 1024) { \
                        if (lean) hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, NTV, MT, false, DMVS_DTYPE_F32, true, false, DMVS_ARITH_SPLIT, 1, true, true>), grid, block, 0, st, d, tiles_x, tiles_y); \
                        else hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, NTV, MT, false, DMVS_DTYPE_F32, true, false, DMVS_ARITH_SPLIT, 1, true, false>), grid, block, 0, st, d, tiles_x, tiles_y); \
                        return dmvs_launch_status(); \
                    } } while (0)
                switch (nt) {      // (a shape whose buffers exceed the LDS falls through to the fp32 kernel; launch_conv2d hands the stride-1 layers at most two n-tiles)
                    case 1: DMVS_SP(1); break;
                    case 2: DMVS_SP(2); break;
                    case 3: if constexpr (S == 2) DMVS_SP(3); break;
                    default: if constexpr (S == 2) DMVS_SP(4); break;
                }
#undef DMVS_SP
            }
        }
        if constexpr (!ZI && MT == 2 && KH * KW > 1 && S == 1) {      // bf16 matrix arithmetic: one tile shape (16 x 8), NCHW fp32 outputs
            if (conv_bf16_honoured(d)) {
#define DMVS_BF(NTV) do { \
                    using BCfg = ConvCfg<KH, KW, S, NTV, MT, DMVS_ARITH_BF16>; \
                    if constexpr ((2 * BCfg::BUF + 32) * 4 <= 150 * 1024) { \
                        hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, NTV, MT, false, DMVS_DTYPE_F32, true, false, DMVS_ARITH_BF16>), grid, block, 0, st, d, tiles_x, tiles_y); \
                        return dmvs_launch_status(); \
                    } } while (0)
                switch (nt) {      // (a shape whose 8-channel double buffer exceeds the LDS falls through to the fp32 kernel)
                    case 1: DMVS_BF(1); break;
                    case 2: DMVS_BF(2); break;
                    case 3: DMVS_BF(3); break;
                    default: DMVS_BF(4); break;
                }
#undef DMVS_BF
            }
        }
        if constexpr (!ZI && (KH * KW == 9 || KH * KW == 25)) {
            // the plain 3x3 / 5x5 layers with one or two n-tiles: 32-pixel-wide tiles where measured better (conv_tile_waves_x) and,
            // when the layer is "lean" (conv_walk_ok), resident tile-walking workgroups
            const int wxv = nt <= 2 ? conv_tile_waves_x(d, (long)d.Hout * d.Wout, nt) : 1;
            const bool walk = nt <= 2 && conv_walk_ok(d);
#define DMVS_TILED(NTV, WXV, WALKV) do { if (v16) DMVS_TILED_(NTV, WXV, WALKV, true); else DMVS_TILED_(NTV, WXV, WALKV, false); } while (0)
#define DMVS_TILED_(NTV, WXV, WALKV, V16V) do { \
                    const int tx_ = (d.Wout + 16 * WXV - 1) / (16 * WXV), rows_ = (4 / WXV) * MT, ty_ = (d.Hout + rows_ - 1) / rows_; \
                    auto kfn = conv2d_mfma_kernel<KH, KW, S, NTV, MT, false, DMVS_DTYPE_F32, true, WALKV, DMVS_ARITH_F32, WXV, V16V>; \
                    long gx = (long)tx_ * ty_ * d.B; \
                    if (WALKV) { \
                        static const int resident = dmvs_resident_workgroups(reinterpret_cast<const void*>(kfn)); \
                        const long per_group = resident / ngroups > 0 ? resident / ngroups : 1; \
                        if (gx > per_group) gx = per_group; \
                    } \
                    hipLaunchKernelGGL(kfn, dim3((unsigned)gx, (unsigned)ngroups), block, 0, st, d, tx_, ty_); \
                    return dmvs_launch_status(); } while (0)
            if (walk) {
                if (nt == 1) DMVS_TILED(1, 1, true);
                if (wxv == 2) DMVS_TILED(2, 2, true);
                DMVS_TILED(2, 1, true);
            } else if (wxv == 2) {
                if (nt == 1) DMVS_TILED(1, 2, false);
                DMVS_TILED(2, 2, false);
            }
#undef DMVS_TILED
#undef DMVS_TILED_
        }
        if constexpr (!ZI) {
            if (conv_lean_ok(d)) {      // plain layer, one tile per workgroup: the lean specialisation (template LEAN)
#define DMVS_LEAN(NTV) do { if (v16) hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, NTV, MT, false, DMVS_DTYPE_F32, true, false, DMVS_ARITH_F32, 1, true, true>), grid, block, 0, st, d, tiles_x, tiles_y); \
                            else hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, NTV, MT, false, DMVS_DTYPE_F32, true, false, DMVS_ARITH_F32, 1, false, true>), grid, block, 0, st, d, tiles_x, tiles_y); } while (0)
                switch (nt) {
                    case 1: DMVS_LEAN(1); break;
                    case 2: DMVS_LEAN(2); break;
                    case 3: DMVS_LEAN(3); break;
                    default: DMVS_LEAN(4); break;
                }
#undef DMVS_LEAN
                return dmvs_launch_status();
            }
            if (v16) {
                switch (nt) {
                    case 1: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 1, MT, false, DMVS_DTYPE_F32, true, false, DMVS_ARITH_F32, 1, true>), grid, block, 0, st, d, tiles_x, tiles_y); break;
                    case 2: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 2, MT, false, DMVS_DTYPE_F32, true, false, DMVS_ARITH_F32, 1, true>), grid, block, 0, st, d, tiles_x, tiles_y); break;
                    case 3: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 3, MT, false, DMVS_DTYPE_F32, true, false, DMVS_ARITH_F32, 1, true>), grid, block, 0, st, d, tiles_x, tiles_y); break;
                    default: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 4, MT, false, DMVS_DTYPE_F32, true, false, DMVS_ARITH_F32, 1, true>), grid, block, 0, st, d, tiles_x, tiles_y); break;
                }
                return dmvs_launch_status();
            }
        }
        switch (nt) {
            case 1: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 1, MT, ZI, DMVS_DTYPE_F32, true>), grid, block, 0, st, d, tiles_x, tiles_y); break;
            case 2: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 2, MT, ZI, DMVS_DTYPE_F32, true>), grid, block, 0, st, d, tiles_x, tiles_y); break;
            case 3: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 3, MT, ZI, DMVS_DTYPE_F32, true>), grid, block, 0, st, d, tiles_x, tiles_y); break;
            default: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 4, MT, ZI, DMVS_DTYPE_F32, true>), grid, block, 0, st, d, tiles_x, tiles_y); break;
        }
        return dmvs_launch_status();
    }
    if constexpr (!ZI && KH * KW > 1) {      // fp32 channel-last outputs in split-bf16 arithmetic (the D[cout][pixel] form: 16-byte channel runs per pixel)
        if (d.out_layout == DMVS_LAYOUT_NHWC && conv_split_honoured<KW>(d)) {
#define DMVS_SPC(NTV) do { \
                using SCfg = ConvCfg<KH, KW, S, NTV, MT, DMVS_ARITH_SPLIT, 1, true>; \
                if constexpr (SCfg::LDS_FLOATS * 4 <= 150 * 1024) { \
                    hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, NTV, MT, false, DMVS_DTYPE_F32, false, false, DMVS_ARITH_SPLIT, 1, true, false>), grid, block, 0, st, d, tiles_x, tiles_y); \
                    return dmvs_launch_status(); \
                } } while (0)
            switch (nt) {
                case 1: DMVS_SPC(1); break;
                case 2: DMVS_SPC(2); break;
                default: break;
            }
#undef DMVS_SPC
        }
    }
    if constexpr (!ZI) {      // fp32 channel-last outputs (FeatureNet's out1 / out2 / out3)
        if (v16) {
            switch (nt) {
                case 1: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 1, MT, false, DMVS_DTYPE_F32, false, false, DMVS_ARITH_F32, 1, true>), grid, block, 0, st, d, tiles_x, tiles_y); break;
                case 2: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 2, MT, false, DMVS_DTYPE_F32, false, false, DMVS_ARITH_F32, 1, true>), grid, block, 0, st, d, tiles_x, tiles_y); break;
                case 3: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 3, MT, false, DMVS_DTYPE_F32, false, false, DMVS_ARITH_F32, 1, true>), grid, block, 0, st, d, tiles_x, tiles_y); break;
                default: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 4, MT, false, DMVS_DTYPE_F32, false, false, DMVS_ARITH_F32, 1, true>), grid, block, 0, st, d, tiles_x, tiles_y); break;
            }
            return dmvs_launch_status();
        }
    }
    switch (nt) {
        case 1: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 1, MT, ZI>), grid, block, 0, st, d, tiles_x, tiles_y); break;
        case 2: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 2, MT, ZI>), grid, block, 0, st, d, tiles_x, tiles_y); break;
        case 3: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 3, MT, ZI>), grid, block, 0, st, d, tiles_x, tiles_y); break;
        default: hipLaunchKernelGGL((conv2d_mfma_kernel<KH, KW, S, 4, MT, ZI>), grid, block, 0, st, d, tiles_x, tiles_y); break;
    }
    return dmvs_launch_status();
}

// 16 x 32-pixel tiles (MT = 8) for the plain 3x3 layers with one n-tile -- the 16 -> 16 layers that sat at 0.47-0.58 of the matrix peak.
// Per MFMA a tall tile has half the tile decode / staging map / weight map, its 34 x 6 halo pieces fill 204 of the 256 lanes of ONE
// DMA wave-instruction set per channel (the 16 x 16 tile's 108 pieces leave 148 lanes idle) and its 10 halo rows feed 8 output rows (6
// feed 4).  The chunk drops to 4 channels (ConvCfg::CK: an 8-channel double buffer would be 61 KB) with the same 72 MFMAs per wave between
// two barriers, 31 KB of LDS.  Measured at B = 96 (profiles/r4_conv_tall_s2_ky_ab.jsonl, bit-identical): 16 -> 16 at 576 x 256 x 320
// 2479 -> 2350 us, at 96 x 256 x 320 420 -> 384, at 96 x 128 x 160 112 -> 101, at 96 x 64 x 80 37 -> 32; 32 -> 16 184 -> 177; 64 -> 16
// unchanged (338 -> 339); the step 75.8 -> 75.0 ms (profiles/r4_tall_tiles_step_ab.json).  Default for <= 32 input channels when the
// launch still has >= 3 tall tiles per CU (small batches keep the 16 x 16 / 16 x 4 tiles: more workgroups, shorter chunk chains).
// Two n-tiles (96 VGPRs, 36 KB): per layer on random data 32 -> 32 at 96 x 128 x 160 387 -> 348 us, 64 -> 32 at 576 -5 % (profiles/
// r4_conv_tall2_ab.jsonl); in the model's step, forced everywhere, the rows that gain are the >= 32-input-channel layers on the
// 128 x 160 planes (64 -> 31 2.51 -> 2.24 ms, 32 -> 32 at 576 images 3.61 -> 3.49), the 6 / 16 / 24-channel and 64 x 80 ones lose a
// little (profiles/r4_tall2_step_ab.json: 54.06 -> 53.90 ms of convolutions, the step unchanged within the box noise) -- so: >= 32
// input channels on planes of >= 128 x 160 pixels.  DMVS_TUNE_TALL(1) = never, (2) = wherever the form applies.
// (16 x 64 tiles, round 5 on the MI355X: 2562 vs 2385 us on 16 -> 16 at 576 x 256 x 320, 118 vs 98 us at 96 x 128 x 160 -- two workgroups
// per CU lose more than the halo saves; removed, profiles/r5_optins.jsonl.)
template <int KH, int KW, int S>
static bool conv_tall_ok(const dmvs_conv2d_desc& d, int nt) {
    if constexpr (KH == 3 && KW == 3 && S == 1) {
        const int mode = (d.tune >> 10) & 3;
        if (mode == 1 || nt > 2) return false;
        if (d.out_layout != DMVS_LAYOUT_NCHW || !conv_lean_ok(d) || !conv_v16_ok<3>(d) || d.Hout < 32) return false;
        if (mode == 2) return true;
        const long tall_tiles = (long)((d.Wout + 15) / 16) * ((d.Hout + 31) / 32) * d.B;
        if (nt == 2) return d.c0 + d.c1 >= 32 && (long)d.Hout * d.Wout >= 128L * 160 && tall_tiles >= 1024;
        return d.c0 + d.c1 <= 32 && tall_tiles >= 768;
    }
    return false;
}

template <int KH, int KW, int S>
int launch_conv2d(const dmvs_conv2d_desc& d_in, hipStream_t st) {
    // a layer the split form is not honoured for is an exact-fp32 layer in every respect (lean / tall / walking kernels included)
    dmvs_conv2d_desc d = d_in;
    if (d.arith == DMVS_ARITH_SPLIT && !conv_split_honoured<KW>(d)) d.arith = DMVS_ARITH_F32;
    const int ntiles = (d.cout_pad + 15) / 16;
    // output channels per workgroup: up to 4 MFMA n-tiles share one staged input tile
    const int nt = ntiles <= 4 ? ntiles : (ntiles % 3 == 0 ? 3 : 4);
    const int ngroups = (ntiles + nt - 1) / nt;
    if constexpr (KH * KW > 1) {
        if (conv_split_honoured<KW>(d)) {      // split-bf16 arithmetic: 16 x 16 tiles for the light families (the weight split is per wave and tap group), 16 x 8 otherwise
            constexpr bool heavy_ = (S == 2) || (KH * KW >= 25);
            const int force_mt_ = (d.tune >> 4) & 7;
            // at most two n-tiles per workgroup: the tap group's weights and the next group's (2 x 3 planes x NT x 4 registers) live in registers
            // (stride 2: a tile's conversion pass covers four times the positions per output, so splitting the output channels over more
            // workgroups costs more than the exposed weight latency -- 32 -> 64 5x5 stride 2: 1.94 ms with four n-tiles, 2.78 with two)
            const int nts = S == 2 ? nt : (nt < 2 ? nt : 2), ngs = (ntiles + nts - 1) / nts;
            if constexpr (!heavy_) {
                const long wg16_ = (long)((d.Wout + 15) / 16) * ((d.Hout + 15) / 16) * d.B * ngs;
                if (force_mt_ == 4 || (force_mt_ == 0 && wg16_ >= 1024)) return launch_conv2d_mt<KH, KW, S, 4, false>(d, st, nts, ngs);
            }
            return launch_conv2d_mt<KH, KW, S, 2, false>(d, st, nts, ngs);
        }
    }
    if constexpr (KH == 3 && KW == 3 && S == 1) {
        if (conv_tall_ok<KH, KW, S>(d, nt)) {
            const int tiles_x = (d.Wout + 15) / 16, tiles_y = (d.Hout + 31) / 32;
            const dim3 grid((unsigned)(tiles_x * tiles_y * d.B), 1u);
            if (nt == 1)
                hipLaunchKernelGGL((conv2d_mfma_kernel<3, 3, 1, 1, 8, false, DMVS_DTYPE_F32, true, false, DMVS_ARITH_F32, 1, true, true>), grid,
                                   dim3(DMVS_BLOCK), 0, st, d, tiles_x, tiles_y);
            else
                hipLaunchKernelGGL((conv2d_mfma_kernel<3, 3, 1, 2, 8, false, DMVS_DTYPE_F32, true, false, DMVS_ARITH_F32, 1, true, true>), grid,
                                   dim3(DMVS_BLOCK), 0, st, d, tiles_x, tiles_y);
            return dmvs_launch_status();
        }
    }
    if (d.in_mode == DMVS_IN_ZEROINSERT2) {        // training: input gradient of the 3x3 / 5x5 stride-2 layers
        if constexpr (S == 1 && ((KH == 3 && KW == 3) || (KH == 5 && KW == 5)))
            return launch_conv2d_mt<KH, KW, S, 2, true>(d, st, nt, ngroups);
        return DMVS_EINVAL;
    }
    // pixel tile = 16 x (4*MT).  Tall tiles amortise the halo and the weight slab; small images take
    // 16x4 tiles so that the 256 CUs still see a few workgroups each; stride-2 / many-tap / wide-N
    // shapes stop at MT=2 to keep the staging registers + accumulators inside the VGPR file.
    // Thresholds from tools/conv_bench.py with the tile height forced (B = 16, us for MT = 1 / 2 / 4):
    //   64->32 at 64x80 (320 16x16 tiles) 44.8 / 47.4 / 54.3;  32->32 at 64x80 28.1 / 28.6 / 32.0;
    //   32->32 at 128x160 (1280 tiles: 1.25 rounds of the ~1024 resident 16x16 workgroups) 86.9 / 80.5 / 85.3;
    //   24->32 at 128x160 70.1 / 65.0 / 67.7;  16->16 at 128x160 (one n-tile) 38.0 / 30.7 / 29.3.
    constexpr bool heavy = (S == 2) || (KH * KW >= 25);
    if (d.out_layout == DMVS_LAYOUT_NHWC_BF16 || d.out_layout == DMVS_LAYOUT_NHWC_F16)      // one tile shape (16x8) for the 16-bit outputs
        return launch_conv2d_mt<KH, KW, S, 2, false>(d, st, nt, ngroups);
    if (S == 1 && KH * KW > 1 && conv_bf16_honoured(d))      // and for the bf16 matrix arithmetic
        return launch_conv2d_mt<KH, KW, S, 2, false>(d, st, nt, ngroups);
    const long wg16 = (long)((d.Wout + 15) / 16) * ((d.Hout + 15) / 16) * d.B * ngroups;
    const int force_mt = (d.tune >> 4) & 7;      // DMVS_TUNE_TILE_MT: A/B runs force the tile height
    // (16 x 16-pixel tiles for the stride-2 / many-tap families, timed per layer in round 5: within +-2 % of the 16 x 8 tiles on the 5x5
    // stride-2 layers, -5 % on the 7x7 and +10 % on the 1x5 / 5x1 ones -- not worth their LDS; the instantiations are gone, profiles/r5_optins.jsonl)
    if constexpr (!heavy) {
        if (force_mt == 4) return launch_conv2d_mt<KH, KW, S, 4, false>(d, st, nt, ngroups);
    }
    if (force_mt == 2) return launch_conv2d_mt<KH, KW, S, 2, false>(d, st, nt, ngroups);
    if (force_mt == 1) return launch_conv2d_mt<KH, KW, S, 1, false>(d, st, nt, ngroups);
    if (wg16 * 2 < 1024) return launch_conv2d_mt<KH, KW, S, 1, false>(d, st, nt, ngroups);
    if (heavy || nt == 4 || (nt >= 2 && wg16 < 2048)) return launch_conv2d_mt<KH, KW, S, 2, false>(d, st, nt, ngroups);
    if constexpr (!heavy) return launch_conv2d_mt<KH, KW, S, 4, false>(d, st, nt, ngroups);
    return DMVS_EINVAL;
}


}  // namespace

namespace dmvs_detail {
// one per (KH, KW, stride) family; defined in conv2d_k33.hip / conv2d_k55.hip / conv2d_k77.hip / conv2d_k15.hip
int launch_conv2d_111(const dmvs_conv2d_desc& d, hipStream_t st);
int launch_conv2d_331(const dmvs_conv2d_desc& d, hipStream_t st);
int launch_conv2d_332(const dmvs_conv2d_desc& d, hipStream_t st);
int launch_conv2d_552(const dmvs_conv2d_desc& d, hipStream_t st);
int launch_conv2d_551(const dmvs_conv2d_desc& d, hipStream_t st);
int launch_conv2d_771(const dmvs_conv2d_desc& d, hipStream_t st);
int launch_conv2d_151(const dmvs_conv2d_desc& d, hipStream_t st);
int launch_conv2d_511(const dmvs_conv2d_desc& d, hipStream_t st);
}  // namespace dmvs_detail
