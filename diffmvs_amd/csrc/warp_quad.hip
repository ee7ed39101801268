// Homography warp + group-wise correlation, "quad per pixel" kernels: GetCost (reference models/module.py:583-667) and the
// stage-1 plane sweep (module.py:514-531, differentiable_warping :181-218) for ANY geometry -- no tile windows, no pre-pass,
// no worklists, no LDS, no barriers.
//
// Mapping.  A reference pixel is owned by the 4 adjacent lanes of a quad; lane q owns correlation group q.  Features are read
// in the GROUP-INTERLEAVED channel-last layout ("NHWC-g4", see dmvs.h): a texel is C/16 units of 64 bytes, unit j =
// [4 channels of group 0 | group 1 | group 2 | group 3], so the quad fetches one unit with ONE fully coalesced 64-byte request
// (lane q: bytes 16q..16q+15) and every lane receives channels of its own group only -- the group dot needs no cross-lane
// reduction at all.
//
// Per (pixel, view) the work is organised by distinct source TEXEL, not by hypothesis:
//   1. the NH hypotheses of the pixel are projected ONCE, split over the quad's lanes (lane q: hypotheses q and q+4);
//   2. the texels their 2x2 footprints touch are collected in a per-pixel bitmask over an 8x8 texel grid anchored at the
//      minimum footprint corner (quad-wide min / or through DPP quad_perm, no LDS); consecutive hypotheses walk the epipolar
//      line in sub-texel steps, so 6 hypotheses touch ~4-7 distinct texels instead of 24 taps;
//   3. each set bit = one texel: fetched once (C/16 loads per lane), dotted with the lane's reference channels once
//      (C/4 FMAs), then scattered to ALL hypotheses with the bilinear "hat" weight  max(0,1-|u-x|) * max(0,1-|v-y|)
//      (identical to the 2x2 tap weights, and exactly zero for hypotheses the texel does not belong to): each lane
//      evaluates the weights of its own two hypotheses, the others arrive by DPP broadcast fused into the FMA.
//   The texel loop is a plain per-lane `while (mask)` (exec-masked, so a pixel with fewer texels issues no requests) unrolled
//   by two so that two texels' loads are in flight per trip.
// A pixel whose hypotheses spread over more than 8 texels along an axis (very low confidence next to a wide baseline) falls
// back to one chunk per hypothesis: same code, NH times.
//
// Semantics kept from the reference: per-tap zero padding with align_corners=True pixel coordinates, NO behind-camera
// mask, z == 0 -> z + 1e-8, non-finite coordinates sample 0.
#include <utility>

#include "dmvs_common.h"

// DMVS_GC_EXP (diagnostic builds only, tools/build_variant.py; 0 = the product): which resource bounds the quad kernels?
//   1: every texel address replaced by the pixel's own position in the view (coalesced, L1-friendly) -- same instructions, no
//      scattered memory traffic: the VALU / issue floor;   2: hat weights + scatter removed (loads + dots stay): the memory
//      floor;   3: only the first 64-byte unit of a texel is loaded: half the requests, the same lines;
//   4: the CEILING PROBE (round 5): projection, texel masks, bit scans, addresses and every load exactly as in the product -- the same
//      line-request stream from the same quads -- but the loaded registers are only waited for (no dot, no hat weights, no scatter;
//      the outputs are zeros): the time of this build is what the memory system alone delivers for this address stream.
#ifndef DMVS_GC_EXP
#define DMVS_GC_EXP 0
#endif

#ifndef DMVS_QUAD_PERM      // (the host emulation predefines it)
#define DMVS_QUAD_PERM(v, ctrl) __builtin_amdgcn_mov_dpp((v), (ctrl), 0xf, 0xf, true)
#endif

namespace {

constexpr int QP_XOR1 = 0xB1;      // quad_perm:[1,0,3,2]
constexpr int QP_XOR2 = 0x4E;      // quad_perm:[2,3,0,1]
constexpr int BIG = 0x3fffffff;

template <int CTRL> __device__ __forceinline__ int qperm(int v) { return DMVS_QUAD_PERM(v, CTRL); }
template <int CTRL> __device__ __forceinline__ float qperm(float v) { return __int_as_float(DMVS_QUAD_PERM(__float_as_int(v), CTRL)); }
__device__ __forceinline__ int quad_min(int v) {
    v = min(v, qperm<QP_XOR1>(v));
    return min(v, qperm<QP_XOR2>(v));
}
__device__ __forceinline__ int quad_max(int v) {
    v = max(v, qperm<QP_XOR1>(v));
    return max(v, qperm<QP_XOR2>(v));
}
__device__ __forceinline__ unsigned quad_or(unsigned v) {
    v |= (unsigned)qperm<QP_XOR1>((int)v);
    return v | (unsigned)qperm<QP_XOR2>((int)v);
}

__device__ __forceinline__ unsigned mad_u24(unsigned a, unsigned b, unsigned c) {      // a * b + c, a and b below 2^24
#ifdef DMVS_HOST_EMULATION
    return a * b + c;
#else
    unsigned d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
#endif
}

struct RayQ {   // p(depth) = rot * (x, y, 1) * depth + trans   (reference module.py:199-205)
    float rx, ry, rz, tx, ty, tz;
    __device__ __forceinline__ void init(const float* m, float x, float y) {
        rx = m[0] * x + m[1] * y + m[2];
        ry = m[3] * x + m[4] * y + m[5];
        rz = m[6] * x + m[7] * y + m[8];
        tx = m[9]; ty = m[10]; tz = m[11];
    }
};

// one hypothesis of this lane: source coordinates, top-left texel of its footprint, whether it can touch the image at all
struct HypQ {
    float u, v;
    int x0, y0;
    bool valid;
};

__device__ __forceinline__ void project_uv_q(const RayQ& r, float depth, float& u, float& v) {
    const float px = r.rx * depth + r.tx, py = r.ry * depth + r.ty;
    float pz = r.rz * depth + r.tz;
    if (pz == 0.0f) pz += 1e-8f;
    // one reciprocal (hardware estimate + one Newton step: within an ulp of the IEEE quotient) shared by u and v
    float inv = __builtin_amdgcn_rcpf(pz);
    inv = fmaf(fmaf(-pz, inv, 1.0f), inv, inv);
    u = px * inv;
    v = py * inv;
}

__device__ __forceinline__ HypQ footprint_q(float u, float v, bool exists, int Hs, int Ws) {
    HypQ h;
    h.u = u;
    h.v = v;
    const float fx = floorf(h.u), fy = floorf(h.v);
    // false for NaN / inf; a footprint with both columns (rows) outside the image only has padding taps
    h.valid = exists && fx >= -1.0f && fx <= (float)(Ws - 1) && fy >= -1.0f && fy <= (float)(Hs - 1);
    h.x0 = h.valid ? (int)fx : BIG;
    h.y0 = h.valid ? (int)fy : BIG;
    return h;
}

__device__ __forceinline__ HypQ project_q(const RayQ& r, float depth, bool exists, int Hs, int Ws) {
    float u, v;
    project_uv_q(r, depth, u, v);
    return footprint_q(u, v, exists, Hs, Ws);
}

typedef float f2q __attribute__((vector_size(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// pointer into the workgroup's LDS (address space 3: the loads become ds_read_*; g++ of the host emulation ignores the attribute)
typedef const __attribute__((address_space(3))) char* lds_cptr;
template <typename P, typename T> struct PtrAs { typedef const T* type; };
#ifndef DMVS_HOST_EMULATION
template <typename T> struct PtrAs<lds_cptr, T> { typedef const __attribute__((address_space(3))) T* type; };
#endif
template <typename V, typename P> __device__ __forceinline__ V ldv(P p) {
#ifdef DMVS_HOST_EMULATION
    V v;
    memcpy(&v, (const void*)p, sizeof(V));       // (the host's vector loads want natural alignment)
    return v;
#else
    return *(typename PtrAs<P, V>::type)(p);
#endif
}

// This lane's C/4 channels (= all channels of its correlation group) of one texel, for the three feature element types.
//   fp32   : NHWC-g4, C/16 units of 64 bytes, the lane reads 16 bytes of each (4 channels)
//   16-bit : plain NHWC -- a group's C/4 channels are already contiguous (8 / 16 / 24 bytes per lane), the quad still reads one
//            contiguous run of 2*C bytes; converted to fp32 on arrival, all arithmetic stays fp32
//   fp32 plain (DMVS_DTYPE_F32_PLAIN): plain NHWC fp32 -- the lane's C/4 channels are C bytes contiguous (C/16 loads of 16 bytes
//            at a C-byte lane pitch instead of one 64-byte run per quad and unit); the training graph's features, whose backward
//            kernels (warp_bwd*.hip) read the same tensors in that order
template <int C, int FT> struct Feat {
    static constexpr int E = C / 4;                                          // channels per lane
    static constexpr bool F32 = FT == DMVS_DTYPE_F32 || FT == DMVS_DTYPE_F32_PLAIN;
    static constexpr int ESIZE = F32 ? 4 : 2;
    static constexpr int TEXEL_BYTES = C * ESIZE;
    static constexpr int NW = F32 ? E : E / 2;              // 32-bit words per lane and texel
    uint32_t w[NW];

    static __device__ __forceinline__ unsigned lane_bytes(int q) { return FT == DMVS_DTYPE_F32 ? (unsigned)q * 16u : (unsigned)q * (E * ESIZE); }

    // P = const char* (global memory) or lds_cptr (the workgroup's staged band: ds_read_b128 / _b64)
    template <typename P> __device__ __forceinline__ void load(P p) {
        if constexpr (F32) {
            constexpr int PITCH = FT == DMVS_DTYPE_F32 ? 64 : 16;      // g4: one 16-byte piece per 64-byte unit; plain: consecutive pieces
#pragma unroll
            for (int j = 0; j < C / 16; ++j) {
                const u32x4 v = ldv<u32x4>(p + ((DMVS_GC_EXP == 3) ? 0 : j * PITCH));
                w[4 * j] = v[0]; w[4 * j + 1] = v[1]; w[4 * j + 2] = v[2]; w[4 * j + 3] = v[3];
            }
        } else if constexpr (NW == 6) {
            // 24 bytes per lane at an 8-byte aligned address: 8-byte pieces (a 16-byte LDS read needs a 16-byte aligned address)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const u32x2 v = ldv<u32x2>(p + j * 8);
                w[2 * j] = v[0]; w[2 * j + 1] = v[1];
            }
        } else if constexpr (NW == 4) {
            const u32x4 v = ldv<u32x4>(p);
            w[0] = v[0]; w[1] = v[1]; w[2] = v[2]; w[3] = v[3];
        } else {
            const u32x2 v = ldv<u32x2>(p);
            w[0] = v[0]; w[1] = v[1];
        }
    }
    __device__ __forceinline__ float get(int i) const {      // channel i of the lane's group
        if constexpr (F32) return __uint_as_float(w[i]);
        else if constexpr (FT == DMVS_DTYPE_BF16) return __uint_as_float((i & 1) ? (w[i >> 1] & 0xffff0000u) : (w[i >> 1] << 16));
        else return dmvs_f16_to_f32((uint16_t)((i & 1) ? (w[i >> 1] >> 16) : (w[i >> 1] & 0xffffu)));
    }
};

// group dot with the lane's (pre-scaled, fp32) reference channels: two partial sums in one register pair, so that packed fp32
// FMAs (v_pk_fma_f32) retire two products per issue slot
template <int C, int FT>
__device__ __forceinline__ float dot_texel(const Feat<C, FT>& t, const float (&ref)[C / 4]) {
    f2q a = {0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < C / 4; i += 2) a = f2q{t.get(i), t.get(i + 1)} * f2q{ref[i], ref[i + 1]} + a;
    return a[0] + a[1];
}

// the TPT texels of a trip together, channel pairs outermost: the dependent packed-FMA chains of the texels interleave (a
// v_pk_fma_f32 straight after the one it depends on costs a wait state)
template <int C, int FT, int TPT>
__device__ __forceinline__ void dot_texels(const Feat<C, FT> (&t)[TPT], const float (&ref)[C / 4], float (&dd)[TPT]) {
    f2q a[TPT];
#pragma unroll
    for (int i = 0; i < TPT; ++i) a[i] = f2q{0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < C / 4; j += 2) {
#pragma unroll
        for (int i = 0; i < TPT; ++i) a[i] = f2q{t[i].get(j), t[i].get(j + 1)} * f2q{ref[j], ref[j + 1]} + a[i];
    }
#pragma unroll
    for (int i = 0; i < TPT; ++i) dd[i] = a[i][0] + a[i][1];
}

// the lane's reference channels, scaled by 1 / (channels per group): cor = MEAN over the group (module.py:529-531)
template <int C, int FT>
__device__ __forceinline__ void load_ref(const void* ref_base, long pixel, int q, float (&ref)[C / 4]) {
    Feat<C, FT> r;
    r.load(reinterpret_cast<const char*>(ref_base) + pixel * Feat<C, FT>::TEXEL_BYTES + Feat<C, FT>::lane_bytes(q));
    const float inv_cg = 1.0f / (float)(C / 4);
#pragma unroll
    for (int i = 0; i < C / 4; ++i) ref[i] = r.get(i) * inv_cg;
}

__device__ __forceinline__ float hat(float rel, float pos) {      // bilinear weight of integer position `pos` for coordinate `rel`
    return fminf(fmaxf(1.0f - fabsf(rel - pos), 0.0f), 1.0f);
}

// acc[k] += W0[k] * d0 + W1[k] * d1 for every hypothesis k of the pixel: W.[k] lives in lane k & 3 of the quad as that lane's
// (k >> 2)-th weight and is read through DPP quad_perm inside the FMA itself (v_fmac_f32_dpp: no broadcast moves).  One asm
// block per texel pair; the leading s_nop covers the VALU-write -> DPP-read hazard of the weights computed just before.
#ifdef DMVS_HOST_EMULATION
template <int K, int HPL>
__device__ __forceinline__ float bcast_w(const float (&w)[HPL]) { return qperm<(K & 3) * 0x55>(w[K >> 2]); }
template <int NH, int HPL, int... K>
__device__ __forceinline__ void scatter_seq(float (&acc)[NH], const float (&w0)[HPL], float d0, const float (&w1)[HPL], float d1,
                                            std::integer_sequence<int, K...>) {
    ((acc[K] = fmaf(bcast_w<K, HPL>(w0), d0, acc[K])), ...);
    ((acc[K] = fmaf(bcast_w<K, HPL>(w1), d1, acc[K])), ...);
}
template <int NH, int HPL>
__device__ __forceinline__ void scatter_pair(float (&acc)[NH], const float (&w0)[HPL], float d0, const float (&w1)[HPL], float d1) {
    scatter_seq<NH, HPL>(acc, w0, d0, w1, d1, std::make_integer_sequence<int, NH>{});
}
#else
#define DMVS_QF(A, W, D, L) "v_fmac_f32_dpp %" #A ", %" #W ", %" #D " quad_perm:[" #L "," #L "," #L "," #L "] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
template <int NH, int HPL>
__device__ __forceinline__ void scatter_pair(float (&acc)[NH], const float (&w0)[HPL], float d0, const float (&w1)[HPL], float d1) {
    if constexpr (NH == 4) {
        asm("s_nop 1\n\t"
            DMVS_QF(0, 4, 6, 0) DMVS_QF(1, 4, 6, 1) DMVS_QF(2, 4, 6, 2) DMVS_QF(3, 4, 6, 3)
            DMVS_QF(0, 5, 7, 0) DMVS_QF(1, 5, 7, 1) DMVS_QF(2, 5, 7, 2) DMVS_QF(3, 5, 7, 3)
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
            : "v"(w0[0]), "v"(w1[0]), "v"(d0), "v"(d1));
    } else if constexpr (NH == 6) {
        asm("s_nop 1\n\t"
            DMVS_QF(0, 6, 10, 0) DMVS_QF(1, 6, 10, 1) DMVS_QF(2, 6, 10, 2) DMVS_QF(3, 6, 10, 3) DMVS_QF(4, 7, 10, 0) DMVS_QF(5, 7, 10, 1)
            DMVS_QF(0, 8, 11, 0) DMVS_QF(1, 8, 11, 1) DMVS_QF(2, 8, 11, 2) DMVS_QF(3, 8, 11, 3) DMVS_QF(4, 9, 11, 0) DMVS_QF(5, 9, 11, 1)
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5])
            : "v"(w0[0]), "v"(w0[1]), "v"(w1[0]), "v"(w1[1]), "v"(d0), "v"(d1));
    } else {
        static_assert(NH == 8, "scatter_pair: 4, 6 or 8 hypotheses");
        asm("s_nop 1\n\t"
            DMVS_QF(0, 8, 12, 0) DMVS_QF(1, 8, 12, 1) DMVS_QF(2, 8, 12, 2) DMVS_QF(3, 8, 12, 3)
            DMVS_QF(4, 9, 12, 0) DMVS_QF(5, 9, 12, 1) DMVS_QF(6, 9, 12, 2) DMVS_QF(7, 9, 12, 3)
            DMVS_QF(0, 10, 13, 0) DMVS_QF(1, 10, 13, 1) DMVS_QF(2, 10, 13, 2) DMVS_QF(3, 10, 13, 3)
            DMVS_QF(4, 11, 13, 0) DMVS_QF(5, 11, 13, 1) DMVS_QF(6, 11, 13, 2) DMVS_QF(7, 11, 13, 3)
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
            : "v"(w0[0]), "v"(w0[1]), "v"(w1[0]), "v"(w1[1]), "v"(d0), "v"(d1));
    }
}
#undef DMVS_QF
#endif

// acc[k] += wscale * (bilinear sample of the lane's group dot at hypothesis k), k < NH, for one (pixel, view).
// own[h] = hypothesis q + 4h of the pixel (q = lane & 3).  Texel (x, y) of the view is read at base + origin + (y * pitch + x) *
// TEXEL_BYTES (32-bit wrapping arithmetic): global memory -- base + origin = the view's [Hs,Ws,C] NHWC-g4 image (+ the lane's
// 16q bytes), pitch = Ws -- or a band of the view staged in LDS (base = the band, origin = lane bytes - the band's corner).
template <int C, int FT, int NH, int TPT, typename P>
__device__ __forceinline__ void quad_accumulate(P base, unsigned view_off, int pitch, const HypQ (&own)[(NH + 3) / 4], int Hs, int Ws,
                                                const float (&ref)[C / 4], float wscale, float (&acc)[NH]) {
    constexpr int HPL = (NH + 3) / 4, TB = Feat<C, FT>::TEXEL_BYTES;
    const int q = threadIdx.x & 3;
    // do all hypotheses of the pixel fit one 8x8 texel grid anchored at the minimum footprint corner?
    int xlo = BIG, ylo = BIG;
#pragma unroll
    for (int h = 0; h < HPL; ++h) {
        xlo = min(xlo, own[h].x0);       // invalid hypotheses carry BIG
        ylo = min(ylo, own[h].y0);
    }
    xlo = quad_min(xlo);
    ylo = quad_min(ylo);
    if (xlo == BIG) return;                                   // no hypothesis of this pixel touches the image (quad-uniform)
    unsigned wide = 0;
#pragma unroll
    for (int h = 0; h < HPL; ++h) wide |= (own[h].valid && (own[h].x0 - xlo > 6 || own[h].y0 - ylo > 6)) ? 1u : 0u;
    const bool fits = quad_or(wide) == 0u;
    const int nchunks = fits ? 1 : NH;
    for (int ch = 0; ch < nchunks; ++ch) {
        bool act[HPL];
#pragma unroll
        for (int h = 0; h < HPL; ++h) act[h] = own[h].valid && (fits || q + 4 * h == ch);
        int xmin = xlo, ymin = ylo;
        if (!fits) {                                          // rare: one hypothesis per chunk, anchored at its own footprint
            int ax = BIG, ay = BIG;
#pragma unroll
            for (int h = 0; h < HPL; ++h) {
                ax = min(ax, act[h] ? own[h].x0 : BIG);
                ay = min(ay, act[h] ? own[h].y0 : BIG);
            }
            xmin = quad_min(ax);
            ymin = quad_min(ay);
        }
        if (xmin == BIG) continue;                            // this chunk's hypothesis is invalid (quad-uniform)
        unsigned mlo = 0, mhi = 0;
        float ur[HPL], vr[HPL];
#pragma unroll
        for (int h = 0; h < HPL; ++h) {
            const int cx = act[h] ? own[h].x0 - xmin : 0, cy = act[h] ? own[h].y0 - ymin : 0;
            const unsigned long long bits = act[h] ? (0x0303ull << (cy * 8 + cx)) : 0ull;
            mlo |= (unsigned)bits;
            mhi |= (unsigned)(bits >> 32);
            ur[h] = act[h] ? own[h].u - (float)xmin : -4.0f;     // -4: every hat weight of an inactive hypothesis is 0
            vr[h] = act[h] ? own[h].v - (float)ymin : -4.0f;
        }
        mlo = quad_or(mlo);
        mhi = quad_or(mhi);
        if (xmin < 0 || ymin < 0 || xmin + 8 > Ws || ymin + 8 > Hs) {
            // texels outside the image are grid_sample's zero padding: drop their bits here, so that the loop below needs
            // neither bounds tests nor clamped addresses.  xmin, ymin >= -1 (footprints with both columns / rows outside are
            // invalid hypotheses) and <= size - 1, so at most the first column / row and a trailing run fall outside.
            const int clo = xmin < 0 ? 1 : 0, chi = min(8, Ws - xmin), rlo = ymin < 0 ? 1 : 0, rhi = min(8, Hs - ymin);
            const unsigned colbits = ((0xffu >> (8 - chi)) & (0xffu << clo) & 0xffu) * 0x01010101u;
            const unsigned long long rowmask = (~0ull >> (64 - 8 * rhi)) & (~0ull << (8 * rlo));
            mlo &= (unsigned)rowmask & colbits;
            mhi &= (unsigned)(rowmask >> 32) & colbits;
        }
#if DMVS_GC_EXP == 1
        const unsigned texel_off = view_off;                  // (the caller put the pixel's own texel there and passes pitch 0)
#else
        const unsigned texel_off = view_off + (unsigned)(__mul24(ymin, pitch) + xmin) * (unsigned)TB;    // of grid cell (0, 0)
#endif
        // rows 0..3 of the grid (mlo), then -- rarely non-empty -- rows 4..7 (mhi): 32-bit bit scans
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
        unsigned m = half ? mhi : mlo;
        const int rbase = half * 4;
        // Round 4, measured and removed: a software-pipelined form of this loop (the next trip's loads issued before this trip's
        // arithmetic, into a second register set; exec-masked asm loads and hand-placed vmcnt, because hipcc's wait-count insertion
        // merges the two paths of a per-lane `if` to vmcnt(0)).  121 instead of 75 VGPRs (4 instead of 6 waves per SIMD) and ~15 % more
        // instructions: 633 vs 560 us per B=96 launch on noise geometry, slower on every geometry and batch
        // (profiles/r4_getcost_pipelined_ab_b96.jsonl).  The kernel is not waiting on any single load chain; DESIGN.md 3.1.
        while (m != 0u) {
            // TPT texels per trip (their loads in flight together); a pixel that runs out repeats its last texel with weight 0
            int bit[TPT];
            bool has[TPT];
#pragma unroll
            for (int i = 0; i < TPT; ++i) {
                has[i] = m != 0u;
                bit[i] = (i == 0 || has[i]) ? __ffs((int)m) - 1 : bit[i > 0 ? i - 1 : 0];
                m &= m - 1u;
            }
            Feat<C, FT> t[TPT];
            float fc[TPT], fr[TPT];
#pragma unroll
            for (int i = 0; i < TPT; ++i) {
                const int c = bit[i] & 7, r = (bit[i] >> 3) + rbase;
                // r * pitch + c < 2^24: one full-rate 24-bit multiply-add each (left alone, hipcc picks the 64-bit v_mad_u64_u32)
#if DMVS_GC_EXP == 1
                t[i].load(base + mad_u24(mad_u24((unsigned)r, (unsigned)pitch, (unsigned)(c & 1)), (unsigned)TB, texel_off));
#else
                t[i].load(base + mad_u24(mad_u24((unsigned)r, (unsigned)pitch, (unsigned)c), (unsigned)TB, texel_off));
#endif
                fc[i] = (float)c;
                fr[i] = (float)r;
            }
            // every load of the trip is issued before anything waits on one: without this fence the scheduler, chasing one
            // more wave of occupancy, re-uses one texel's registers and serialises load -> wait -> FMAs per texel
            __builtin_amdgcn_sched_barrier(0);
#if DMVS_GC_EXP == 4
#pragma unroll
            for (int i = 0; i < TPT; ++i)
#pragma unroll
                for (int j = 0; j < Feat<C, FT>::NW; ++j) asm volatile("" ::"v"(t[i].w[j]));      // the loads must land; nothing is computed from them
            (void)ur; (void)vr; (void)fc; (void)fr; (void)has;
            continue;
#endif
            float dd[TPT], w[TPT][HPL];
            dot_texels<C, FT, TPT>(t, ref, dd);
#if DMVS_GC_EXP == 2
#pragma unroll
            for (int i = 0; i < TPT; ++i) acc[0] += has[i] ? dd[i] * wscale : 0.0f;
            (void)w; (void)ur; (void)vr; (void)fc; (void)fr;
#else
#pragma unroll
            for (int i = 0; i < TPT; ++i) {
                dd[i] = has[i] ? dd[i] * wscale : 0.0f;
#pragma unroll
                for (int h = 0; h < HPL; ++h) w[i][h] = hat(ur[h], fc[i]) * hat(vr[h], fr[i]);
            }
#pragma unroll
            for (int i = 0; i < TPT; i += 2) scatter_pair<NH, HPL>(acc, w[i], dd[i], w[i + 1], dd[i + 1]);
#endif
        }
        }
    }
}

// ------------------------------------------------------------------------------------------ GetCost
#ifndef DMVS_GC_BLOCK       // threads per GetCost workgroup = 4 x pixels of its tile; log2 of the tile width (diagnostic builds vary both)
#define DMVS_GC_BLOCK DMVS_BLOCK
#endif
#ifndef DMVS_GC_TW_SHIFT
#define DMVS_GC_TW_SHIFT 5
#endif
template <int C, int N, int TPT, int FT>
__global__ void __launch_bounds__(DMVS_GC_BLOCK) getcost_quad_kernel(const dmvs_getcost_desc d) {
    constexpr int HPL = (N + 3) / 4, PPB = DMVS_GC_BLOCK / 4;
    constexpr int tw_shift = DMVS_GC_TW_SHIFT;
    const int q = threadIdx.x & 3;
    const int H = d.H, W = d.W;
    const int hw = H * W;
    // grid = (64-pixel tiles of one image, B): the batch item is workgroup-uniform, so cameras, depth range and every
    // tensor base are scalar registers / scalar loads
    const int b = blockIdx.y;
    // The workgroup's 64 pixels are a 32 x 2 TILE, a wave = 16 consecutive pixels of one row (reference loads and cost stores stay 64-byte
    // runs); until round 5 they were a 64-pixel row segment.  The two pixel rows of a tile read the same source rows.  Launched back to back
    // on a warm GPU (profiles/r5_getcost_mapping_sweep_b96.jsonl: every tile width x workgroup size as a variant build, all bit-identical):
    // 569 -> 545 us per B=96 launch on noise geometry, 639 -> 607 with random confidences, 558 -> 529 on scene geometry; 16 x 4 tiles and
    // 128- / 64-thread workgroups are within 2 % of this, 512 threads and 8 x 8 tiles slower.  Inside the model's step, where the feature
    // maps have just been evicted by the convolutions in between, it is 576 -> 570 us.  The tile width is a compile-time constant.
    const int tw_mask = (1 << tw_shift) - 1, th = PPB >> tw_shift;
    const int tiles_x = (W + tw_mask) >> tw_shift;
    const int tile = (int)dmvs_xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int p_in = threadIdx.x >> 2;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int xx = (tx << tw_shift) + (p_in & tw_mask), yy = ty * th + (p_in >> tw_shift);
    const bool live = xx < W && yy < H;
    const int yx = live ? yy * W + xx : hw - 1;
    const int y = yx / W, x = yx - y * W;
    const long pc = (long)b * hw + yx;

    // hypotheses in normalised inverse depth (reference :259-276); this lane projects hypotheses q and q + 4
    const float cur_inv = d.inv_depth[pc];
    float radius = (float)(N / 2) * d.interval;
    if (d.confidence) {
        const float r0 = d.min_radius * radius, r1 = d.max_radius * radius;
        radius = r0 + (1.0f - d.confidence[pc]) * (r1 - r0);
    }
    const float lo = cur_inv - radius, hi = cur_inv + radius;
    const float step = (hi - lo) / (float)(N - 1);
    const float dmin = d.disp_min[b], dmax = d.disp_max[b];
    float own_depth[HPL];
    bool exists[HPL];
#pragma unroll
    for (int h = 0; h < HPL; ++h) {
        const int k = q + 4 * h;
        exists[h] = k < N;
        float sk = (float)(exists[h] ? k : 0) * step;
        sk += lo;
        sk = fminf(fmaxf(sk, 0.0f), 1.0f);
        own_depth[h] = dmvs_disp_to_depth(sk, dmin, dmax);
        if (live && exists[h]) d.out_samples[((long)b * d.samp_cstride + d.samp_coffset + k) * (long)hw + yx] = sk;
    }

    float ref[C / 4];
    load_ref<C, FT>(d.ref, pc, q, ref);

    float acc[N];
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = 0.0f;
    float wsum = 1e-8f;
    const int Hv = H >> d.vw_shift, Wv = W >> d.vw_shift;
    const int vwi = (y >> d.vw_shift) * Wv + (x >> d.vw_shift);
    const char* base = reinterpret_cast<const char*>(d.src);
    for (int s = 0; s < d.S; ++s) {
        const float w = d.view_w[((long)b * d.S + s) * (long)(Hv * Wv) + vwi];
        wsum += w;
        RayQ ray;
        ray.init(d.rt + ((long)b * d.S + s) * 12, (float)x, (float)y);
        HypQ own[HPL];
#pragma unroll
        for (int h = 0; h < HPL; ++h) own[h] = project_q(ray, own_depth[h], exists[h], H, W);
        // the view's image: a 64-bit workgroup-uniform (scalar) base, 32-bit offsets inside the view
        const char* vbase = base + ((long)s * d.B + b) * (long)hw * Feat<C, FT>::TEXEL_BYTES;
#if DMVS_GC_EXP == 1
        quad_accumulate<C, FT, N, TPT>(vbase, Feat<C, FT>::lane_bytes(q) + (unsigned)min(yx, hw - 2) * (unsigned)Feat<C, FT>::TEXEL_BYTES, 0, own, H, W, ref, w, acc);
#else
        quad_accumulate<C, FT, N, TPT>(vbase, Feat<C, FT>::lane_bytes(q), W, own, H, W, ref, w, acc);
#endif
    }
    if (live) {
        const float inv_w = 1.0f / wsum;
#pragma unroll
        for (int k = 0; k < N; ++k)
            d.out_cost[((long)b * d.cost_cstride + d.cost_coffset + q * N + k) * (long)hw + yx] = acc[k] * inv_w;
    }
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
        quad_accumulate<C, FT, NB, TPT>(base, view_off, Ws, own, Hs, Ws, ref, 1.0f, acc);
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

template <int C, int TPT, int FT, int BAND_BYTES>
__global__ void __launch_bounds__(DMVS_BLOCK)
warp_init_band_kernel(const void* __restrict__ ref_f, const void* __restrict__ src, const float* __restrict__ rt,
                      const float* __restrict__ disp_min, const float* __restrict__ disp_max, float* __restrict__ out, int B, int S,
                      int D, int H, int W, int Hs, int Ws, int tiles_x) {
    constexpr int NB = 8, HPL = 2, TB = Feat<C, FT>::TEXEL_BYTES, NWAVE = DMVS_BLOCK / 64;
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
            for (int i0 = wave * 64; i0 < npieces; i0 += DMVS_BLOCK) {
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
            if (staged) quad_accumulate<C, FT, NB, TPT>((lds_cptr)band, band_off, ncols, own, Hs, Ws, ref, 1.0f, acc);
            else quad_accumulate<C, FT, NB, TPT>(gbase, view_off, Ws, own, Hs, Ws, ref, 1.0f, acc);
            store_planes<NB>(view_out, lane_off, d0, D, hw, live, acc);
        }
        c0 += cnt;
    }
}

}  // namespace

// texels per trip of the texel loop: 2 (4 measured 8-10 % slower on the MI355X: one wave per SIMD less, more idle slots when a
// pixel's texel count is not a multiple of the trip)
#ifndef DMVS_QUAD_TPT       // (diagnostic builds: the ceiling probe is also timed with 4 texels in flight per trip)
#define DMVS_QUAD_TPT 2
#endif
constexpr int QUAD_TPT = DMVS_QUAD_TPT;

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
    if (d.G != 4 || !d.ref || !d.src || !d.rt || !d.inv_depth || !d.view_w || !d.out_cost || !d.out_samples) return DMVS_EINVAL;
    if (d.feat_dtype < DMVS_DTYPE_F32 || d.feat_dtype > DMVS_DTYPE_F32_PLAIN) return DMVS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // 24-bit row multiplies and 32-bit byte offsets inside ONE view's image (2^24 texels x <= 192 bytes < 2^32); the source
    // stack as a whole may be any size (64-bit per-view bases); grid.y
    if ((long)d.H * d.W >= (1L << 24) || d.B > 65535) return DMVS_EINVAL;
    // 32 x 2-pixel tiles (a ragged last tile column idles its surplus lanes: every stage-2 / stage-3 width of the reference's datasets but
    // DTU's 400 is a multiple of 32)
    constexpr int tw = 1 << DMVS_GC_TW_SHIFT, th = (DMVS_GC_BLOCK / 4) >> DMVS_GC_TW_SHIFT;
    dim3 grid((unsigned)(((d.W + tw - 1) / tw) * ((d.H + th - 1) / th)), (unsigned)d.B), block(DMVS_GC_BLOCK);
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
