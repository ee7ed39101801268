// Shared pieces of the quad-per-pixel homography-warp kernels (warp_quad.hip: GetCost and the stage-1 plane sweep; csrc/probe/getcost_probe.hip:
// the bench-only memory-system probe of GetCost's address stream).  Everything here has internal linkage.
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
// Semantics kept from the reference (models/module.py:181-218): per-tap zero padding with align_corners=True pixel coordinates, NO
// behind-camera mask, z == 0 -> z + 1e-8, non-finite coordinates sample 0.
#pragma once
#include <utility>

#include "dmvs_common.h"

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
                const u32x4 v = ldv<u32x4>(p + j * PITCH);
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

// The arithmetic of a texel-loop trip (the Body of quad_accumulate): group dot of each texel with the lane's reference channels,
// the bilinear hat weights of the lane's own hypotheses, the DPP scatter into every hypothesis of the pixel.
struct QuadProduct {
    static constexpr bool kPairUnits = false;      // load unit of the texel loop: one texel (see quad_accumulate)
    template <int C, int FT, int NH, int TPT>
    static __device__ __forceinline__ void texels(const Feat<C, FT> (&t)[TPT], const bool (&has)[TPT], const float (&fc)[TPT], const float (&fr)[TPT],
                                                  const float (&ur)[(NH + 3) / 4], const float (&vr)[(NH + 3) / 4], const float (&ref)[C / 4],
                                                  float wscale, float (&acc)[NH]) {
        constexpr int HPL = (NH + 3) / 4;
        float dd[TPT], w[TPT][HPL];
        dot_texels<C, FT, TPT>(t, ref, dd);
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
            dd[i] = has[i] ? dd[i] * wscale : 0.0f;
#pragma unroll
            for (int h = 0; h < HPL; ++h) w[i][h] = hat(ur[h], fc[i]) * hat(vr[h], fr[i]);
        }
#pragma unroll
        for (int i = 0; i < TPT; i += 2) scatter_pair<NH, HPL>(acc, w[i], dd[i], w[i + 1], dd[i + 1]);
    }
};

// acc[k] += wscale * (bilinear sample of the lane's group dot at hypothesis k), k < NH, for one (pixel, view).
// own[h] = hypothesis q + 4h of the pixel (q = lane & 3).  Texel (x, y) of the view is read at base + origin + (y * pitch + x) *
// TEXEL_BYTES (32-bit wrapping arithmetic): global memory -- base + origin = the view's [Hs,Ws,C] NHWC-g4 image (+ the lane's
// 16q bytes), pitch = Ws -- or a band of the view staged in LDS (base = the band, origin = lane bytes - the band's corner).
//
// Body = what happens to the TPT texels of a loop trip once their loads are issued.  QuadProduct (below) is the arithmetic of the path:
// group dot, hat weights, scatter.  The memory-system probe of tools/ (csrc/probe/getcost_probe.hip, a bench-only library) passes a body
// that only waits for the loaded registers: the projection, the texel masks, the bit scans, the addresses and the loads -- this function
// up to the call of Body::texels -- are then the product's, instruction for instruction.
template <typename Body, int C, int FT, int NH, int TPT, typename P>
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
        const unsigned texel_off = view_off + (unsigned)(__mul24(ymin, pitch) + xmin) * (unsigned)TB;    // of grid cell (0, 0)
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
            [[maybe_unused]] bool has_r[TPT];
#pragma unroll
            for (int i = 0; i < TPT; ++i) {
                has[i] = m != 0u;
                bit[i] = (i == 0 || has[i]) ? __ffs((int)m) - 1 : bit[i > 0 ? i - 1 : 0];
                if constexpr (Body::kPairUnits) {      // the unit also covers the texel's right-hand neighbour in the row: both bits leave the mask
                    const unsigned b1 = has[i] ? 1u << bit[i] : 0u, nb = (bit[i] & 7) != 7 ? b1 << 1 : 0u;
                    has_r[i] = (m & nb) != 0u;
                    m &= ~(b1 | nb);
                } else {
                    m &= m - 1u;
                }
            }
            Feat<C, FT> t[TPT];
            float fc[TPT], fr[TPT];
#pragma unroll
            for (int i = 0; i < TPT; ++i) {
                const int c = bit[i] & 7, r = (bit[i] >> 3) + rbase;
                fc[i] = (float)c;
                fr[i] = (float)r;
                if constexpr (!Body::kPairUnits) {
                    // r * pitch + c < 2^24: one full-rate 24-bit multiply-add each (left alone, hipcc picks the 64-bit v_mad_u64_u32)
                    t[i].load(base + mad_u24(mad_u24((unsigned)r, (unsigned)pitch, (unsigned)c), (unsigned)TB, texel_off));
                }
            }
            if constexpr (Body::kPairUnits) {
                // PAIR UNITS (csrc/probe only so far: what would a 16-bit C = 16 feature layout buy whose load unit is the texel pair (x, x + 1) --
                // 64 contiguous bytes, the request of one fp32 C = 32 unit -- instead of the 32-byte texel?).  The quad reads the pair with ONE
                // 16-byte load per lane (lane q: bytes 16q .. 16q + 15 of the pair; view_off already holds the lane's 8q of the single-texel form).
                static_assert(Feat<C, FT>::NW == 2, "pair units: 16-bit features with 16 channels");
                Feat<C, FT> tr[TPT];
#pragma unroll
                for (int i = 0; i < TPT; ++i) {
                    const int c = bit[i] & 7, r = (bit[i] >> 3) + rbase;
                    const u32x4 v = ldv<u32x4>(base + (mad_u24(mad_u24((unsigned)r, (unsigned)pitch, (unsigned)c), (unsigned)TB, texel_off) + (unsigned)q * (TB / 4)));
                    t[i].w[0] = v[0]; t[i].w[1] = v[1];
                    tr[i].w[0] = v[2]; tr[i].w[1] = v[3];
                }
                __builtin_amdgcn_sched_barrier(0);
                float fcr[TPT];
#pragma unroll
                for (int i = 0; i < TPT; ++i) fcr[i] = fc[i] + 1.0f;
                Body::template texels<C, FT, NH, TPT>(t, has, fc, fr, ur, vr, ref, wscale, acc);
                Body::template texels<C, FT, NH, TPT>(tr, has_r, fcr, fr, ur, vr, ref, wscale, acc);
                continue;
            }
            // every load of the trip is issued before anything waits on one: without this fence the scheduler, chasing one
            // more wave of occupancy, re-uses one texel's registers and serialises load -> wait -> FMAs per texel
            __builtin_amdgcn_sched_barrier(0);
            Body::template texels<C, FT, NH, TPT>(t, has, fc, fr, ur, vr, ref, wscale, acc);
        }
        }
    }
}

// ------------------------------------------------------------------------------------------ GetCost
// threads per GetCost workgroup (= 4 x the pixels of its tile) and log2 of the tile width: a 32 x 2-pixel tile.  (Round 5 swept both as
// variant builds -- profiles/r5_getcost_mapping_sweep_b96.jsonl -- and settled here; they are plain constants now.)
constexpr int GC_BLOCK = DMVS_BLOCK, GC_TW_SHIFT = 5;
// texels per trip of the texel loop: 2 (4 measured 8-10 % slower on the MI355X: one wave per SIMD less, more idle slots when a
// pixel's texel count is not a multiple of the trip)
constexpr int QUAD_TPT = 2;

// The whole GetCost kernel as a device function of its texel Body (QuadProduct: the path; the loads-only body of csrc/probe: the
// memory-system ceiling of this very address stream).  The __global__ wrappers live in the translation units.
template <typename Body, int C, int N, int TPT, int FT>
__device__ __forceinline__ void getcost_quad_body(const dmvs_getcost_desc& d) {
    constexpr int HPL = (N + 3) / 4, PPB = GC_BLOCK / 4;
    constexpr int tw_shift = GC_TW_SHIFT;
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
        quad_accumulate<Body, C, FT, N, TPT>(vbase, Feat<C, FT>::lane_bytes(q), W, own, H, W, ref, w, acc);
    }
    if (live) {
        const float inv_w = 1.0f / wsum;
#pragma unroll
        for (int k = 0; k < N; ++k)
            d.out_cost[((long)b * d.cost_cstride + d.cost_coffset + q * N + k) * (long)hw + yx] = acc[k] * inv_w;
    }
}
// launch grid of the GetCost kernels: (32 x 2-pixel tiles of one image, B).  A ragged last tile column idles its surplus lanes: every stage-2 /
// stage-3 width of the reference's datasets but DTU's 400 is a multiple of 32.
static inline dim3 getcost_grid(const dmvs_getcost_desc& d) {
    constexpr int tw = 1 << GC_TW_SHIFT, th = (GC_BLOCK / 4) >> GC_TW_SHIFT;
    return dim3((unsigned)(((d.W + tw - 1) / tw) * ((d.H + th - 1) / th)), (unsigned)d.B);
}

// descriptor checks shared by the entry points that launch getcost_quad_body
static inline bool getcost_desc_ok(const dmvs_getcost_desc& d) {
    if (d.G != 4 || !d.ref || !d.src || !d.rt || !d.inv_depth || !d.view_w || !d.out_cost || !d.out_samples) return false;
    if (d.feat_dtype < DMVS_DTYPE_F32 || d.feat_dtype > DMVS_DTYPE_F32_PLAIN) return false;
    // 24-bit row multiplies and 32-bit byte offsets inside ONE view's image (2^24 texels x <= 192 bytes < 2^32); the source
    // stack as a whole may be any size (64-bit per-view bases); grid.y
    return (long)d.H * d.W < (1L << 24) && d.B <= 65535;
}

}  // namespace
