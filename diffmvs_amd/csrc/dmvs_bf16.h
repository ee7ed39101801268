// bf16 matrix-core helpers shared by the 2-D convolutions (conv2d_tiled.h) and the fused FeatureNet stem (stem.hip): the operand type of
// v_mfma_f32_16x16x32_bf16, round-to-nearest conversions, and the three-way split of fp32 values behind DMVS_ARITH_SPLIT.
#pragma once
#include "dmvs_common.h"

typedef float f32x4_bf __attribute__((ext_vector_type(4)));

// 8 bf16 operand values of a lane for v_mfma_f32_16x16x32_bf16 (k-slots 8*(lane>>4) .. +7), as raw 16-bit patterns
#ifdef DMVS_HOST_EMULATION
typedef hipemu_s16x8 bf16x8;
__device__ __forceinline__ bf16x8 dmvs_pack_bf16x8(const float (&v)[8]) {
    bf16x8 r;
    for (int j = 0; j < 8; ++j) r[j] = (short)dmvs_f32_to_bf16(v[j]);
    return r;
}
__device__ __forceinline__ f32x4_bf dmvs_mfma_bf16(bf16x8 a, bf16x8 b, f32x4_bf c) { return hipemu_mfma_f32_16x16x32_bf16(a, b, c); }
#else
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 dmvs_pack_bf16x8(const float (&v)[8]) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const bf16x2 p = __builtin_convertvector(f32x2{v[j], v[j + 1]}, bf16x2);       // v_cvt_pk_bf16_f32: round to nearest even
        const s16x2 q = __builtin_bit_cast(s16x2, p);
        r[j] = q[0];
        r[j + 1] = q[1];
    }
    return r;
}
__device__ __forceinline__ f32x4_bf dmvs_mfma_bf16(bf16x8 a, bf16x8 b, f32x4_bf c) {
    typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, a), __builtin_bit_cast(mfma_bf16x8, b), c, 0, 0, 0);
}
#endif

// DMVS_ARITH_SPLIT: an fp32 value as the sum of three bf16 values, hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), each rounded to
// nearest even: the two remainders are exact in fp32, |mid| <= 2^-9 |x|, |lo| <= 2^-18 |x|, and x - (hi + mid + lo) is below 2^-27 |x|.  (Splitting
// by truncation is exact but gives all three parts the sign of x: the dropped partial products then all have the product's sign and add up to
// a bias of ~1e-7 x sum |a b| -- measured 2.4x the fma chain's error on a 7x7 layer.  Rounded parts have independent signs.)  For 8 values at
// once, packed as the three k-slot operands of v_mfma_f32_16x16x32_bf16: 1.5 conversions + 2 unpacks + 2 subtractions per value.  (inf -> NaN.)
__device__ __forceinline__ void dmvs_bf16x8_to_f32(const bf16x8& p, float (&f)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = __uint_as_float((uint32_t)(uint16_t)p[j] << 16);
}
__device__ __forceinline__ void dmvs_split3_bf16x8(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    float f[8], r1[8], r2[8];
    h = dmvs_pack_bf16x8(v);
    dmvs_bf16x8_to_f32(h, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) r1[j] = v[j] - f[j];
    m = dmvs_pack_bf16x8(r1);
    dmvs_bf16x8_to_f32(m, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) r2[j] = r1[j] - f[j];
    l = dmvs_pack_bf16x8(r2);
}

