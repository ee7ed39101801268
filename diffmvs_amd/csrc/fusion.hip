// Geometric consistency check of the depth-map fusion step (reference filter.py:8-93, :230-259): for every reference pixel
// and every source view -- project the reference depth into the source view, sample the source depth map there
// (cv2.remap INTER_LINEAR, constant-0 border), lift the sampled point back into the reference view, and test the
// reprojection distance and the relative depth difference against one or several threshold pairs.
// One lane per reference pixel, source views in a loop; HBM-trivial (each map read once), arithmetic in fp64 exactly where
// the reference's NumPy promotion makes it fp64 (the pixel grid is int64, so everything downstream of `grid * depth` is).
#include "dmvs_common.h"

namespace {

struct Mats {      // per source view, row-major, the fp32-valued matrices the caller composed in the reference's dtypes
    float kref_inv[9], src_from_ref[12], ksrc[9], ksrc_inv[9], ref_from_src[12], kref[9];
};
static_assert(sizeof(Mats) == DMVS_GEO_MATS_FLOATS * 4, "layout documented in dmvs.h");

__device__ __forceinline__ void mul3(const float* m, double x, double y, double z, double& ox, double& oy, double& oz) {
    ox = (double)m[0] * x + (double)m[1] * y + (double)m[2] * z;
    oy = (double)m[3] * x + (double)m[4] * y + (double)m[5] * z;
    oz = (double)m[6] * x + (double)m[7] * y + (double)m[8] * z;
}
__device__ __forceinline__ void mul34(const float* m, double x, double y, double z, double& ox, double& oy, double& oz) {
    ox = (double)m[0] * x + (double)m[1] * y + (double)m[2] * z + (double)m[3];
    oy = (double)m[4] * x + (double)m[5] * y + (double)m[6] * z + (double)m[7];
    oz = (double)m[8] * x + (double)m[9] * y + (double)m[10] * z + (double)m[11];
}

// cv2.remap(src, mapx, mapy, INTER_LINEAR) with the default constant-0 border, single-channel fp32: coordinates are
// rounded to 1/32 pixel (round half to even), the four taps are blended with fp32 weights (1-fy)(1-fx), (1-fy)fx, ...
__device__ __forceinline__ float remap_linear(const float* __restrict__ img, int Hs, int Ws, float mx, float my) {
    if (!(fabsf(mx) < 1.0e7f) || !(fabsf(my) < 1.0e7f)) return 0.0f;      // NaN / far outside: every tap is border
    const int sx = (int)rintf(mx * 32.0f), sy = (int)rintf(my * 32.0f);
    const int ix = sx >> 5, iy = sy >> 5;
    const float fx = (float)(sx & 31) * (1.0f / 32.0f), fy = (float)(sy & 31) * (1.0f / 32.0f);
    const float w00 = (1.0f - fy) * (1.0f - fx), w01 = (1.0f - fy) * fx, w10 = fy * (1.0f - fx), w11 = fy * fx;
    const bool x0 = (unsigned)ix < (unsigned)Ws, x1 = (unsigned)(ix + 1) < (unsigned)Ws;
    const bool y0 = (unsigned)iy < (unsigned)Hs, y1 = (unsigned)(iy + 1) < (unsigned)Hs;
    const float v00 = (x0 && y0) ? img[(long)iy * Ws + ix] : 0.0f;
    const float v01 = (x1 && y0) ? img[(long)iy * Ws + ix + 1] : 0.0f;
    const float v10 = (x0 && y1) ? img[(long)(iy + 1) * Ws + ix] : 0.0f;
    const float v11 = (x1 && y1) ? img[(long)(iy + 1) * Ws + ix + 1] : 0.0f;
    return v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
}

__global__ void __launch_bounds__(DMVS_BLOCK)
geo_consistency_kernel(const float* __restrict__ depth_ref, const float* __restrict__ depth_src, const float* __restrict__ mats,
                       const double* __restrict__ pix_thres, const float* __restrict__ rel_thres, int L, int use_range, float range_min,
                       float range_max, int* __restrict__ level_counts, float* __restrict__ depth_sum, int S, int H, int W, int Hs, int Ws) {
    const long p = (long)blockIdx.x * DMVS_BLOCK + threadIdx.x;
    if (p >= (long)H * W) return;
    const int y = (int)(p / W), x = (int)(p - (long)y * W);
    const float dref = depth_ref[p];
    const bool in_range = !use_range || (dref > range_min && dref < range_max);
    int counts[DMVS_GEO_MAX_LEVELS];
#pragma unroll
    for (int l = 0; l < DMVS_GEO_MAX_LEVELS; ++l) counts[l] = 0;
    float dsum = 0.0f;
    for (int s = 0; s < S; ++s) {
        const Mats& m = *reinterpret_cast<const Mats*>(mats + (long)s * DMVS_GEO_MATS_FLOATS);
        // reference pixel -> source view (filter.py:20-31)
        const double dd = (double)dref;
        double cx, cy, cz, sx3, sy3, sz3, kx, ky, kz;
        mul3(m.kref_inv, (double)x * dd, (double)y * dd, dd, cx, cy, cz);
        mul34(m.src_from_ref, cx, cy, cz, sx3, sy3, sz3);
        mul3(m.ksrc, sx3, sy3, sz3, kx, ky, kz);
        const double us = kx / kz, vs = ky / kz;
        const float sampled = remap_linear(depth_src + (long)s * Hs * Ws, Hs, Ws, (float)us, (float)vs);      // :32-35
        // sampled source point -> back into the reference view (:37-51)
        const double sd = (double)sampled;
        double bx, by, bz, rx, ry, rz;
        mul3(m.ksrc_inv, us * sd, vs * sd, sd, bx, by, bz);
        mul34(m.ref_from_src, bx, by, bz, rx, ry, rz);
        const float depth_reproj = (float)rz;
        double qx, qy, qz;
        mul3(m.kref, rx, ry, rz, qx, qy, qz);
        if (qx == 0.0) qx = 1e-5;
        if (qy == 0.0) qy = 1e-5;
        if (qz == 0.0) qz = 1e-5;
        const float xr = (float)fmin(fmax(qx / qz, -1e8), 1e8), yr = (float)fmin(fmax(qy / qz, -1e8), 1e8);
        // :80-91 (dist in fp64: fp32 map minus the int64 grid), relative depth difference in fp32
        const double ex = (double)xr - (double)x, ey = (double)yr - (double)y;
        const double dist = sqrt(ex * ex + ey * ey);
        const float rel = fabsf(depth_reproj - dref) / dref;
        bool last = false;
        for (int l = 0; l < L; ++l) {
            const bool ok = dist < pix_thres[l] && rel < rel_thres[l] && in_range;
            counts[l] += ok ? 1 : 0;
            last = ok;
        }
        dsum += last ? depth_reproj : 0.0f;          // depth_reproj[~mask] = 0 with the LAST level's mask (:91, :257)
    }
    for (int l = 0; l < L; ++l) level_counts[(long)l * H * W + p] = counts[l];
    depth_sum[p] = dsum;
}

}  // namespace

extern "C" int dmvs_geo_consistency_f32(const float* depth_ref, const float* depth_src, const float* mats, const double* pix_thres,
                                        const float* rel_thres, int32_t L, int32_t use_range, float range_min, float range_max,
                                        int32_t* level_counts, float* depth_sum, int32_t S, int32_t H, int32_t W, int32_t Hs, int32_t Ws,
                                        void* stream) {
    if (!depth_ref || !depth_src || !mats || !pix_thres || !rel_thres || !level_counts || !depth_sum) return DMVS_EINVAL;
    if (L < 1 || L > DMVS_GEO_MAX_LEVELS || S < 1 || H < 1 || W < 1 || Hs < 1 || Ws < 1) return DMVS_EINVAL;
    dim3 grid(dmvs_ceil_div((long)H * W, DMVS_BLOCK)), block(DMVS_BLOCK);
    hipLaunchKernelGGL(geo_consistency_kernel, grid, block, 0, (hipStream_t)stream, depth_ref, depth_src, mats, pix_thres, rel_thres, L,
                       use_range, range_min, range_max, level_counts, depth_sum, S, H, W, Hs, Ws);
    return dmvs_launch_status();
}
