// Training-mode BatchNorm (+ optional ReLU) over [B, C, S] fp32 tensors, S = H*W or D*H*W: forward with batch statistics
// and running-stat update, and backward.  Replaces nn.BatchNorm2d/3d inside module.Conv2d / Conv3d / ConvBnReLU / ConvBn
// in train mode (reference models/module.py:24-58, :60-96, :279-301).  HBM-bound: forward reads x twice and writes y,
// backward reads (x, dy) twice and writes dx.  No atomics: per-chunk partial sums go through caller-owned scratch
// and are folded in double precision by one workgroup per channel => run-to-run reproducible statistics/gradients.
//
// `views` > 1 normalises V independent sub-batches in one launch: the reference applies FeatureNet once per image of
// the view stack and PixelViewWeight once per source view (diffusion.py:156-157, module.py:533), so every such call
// has its own batch statistics and updates the running statistics once, in view order.  Batching the V calls into one
// tensor (rows b = v*B/V + i if view_major, else b = i*V + v) keeps those semantics with V times fewer launches.
#include "dmvs_common.h"

namespace {
constexpr int BN_CHUNK = 16384;            // elements of one (b, c) row per partial-sum workgroup

__device__ __forceinline__ float2 block_sum2(float a, float b, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64);
        b += __shfl_down(b, o, 64);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        red[wave * 2] = a;
        red[wave * 2 + 1] = b;
    }
    __syncthreads();
    float2 r = make_float2(0.0f, 0.0f);
    if (threadIdx.x == 0) {
        for (int w = 0; w < DMVS_BLOCK / 64; ++w) {
            r.x += red[w * 2];
            r.y += red[w * 2 + 1];
        }
    }
    return r;     // valid on thread 0
}

// partial[c][chunk] = (sum x, sum x^2)   |   backward: (sum dz, sum dz*xhat),  dz = dy * [act passes]
template <bool BWD>
__global__ void __launch_bounds__(DMVS_BLOCK)
bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                  const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
                  float2* __restrict__ partial, int C, int S, int chunks_per_row, int relu, int views, int rows_per_view,
                  int view_major) {
    __shared__ float red[2 * DMVS_BLOCK / 64];
    const int c = blockIdx.y, chunk = blockIdx.x;
    const int b = chunk / chunks_per_row, k = chunk - b * chunks_per_row;
    const size_t row = ((size_t)b * C + c) * S;
    const int lo = k * BN_CHUNK, hi = lo + BN_CHUNK < S ? lo + BN_CHUNK : S;
    float s0 = 0.0f, s1 = 0.0f;
    float mu = 0.0f, rs = 0.0f, g = 1.0f, bt = 0.0f;
    if (BWD) {
        const int v = view_major ? b / rows_per_view : b % views;
        mu = mean[v * C + c];
        rs = rstd[v * C + c];
        g = gamma[c];
        bt = beta[c];
    }
    const bool vec = (S & 3) == 0;
    if (vec) {
        const float4* x4 = reinterpret_cast<const float4*>(x + row);
        const float4* d4 = BWD ? reinterpret_cast<const float4*>(dy + row) : nullptr;
        for (int i = (lo >> 2) + threadIdx.x; i < (hi >> 2); i += DMVS_BLOCK) {
            const float4 v = x4[i];
            if (!BWD) {
                s0 += (v.x + v.y) + (v.z + v.w);
                s1 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            } else {
                const float4 dv = d4[i];
                const float xs[4] = {v.x, v.y, v.z, v.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (xs[j] - mu) * rs;
                    const float dz = (relu && xh * g + bt <= 0.0f) ? 0.0f : ds[j];
                    s0 += dz;
                    s1 = fmaf(dz, xh, s1);
                }
            }
        }
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += DMVS_BLOCK) {
            const float v = x[row + i];
            if (!BWD) {
                s0 += v;
                s1 = fmaf(v, v, s1);
            } else {
                const float xh = (v - mu) * rs;
                const float dz = (relu && xh * g + bt <= 0.0f) ? 0.0f : dy[row + i];
                s0 += dz;
                s1 = fmaf(dz, xh, s1);
            }
        }
    }
    const float2 t = block_sum2(s0, s1, red);
    if (threadIdx.x == 0) partial[(size_t)c * gridDim.x + chunk] = t;
}

// one workgroup per channel: fold the partials of every view in double, views in order.
// forward : mean, rstd (biased variance) per view -> out0/out1 [views][C];  running stats <- (1-m)*running +
//           m*(mean | unbiased variance), once per view, sequentially (the reference's per-call updates)
// backward: per view dbeta_v = sum dz, dgamma_v = sum dz*xhat -> out0/out1 [views][C]; totals -> tot0/tot1 [C]
template <bool BWD>
__global__ void __launch_bounds__(DMVS_BLOCK)
bn_finalize_kernel(const float2* __restrict__ partial, int nchunk, int chunks_per_row, double count, float* __restrict__ out0,
                   float* __restrict__ out1, float* __restrict__ tot0, float* __restrict__ tot1,
                   float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps, int C,
                   int views, int rows_per_view, int view_major) {
    __shared__ double red[2 * DMVS_BLOCK / 64];
    const int c = blockIdx.x;
    double t0 = 0.0, t1 = 0.0;
    float rm = 0.0f, rv = 0.0f;
    if (!BWD && running_mean && threadIdx.x == 0) {
        rm = running_mean[c];
        rv = running_var[c];
    }
    for (int v = 0; v < views; ++v) {
        double a = 0.0, q = 0.0;
        for (int i = threadIdx.x; i < nchunk; i += DMVS_BLOCK) {
            const int b = i / chunks_per_row;
            if ((view_major ? b / rows_per_view : b % views) != v) continue;
            const float2 p = partial[(size_t)c * nchunk + i];
            a += (double)p.x;
            q += (double)p.y;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_down(a, o, 64);
            q += __shfl_down(q, o, 64);
        }
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        __syncthreads();
        if (lane == 0) {
            red[wave * 2] = a;
            red[wave * 2 + 1] = q;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            a = q = 0.0;
            for (int w = 0; w < DMVS_BLOCK / 64; ++w) {
                a += red[w * 2];
                q += red[w * 2 + 1];
            }
            if (BWD) {
                out0[v * C + c] = (float)a;      // dbeta of this view
                out1[v * C + c] = (float)q;      // dgamma of this view
                t0 += a;
                t1 += q;
            } else {
                const double mean = a / count;
                double var = q / count - mean * mean;
                var = var > 0.0 ? var : 0.0;
                out0[v * C + c] = (float)mean;
                out1[v * C + c] = (float)(1.0 / sqrt(var + (double)eps));
                if (running_mean) {
                    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                    rm = (float)((1.0 - momentum) * rm + momentum * mean);
                    rv = (float)((1.0 - momentum) * rv + momentum * unbiased);
                }
            }
        }
    }
    if (threadIdx.x == 0) {
        if (BWD) {
            tot0[c] = (float)t0;
            tot1[c] = (float)t1;
        } else if (running_mean) {
            running_mean[c] = rm;
            running_var[c] = rv;
        }
    }
}

// forward : y = act((x - mean) * rstd * gamma + beta)
// backward: dx = gamma * rstd * (dz - dbeta/N - xhat * dgamma/N)
template <bool BWD>
__global__ void __launch_bounds__(DMVS_BLOCK)
bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
                const float* __restrict__ dbeta, const float* __restrict__ dgamma, float* __restrict__ out, int C, int S,
                float inv_count, int relu, int views, int rows_per_view, int view_major) {
    const int bc = blockIdx.y, c = bc % C, b = bc / C;
    const int vc = (view_major ? b / rows_per_view : b % views) * C + c;
    const size_t row = (size_t)bc * S;
    const float mu = mean[vc], rs = rstd[vc], g = gamma[c], bt = beta[c];
    const float k1 = BWD ? dbeta[vc] * inv_count : 0.0f, k2 = BWD ? dgamma[vc] * inv_count : 0.0f;
    const float a = rs * g, b0 = bt - mu * rs * g;
    if ((S & 3) == 0) {
        const int i = blockIdx.x * DMVS_BLOCK + threadIdx.x;
        if (i >= (S >> 2)) return;
        const float4 v = reinterpret_cast<const float4*>(x + row)[i];
        float xs[4] = {v.x, v.y, v.z, v.w}, o[4];
        if (!BWD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = fmaf(xs[j], a, b0);
                o[j] = relu ? fmaxf(y, 0.0f) : y;
            }
        } else {
            const float4 dv = reinterpret_cast<const float4*>(dy + row)[i];
            const float ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (xs[j] - mu) * rs;
                const float dz = (relu && xh * g + bt <= 0.0f) ? 0.0f : ds[j];
                o[j] = a * (dz - k1 - xh * k2);
            }
        }
        reinterpret_cast<float4*>(out + row)[i] = make_float4(o[0], o[1], o[2], o[3]);
    } else {
        for (int j = 0; j < 4; ++j) {
            const int i = (blockIdx.x * DMVS_BLOCK + threadIdx.x) * 4 + j;
            if (i >= S) return;
            const float xv = x[row + i];
            if (!BWD) {
                const float y = fmaf(xv, a, b0);
                out[row + i] = relu ? fmaxf(y, 0.0f) : y;
            } else {
                const float xh = (xv - mu) * rs;
                const float dz = (relu && xh * g + bt <= 0.0f) ? 0.0f : dy[row + i];
                out[row + i] = a * (dz - k1 - xh * k2);
            }
        }
    }
}

inline int chunks_per_row(int S) { return (S + BN_CHUNK - 1) / BN_CHUNK; }
}  // namespace

static int64_t bn_ws_bytes(int B, int C, int S, int views) {
    return (int64_t)C * B * chunks_per_row(S) * (int64_t)sizeof(float2) + (int64_t)2 * views * C * (int64_t)sizeof(float);
}

extern "C" int dmvs_batchnorm_workspace_f32(int32_t B, int32_t C, int32_t S, int32_t views, int64_t* bytes) {
    if (!bytes || B <= 0 || C <= 0 || S <= 0 || views <= 0 || B % views) return DMVS_EINVAL;
    *bytes = bn_ws_bytes(B, C, S, views);
    return 0;
}

extern "C" int dmvs_batchnorm_train_fwd_f32(const float* x, const float* gamma, const float* beta, float* running_mean,
                                            float* running_var, float* y, float* save_mean, float* save_rstd,
                                            float* workspace, int64_t workspace_bytes, int32_t B, int32_t C, int32_t S,
                                            int32_t views, int32_t view_major, float momentum, float eps, int32_t act,
                                            void* stream) {
    if (!x || !gamma || !beta || !y || !save_mean || !save_rstd || !workspace || B <= 0 || C <= 0 || S <= 0) return DMVS_EINVAL;
    if (views <= 0 || B % views) return DMVS_EINVAL;
    if ((act != DMVS_ACT_NONE && act != DMVS_ACT_RELU) || (!running_mean) != (!running_var)) return DMVS_EINVAL;
    if (((uintptr_t)x | (uintptr_t)y) & 15) return DMVS_EINVAL;
    const int cpr = chunks_per_row(S), nchunk = B * cpr, rpv = B / views;
    if (workspace_bytes < bn_ws_bytes(B, C, S, views)) return DMVS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float2* part = reinterpret_cast<float2*>(workspace);
    hipLaunchKernelGGL((bn_partial_kernel<false>), dim3(nchunk, C), dim3(DMVS_BLOCK), 0, st, x, (const float*)nullptr, gamma, beta,
                       (const float*)nullptr, (const float*)nullptr, part, C, S, cpr, 0, views, rpv, view_major);
    hipLaunchKernelGGL((bn_finalize_kernel<false>), dim3(C), dim3(DMVS_BLOCK), 0, st, part, nchunk, cpr, (double)rpv * S, save_mean,
                       save_rstd, (float*)nullptr, (float*)nullptr, running_mean, running_var, momentum, eps, C, views, rpv,
                       view_major);
    hipLaunchKernelGGL((bn_apply_kernel<false>), dim3(dmvs_ceil_div((S + 3) / 4, DMVS_BLOCK), B * C), dim3(DMVS_BLOCK), 0, st, x,
                       (const float*)nullptr, gamma, beta, save_mean, save_rstd, (const float*)nullptr, (const float*)nullptr, y, C,
                       S, 0.0f, act == DMVS_ACT_RELU, views, rpv, view_major);
    return dmvs_launch_status();
}

extern "C" int dmvs_batchnorm_train_bwd_f32(const float* x, const float* dy, const float* gamma, const float* beta,
                                            const float* save_mean, const float* save_rstd, float* dx, float* dgamma,
                                            float* dbeta, float* workspace, int64_t workspace_bytes, int32_t B, int32_t C,
                                            int32_t S, int32_t views, int32_t view_major, int32_t act, void* stream) {
    if (!x || !dy || !gamma || !beta || !save_mean || !save_rstd || !dx || !dgamma || !dbeta || !workspace) return DMVS_EINVAL;
    if (B <= 0 || C <= 0 || S <= 0 || views <= 0 || B % views || (act != DMVS_ACT_NONE && act != DMVS_ACT_RELU)) return DMVS_EINVAL;
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) return DMVS_EINVAL;
    const int cpr = chunks_per_row(S), nchunk = B * cpr, rpv = B / views;
    if (workspace_bytes < bn_ws_bytes(B, C, S, views)) return DMVS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float2* part = reinterpret_cast<float2*>(workspace);
    float* dbeta_v = reinterpret_cast<float*>(part + (size_t)C * nchunk);      // per-view sums [views][C]
    float* dgamma_v = dbeta_v + (size_t)views * C;
    const int relu = act == DMVS_ACT_RELU;
    hipLaunchKernelGGL((bn_partial_kernel<true>), dim3(nchunk, C), dim3(DMVS_BLOCK), 0, st, x, dy, gamma, beta, save_mean, save_rstd,
                       part, C, S, cpr, relu, views, rpv, view_major);
    hipLaunchKernelGGL((bn_finalize_kernel<true>), dim3(C), dim3(DMVS_BLOCK), 0, st, part, nchunk, cpr, (double)rpv * S, dbeta_v,
                       dgamma_v, dbeta, dgamma, (float*)nullptr, (float*)nullptr, 0.0f, 0.0f, C, views, rpv, view_major);
    hipLaunchKernelGGL((bn_apply_kernel<true>), dim3(dmvs_ceil_div((S + 3) / 4, DMVS_BLOCK), B * C), dim3(DMVS_BLOCK), 0, st, x, dy,
                       gamma, beta, save_mean, save_rstd, dbeta_v, dgamma_v, dx, C, S, (float)(1.0 / ((double)rpv * S)), relu, views,
                       rpv, view_major);
    return dmvs_launch_status();
}

// ------------------------------------------------------------------------------------------
// Backward of  y = silu(((x - mean_g) * rstd_g * gamma + beta) * (scale + 1) + shift)   (Block.forward of the diffusion
// Unet, reference models/update.py:124-133, in training).  `stats` = the (sum, sum of squares) per (batch, group) that the
// forward accumulated (dmvs_groupnorm_silu_f32 or the convolution's fused epilogue).
//   du = dy * silu'(u);  per (b, c): S1 = sum du, S2 = sum du * xhat
//   dshift = S1, dscale = gamma*S2 + beta*S1, dbeta = sum_b s1*S1, dgamma = sum_b s1*S2            (s1 = scale + 1)
//   dx = rstd * (gamma*s1*du - m1 - xhat*m2),  m1 | m2 = sum_{c in group} gamma*s1*(S1 | S2) / n
// Three launches (row partial sums -> per-row fold + small outputs -> apply), no atomics.
namespace {

__device__ __forceinline__ void gn_row_consts(const double* stats, int b, int c, int C, int HW, int groups, float eps, float& mean,
                                              float& rstd) {
    const int cg = C / groups, g = c / cg;
    const double n = (double)cg * HW;
    double s1, s2;
    dmvs_gn_read_pair(&stats[2 * (b * groups + g)], s1, s2);
    const double m = s1 / n;
    double var = s2 / n - m * m;
    var = var < 0.0 ? 0.0 : var;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}

__device__ __forceinline__ float silu_grad(float u) {
    const float s = dmvs_sigmoid(u);
    return s * (1.0f + u * (1.0f - s));
}

constexpr int GN_CHUNK = 8192;

__global__ void __launch_bounds__(DMVS_BLOCK)
gn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                      const float* __restrict__ beta, const float* __restrict__ scale_shift, const double* __restrict__ stats,
                      float2* __restrict__ partial, int C, int HW, int groups, float eps, int chunks) {
    __shared__ float red[2 * DMVS_BLOCK / 64];
    const int bc = blockIdx.y, b = bc / C, c = bc % C;
    float mean, rstd;
    gn_row_consts(stats, b, c, C, HW, groups, eps, mean, rstd);
    const float s1 = scale_shift ? scale_shift[(long)b * 2 * C + c] + 1.0f : 1.0f, s0 = scale_shift ? scale_shift[(long)b * 2 * C + C + c] : 0.0f;
    const float g = gamma[c], bt = beta[c];
    const long base = (long)bc * HW;
    const int lo = blockIdx.x * GN_CHUNK, hi = min(lo + GN_CHUNK, HW);
    float a1 = 0.0f, a2 = 0.0f;
    for (int i = lo + threadIdx.x; i < hi; i += DMVS_BLOCK) {
        const float xh = (x[base + i] - mean) * rstd;
        const float u = fmaf(fmaf(xh, g, bt), s1, s0);
        const float du = dy[base + i] * silu_grad(u);
        a1 += du;
        a2 = fmaf(du, xh, a2);
    }
    const float2 t = block_sum2(a1, a2, red);
    if (threadIdx.x == 0) partial[(long)bc * chunks + blockIdx.x] = t;
}

// one workgroup per batch item: fold the chunk partials of every channel row, per-(b,c) outputs and the group means
__global__ void __launch_bounds__(DMVS_BLOCK)
gn_bwd_fold_kernel(const float2* __restrict__ partial, const float* __restrict__ gamma, const float* __restrict__ beta,
                   const float* __restrict__ scale_shift, float* __restrict__ rows /* [B][C][2] = S1, S2 */,
                   float* __restrict__ dscale_shift, int C, int chunks) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += DMVS_BLOCK) {
        double a = 0.0, q = 0.0;
        for (int k = 0; k < chunks; ++k) {
            const float2 p = partial[((long)b * C + c) * chunks + k];
            a += (double)p.x;
            q += (double)p.y;
        }
        rows[((long)b * C + c) * 2] = (float)a;
        rows[((long)b * C + c) * 2 + 1] = (float)q;
        if (dscale_shift) {
            dscale_shift[(long)b * 2 * C + c] = (float)(gamma[c] * q + beta[c] * a);       // d scale
            dscale_shift[(long)b * 2 * C + C + c] = (float)a;                              // d shift
        }
    }
}

// dgamma / dbeta: one thread per channel, sum over the batch
__global__ void gn_bwd_param_kernel(const float* __restrict__ rows, const float* __restrict__ scale_shift, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta, int B, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double dg = 0.0, db = 0.0;
    for (int b = 0; b < B; ++b) {
        const double s1 = scale_shift ? (double)scale_shift[(long)b * 2 * C + c] + 1.0 : 1.0;
        db += s1 * rows[((long)b * C + c) * 2];
        dg += s1 * rows[((long)b * C + c) * 2 + 1];
    }
    dgamma[c] = (float)dg;
    dbeta[c] = (float)db;
}

__global__ void __launch_bounds__(DMVS_BLOCK)
gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                    const float* __restrict__ beta, const float* __restrict__ scale_shift, const double* __restrict__ stats,
                    const float* __restrict__ rows, float* __restrict__ dx, int C, int HW, int groups, float eps) {
    const int bc = blockIdx.y, b = bc / C, c = bc % C;
    float mean, rstd;
    gn_row_consts(stats, b, c, C, HW, groups, eps, mean, rstd);
    const int cg = C / groups, g0 = (c / cg) * cg;
    float m1 = 0.0f, m2 = 0.0f;
    for (int cc = g0; cc < g0 + cg; ++cc) {
        const float s1c = scale_shift ? scale_shift[(long)b * 2 * C + cc] + 1.0f : 1.0f;
        const float w = gamma[cc] * s1c;
        m1 = fmaf(w, rows[((long)b * C + cc) * 2], m1);
        m2 = fmaf(w, rows[((long)b * C + cc) * 2 + 1], m2);
    }
    const float inv_n = 1.0f / ((float)cg * (float)HW);
    m1 *= inv_n;
    m2 *= inv_n;
    const float s1 = scale_shift ? scale_shift[(long)b * 2 * C + c] + 1.0f : 1.0f, s0 = scale_shift ? scale_shift[(long)b * 2 * C + C + c] : 0.0f;
    const float g = gamma[c], bt = beta[c], gs = g * s1;
    const long base = (long)bc * HW;
    for (long i = (long)blockIdx.x * DMVS_BLOCK + threadIdx.x; i < HW; i += (long)gridDim.x * DMVS_BLOCK) {
        const float xh = (x[base + i] - mean) * rstd;
        const float u = fmaf(fmaf(xh, g, bt), s1, s0);
        const float du = dy[base + i] * silu_grad(u);
        dx[base + i] = rstd * (gs * du - m1 - xh * m2);
    }
}

}  // namespace

extern "C" int dmvs_groupnorm_silu_bwd_workspace_f32(int32_t B, int32_t C, int32_t HW, int64_t* bytes) {
    if (!bytes || B <= 0 || C <= 0 || HW <= 0) return DMVS_EINVAL;
    const int chunks = (HW + GN_CHUNK - 1) / GN_CHUNK;
    *bytes = (int64_t)B * C * chunks * (int64_t)sizeof(float2) + (int64_t)B * C * 2 * (int64_t)sizeof(float);
    return 0;
}

extern "C" int dmvs_groupnorm_silu_bwd_f32(const float* x, const float* dy, const float* gamma, const float* beta,
                                           const float* scale_shift, const double* stats, float* dx, float* dgamma, float* dbeta,
                                           float* dscale_shift, float* workspace, int64_t workspace_bytes, int32_t B, int32_t C,
                                           int32_t HW, int32_t groups, float eps, void* stream) {
    if (!x || !dy || !gamma || !beta || !stats || !dx || !dgamma || !dbeta || !workspace) return DMVS_EINVAL;
    if (B <= 0 || C <= 0 || HW <= 0 || groups <= 0 || C % groups || (scale_shift && !dscale_shift)) return DMVS_EINVAL;
    const int chunks = (HW + GN_CHUNK - 1) / GN_CHUNK;
    int64_t need = 0;
    dmvs_groupnorm_silu_bwd_workspace_f32(B, C, HW, &need);
    if (workspace_bytes < need) return DMVS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float2* part = reinterpret_cast<float2*>(workspace);
    float* rows = reinterpret_cast<float*>(part + (size_t)B * C * chunks);
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(chunks, B * C), dim3(DMVS_BLOCK), 0, st, x, dy, gamma, beta, scale_shift, stats, part,
                       C, HW, groups, eps, chunks);
    hipLaunchKernelGGL(gn_bwd_fold_kernel, dim3(B), dim3(DMVS_BLOCK), 0, st, part, gamma, beta, scale_shift, rows, dscale_shift, C, chunks);
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3(dmvs_ceil_div(C, 64)), dim3(64), 0, st, rows, scale_shift, dgamma, dbeta, B, C);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(dmvs_ceil_div(HW, DMVS_BLOCK * 4), B * C), dim3(DMVS_BLOCK), 0, st, x, dy, gamma, beta,
                       scale_shift, stats, rows, dx, C, HW, groups, eps);
    return dmvs_launch_status();
}
