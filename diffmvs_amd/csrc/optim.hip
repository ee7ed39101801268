// Training-step tail over ONE flat fp32 parameter bucket (reference train.py:200-203, :321-326):
// global gradient norm (clip_grad_norm_), then AdamW with the clip coefficient and the data-parallel 1/world
// averaging folded in.  The bucket is what RCCL all-reduces (925 435 elements for CasDiffMVS = 3.7 MB):
// latency-bound, so the whole tail is two launches reading each array once.
#include "dmvs_common.h"

__global__ void __launch_bounds__(DMVS_BLOCK)
sumsq_kernel(const float* __restrict__ g, long n, double* __restrict__ out) {
    __shared__ double red[DMVS_BLOCK / 64];
    double acc = 0.0;
    const long n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long i = (long)blockIdx.x * DMVS_BLOCK + threadIdx.x; i < n4; i += (long)gridDim.x * DMVS_BLOCK) {
        const float4 v = g4[i];
        acc += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[(n4 << 2) + threadIdx.x];
        acc += (double)(v * v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < DMVS_BLOCK / 64; ++w) t += red[w];
        atomicAdd(out, t);
    }
}

extern "C" int dmvs_sumsq_f32(const float* g, int64_t n, double* out, void* stream) {
    if (!g || !out || n < 0 || ((uintptr_t)g & 15)) return DMVS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double), st);
    if (e != hipSuccess) return (int)e;
    if (n == 0) return 0;
    unsigned blocks = dmvs_ceil_div(n >> 2, DMVS_BLOCK * 4);
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(DMVS_BLOCK), 0, st, g, (long)n, out);
    return dmvs_launch_status();
}

// torch.optim.AdamW (decoupled weight decay, bias-corrected, no amsgrad), one step:
//   g' = g * grad_scale * min(1, max_norm / (grad_scale*sqrt(sumsq) + 1e-6))      (clip_grad_norm_, train.py:202)
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ void __launch_bounds__(DMVS_BLOCK)
adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
             float lr, float b1, float b2, float eps, float wd, float bc1, float rsqrt_bc2, float grad_scale,
             const double* __restrict__ sumsq, float max_norm) {
    float gs = grad_scale;
    if (sumsq) {
        const float norm = grad_scale * (float)sqrt(*sumsq);
        gs *= fminf(1.0f, max_norm / (norm + 1e-6f));
    }
    const float step = lr / bc1, decay = 1.0f - lr * wd;
    for (long i = (long)blockIdx.x * DMVS_BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * DMVS_BLOCK) {
        const float gi = g[i] * gs;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] * decay - step * (mi / (sqrtf(vi) * rsqrt_bc2 + eps));
    }
}

extern "C" int dmvs_adamw_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                                   const double* sumsq, float max_norm, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return DMVS_EINVAL;
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    unsigned blocks = dmvs_ceil_div(n, DMVS_BLOCK * 4);
    blocks = blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(DMVS_BLOCK), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, (long)n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)),
                       grad_scale, sumsq, max_norm);
    return dmvs_launch_status();
}
