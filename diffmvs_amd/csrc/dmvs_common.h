// Shared device helpers for the gfx950 kernels of libdmvs_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dmvs.h"

#define DMVS_BLOCK 256

static inline int dmvs_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

static inline unsigned dmvs_ceil_div(long a, long b) { return (unsigned)((a + b - 1) / b); }

__device__ __forceinline__ float dmvs_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

__device__ __forceinline__ float dmvs_act(float v, int act) {
    switch (act) {
        case DMVS_ACT_RELU: return fmaxf(v, 0.0f);
        case DMVS_ACT_SIGMOID: return dmvs_sigmoid(v);
        case DMVS_ACT_TANH: return tanhf(v);
        case DMVS_ACT_SILU: return v * dmvs_sigmoid(v);
        default: return v;
    }
}

// disp_to_depth (reference models/module.py:220-227): normalised inverse depth -> metric depth
__device__ __forceinline__ float dmvs_disp_to_depth(float nd, float disp_min, float disp_max) {
    float scaled = disp_min + (disp_max - disp_min) * nd;
    scaled = fmaxf(scaled, 1e-6f);
    return 1.0f / scaled;
}
