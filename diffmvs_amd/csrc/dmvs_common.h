// Shared device helpers for the gfx950 kernels of libdmvs_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dmvs.h"

#define DMVS_BLOCK 256

// Scheduling fence for unrolled per-hypothesis code: `var` (an input of the NEXT step) is redefined by an empty asm that
// consumes `dep` (a result of THIS step), so the compiler cannot hoist the next step's address/projection arithmetic
// above this step and keep all of it live at once.  (The host emulation predefines it as a no-op.)
#ifndef DMVS_ORDER_AFTER
#define DMVS_ORDER_AFTER(var, dep) asm volatile("" : "+v"(var) : "v"(dep))
#endif

// Workgroup barrier that publishes LDS writes (ds_write: lgkmcnt) but does NOT drain this wave's outstanding vector-memory
// operations, so that an LDS-DMA prefetch stays in flight across it.  Only for barriers whose producers are ds_writes.
// (The host emulation predefines it as a plain barrier.)
#ifndef DMVS_LDS_BARRIER
#define DMVS_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// Workgroup barrier that PUBLISHES LDS-DMA DATA (global_load_lds_*): every wave first drains its own outstanding vector-memory
// operations (s_waitcnt vmcnt(0): the LDS-DMA it issued has written LDS), then the barrier.  The wait is EXPLICIT on purpose.
// Rounds 2-4 relied on "__syncthreads() waits vmcnt(0)" -- it does not: on gfx950 hipcc emits no vmcnt wait for the
// workgroup-scope fence of __syncthreads() (the waves of a workgroup share a CU), and its wait-count pass does not tie an
// LDS-DMA to the later ds_reads of the staged bytes in these kernels; where a vmcnt(0) did sit in front of such a barrier it
// was there for an unrelated register dependency.  The round-4 stem kernel lost that accident: its tile loop had its only
// vmcnt(0) in front of the loop, so from the second tile of a workgroup on, conv0.0 could read a halo that was still landing
// -- the run-to-run differences the round-4 driver run found (tools/determinism.py, tools/isa_dma_audit.py).
// EVERY barrier that follows an LDS-DMA whose data is read after it must be this one.  (Plain barrier under the host emulation.)
#ifndef DMVS_DMA_BARRIER
#define DMVS_DMA_BARRIER()                               \
    do {                                                 \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        __syncthreads();                                 \
    } while (0)
#endif

static inline int dmvs_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// Workgroups of `kernel` (256 threads, static LDS only) the whole chip holds at once: the grid of a resident, tile-walking
// launch.  The occupancy query over-reports by one workgroup per CU for SGPR-heavy kernels at 7-8 per CU
// (MI355X_MICROARCH.md, residency) -- and a surplus workgroup would start only when a resident one has walked ALL its tiles
// (a tail as long as the launch) -- so the count is capped at 6 per CU, below that band.
#ifdef DMVS_HOST_EMULATION
static inline int dmvs_resident_workgroups(const void*) { return 2; }       // (so that the CPU tests walk several tiles per workgroup)
#else
static inline int dmvs_resident_workgroups(const void* kernel) {
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, DMVS_BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > 6) per_cu = 6;
    return per_cu * prop.multiProcessorCount;
}
#endif

static inline unsigned dmvs_ceil_div(long a, long b) { return (unsigned)((a + b - 1) / b); }
__device__ __forceinline__ unsigned dmvs_ceil_div_dev(long a, long b) { return (unsigned)((a + b - 1) / b); }

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

// GroupNorm statistics (per-(batch item, group) sum and sum of squares) are accumulated from many workgroups.  Floating-point
// atomics would make the result depend on the arrival order; the 8-byte slots therefore hold FIXED-POINT integers (2^-16
// units): integer addition is associative, so the statistics -- and with them the whole forward -- are bit-reproducible
// run to run.  Resolution 1.5e-5 per contribution (a workgroup's partial sum, magnitude 1e2..1e6).
// Magnitude contract: statistics up to 3.5e13 (|fixed| < 2^61; a random-weight network's pre-normalisation planes reach
// 1e12).  A contribution that is not finite, or whose magnitude reaches 2^44 (1.8e13), POISONS the (sum, sum of squares) PAIR:
// bits 61 and 62 of the sum-of-squares slot -- which only ever receives non-negative adds -- are set with an atomic OR, and
// dmvs_gn_read_pair returns NaN for both statistics when that slot is at or beyond 2^61 (or the sum is beyond +-2^61).  The mark is
// sticky: an OR cannot be undone by later in-range adds (round 3 overwrote the slot with a sentinel VALUE, which enough later adds
// could carry back into range -- ADVICE round 3); only further in-range contributions summing past 2^63 units (1.4e14, four times
// the contract) could wrap it.  A NaN / Inf / out-of-range activation makes its group's outputs NaN, like the floating-point
// statistics of the reference would, instead of finite garbage or a silent wrap.
// Callers keep treating the buffer as opaque zero-initialised 8-byte slots (all-zero bits = 0 in either reading).
#define DMVS_GN_FIX 65536.0
#define DMVS_GN_POISON 0x6000000000000000ull
// pair = &stats[2 * (b * groups + g)]; which = 0: sum, 1: sum of squares
__device__ __forceinline__ void dmvs_gn_accumulate(double* pair, int which, double v) {
    unsigned long long* p = reinterpret_cast<unsigned long long*>(pair);
    if (!(fabs(v) < 17592186044416.0)) {         // 2^44; false for NaN as well
#ifdef DMVS_HOST_EMULATION
        __atomic_fetch_or(p + 1, DMVS_GN_POISON, __ATOMIC_RELAXED);
#else
        atomicOr(p + 1, DMVS_GN_POISON);
#endif
        return;
    }
    atomicAdd(p + which, (unsigned long long)(long long)llrint(v * DMVS_GN_FIX));
}
__device__ __forceinline__ void dmvs_gn_read_pair(const double* pair, double& sum, double& sumsq) {
    const long long f0 = reinterpret_cast<const long long*>(pair)[0];
    const unsigned long long f1 = reinterpret_cast<const unsigned long long*>(pair)[1];
    if (f1 >= (1ull << 61) || f0 >= (1ll << 61) || f0 <= -(1ll << 61)) {
        sum = sumsq = __builtin_nan("");
        return;
    }
    sum = (double)f0 * (1.0 / DMVS_GN_FIX);
    sumsq = (double)(long long)f1 * (1.0 / DMVS_GN_FIX);
}

// 16-bit feature storage (DMVS_DTYPE_BF16 / DMVS_DTYPE_F16): round-to-nearest-even conversions; arithmetic stays fp32.
__device__ __forceinline__ uint16_t dmvs_f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);      // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float dmvs_bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
#ifdef DMVS_HOST_EMULATION
__device__ __forceinline__ uint16_t dmvs_f32_to_f16(float f) { return hipemu_f32_to_f16(f); }
__device__ __forceinline__ float dmvs_f16_to_f32(uint16_t h) { return hipemu_f16_to_f32(h); }
#else
__device__ __forceinline__ uint16_t dmvs_f32_to_f16(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ float dmvs_f16_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
#endif
template <int DT> __device__ __forceinline__ uint16_t dmvs_to_x16(float f) { return DT == DMVS_DTYPE_BF16 ? dmvs_f32_to_bf16(f) : dmvs_f32_to_f16(f); }
template <int DT> __device__ __forceinline__ float dmvs_from_x16(uint16_t h) { return DT == DMVS_DTYPE_BF16 ? dmvs_bf16_to_f32(h) : dmvs_f16_to_f32(h); }

// disp_to_depth (reference models/module.py:220-227): normalised inverse depth -> metric depth
__device__ __forceinline__ float dmvs_disp_to_depth(float nd, float disp_min, float disp_max) {
    float scaled = disp_min + (disp_max - disp_min) * nd;
    scaled = fmaxf(scaled, 1e-6f);
    return 1.0f / scaled;
}

// Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, each with a private
// 4 MiB L2).  Remap so that every XCD walks one contiguous eighth of the logical blocks: neighbouring
// tiles share halo rows / epipolar bands, which then stay resident in that XCD's L2 instead of being
// re-fetched by all eight.  Bijective for any grid size (cdna_hip_programming.md T1).
__device__ __forceinline__ unsigned dmvs_xcd_contiguous_block(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// The in-between mapping (round 6): XCD x takes GROUPS of g consecutive logical blocks -- blocks b, b + 8, b + 16, ... of one XCD are the
// g members of one group, then of the group 8 further on -- so that x-adjacent tiles, whose rows are halves of the same 128-byte lines
// (a 16-pixel fp32 tile row is 64 bytes) and whose halo columns overlap, meet in ONE L2, while the eight XCDs still sweep the same
// region of the tensor at the same time (the HBM channel spread of the plain round-robin order, which the contiguous-eighths
// mapping above gives up).  Bijective: the last nblk % (8 g) blocks keep their own index.
__device__ __forceinline__ unsigned dmvs_xcd_grouped_block(unsigned bid, unsigned nblk, unsigned g) {
    const unsigned span = 8u * g, full = nblk / span * span;
    if (bid >= full) return bid;
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    return ((idx / g) * 8u + xcd) * g + idx % g;
}
