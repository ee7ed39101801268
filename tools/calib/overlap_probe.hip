// Stand-alone probe (no torch): can a gfx950 CU overlap fp32-MFMA work with HBM streaming at all -- in hardware, before any kernel structure?
//
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/calib/overlap_probe.hip -o /tmp/overlap_probe && /tmp/overlap_probe
//
// Why (round 6): every matrix kernel of the step fits  time ~ (flops / fp32-MFMA peak) + (algorithmic bytes / ~4.5 TB/s)  -- the `sum`
// column of bench.py --conv-table is 0.83-1.01 for every large layer -- although the kernels double-buffer their LDS-DMA staging under
// the MFMAs, and no tile shape, staging form, stagger or occupancy change moved that.  Either the part cannot run the two at full rate
// at once (clock / power management, a shared issue or data path), or the kernels fail to.  This decides which: workgroups of 8 waves
// whose waves 0-3 (one per SIMD) issue only v_mfma_f32_16x16x4_f32 and whose waves 4-7 only stream memory, with the two amounts of
// work sized to take the same time alone; then both at once.  Three memory forms: plain copy (global_load_dwordx4 -> VGPR ->
// global_store_dwordx4), read-only through LDS-DMA (global_load_lds_dwordx4 into LDS, what the convolutions' staging issues), and
// write-only.  Output: one JSON object per line with the time of each part alone, of both together, and the shader clock
// (s_memtime against the 100 MHz real-time counter) seen by an MFMA wave in each run.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MEM: 0 copy, 1 LDS-DMA read, 2 write-only.  do_mfma / do_mem: which halves of the workgroup work (the other half exits at once).
// COMP: 0 v_mfma_f32_16x16x4_f32 (16 per trip), 1 v_fma_f32 (128 per trip: the same 512 issue cycles), 2 v_mfma_f32_16x16x32_bf16 (32 per trip).
// SPLIT: the two kinds of waves live in DIFFERENT workgroups (see the kernel): is the interference inside a CU, inside an XCD, or chip-wide?
template <int MEM, int COMP = 0, int SPLIT = 0>
__global__ void __launch_bounds__(512) overlap_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, float* out, long long* clocks, int mfma_iters,
                                                      long n16, int do_mfma, int do_mem) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 64 * 4 * 8];      // 4 memory waves x 8 slots of 64 x 16 bytes
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // SPLIT 1: block parity = XCD parity (blocks go round-robin over the 8 XCDs): even XCDs compute, odd XCDs stream.
    // SPLIT 2: parity of blockIdx.x >> 3 = alternate workgroups of ONE XCD (different CUs at one workgroup per CU)
    const unsigned role = SPLIT == 2 ? (blockIdx.x >> 3) & 1u : blockIdx.x & 1u, half_id = SPLIT == 2 ? ((blockIdx.x >> 4) << 3) | (blockIdx.x & 7u) : blockIdx.x >> 1;
    if (SPLIT) {
        if (role) do_mfma = 0;
        else do_mem = 0;
    }
    if (wave < 4) {
        if (!do_mfma) return;
        f32x4 acc[4];
        for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const float a = 1e-3f * (float)(threadIdx.x & 15), b = 1.0f + 1e-6f * (float)(threadIdx.x >> 4);
        float v[8];
        for (int k = 0; k < 8; ++k) v[k] = (float)(threadIdx.x + k);
        bf16x8 ha, hb;
        for (int k = 0; k < 8; ++k) { ha[k] = (__bf16)(a + k); hb[k] = (__bf16)(b - k); }
        const long long c0 = (long long)__builtin_readcyclecounter();
        const long long r0 = (long long)wall_clock64();
        for (int i = 0; i < mfma_iters; ++i) {
            if constexpr (COMP == 0) {
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0);
            } else if constexpr (COMP == 1) {
#pragma unroll
                for (int k = 0; k < 128; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(b), "v"(a));
            } else {
#pragma unroll
                for (int j = 0; j < 32; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[j & 3], 0, 0, 0);
            }
        }
        for (int k = 0; k < 8; ++k) acc[0][0] += v[k];
        const long long c1 = (long long)__builtin_readcyclecounter();
        const long long r1 = (long long)wall_clock64();
        float s = 0.0f;
        for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
        if (threadIdx.x == 0 && blockIdx.x == 0) {
            clocks[0] = c1 - c0;
            clocks[1] = r1 - r0;
        }
        return;
    }
    if (!do_mem) return;
    // memory waves: global wave index over the grid, each trip moves 8 x 64 x 16 bytes = 8 KB per wave
    const long gw = SPLIT ? (long)half_id * 4 + (wave - 4) : (long)blockIdx.x * 4 + (wave - 4), nw = (long)(SPLIT ? gridDim.x / 2 : gridDim.x) * 4;
    for (long base = gw * 512; base + 512 <= n16; base += nw * 512) {
        if constexpr (MEM == 0) {
            f32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(src + base + k * 64 + lane);
#pragma unroll
            for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(v[k], dst + base + k * 64 + lane);
        } else if constexpr (MEM == 1) {
            float* my = lds + (wave - 4) * (64 * 4 * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src + base + k * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(my + k * 256), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const f32x4 v = {1.0f, 2.0f, 3.0f, (float)lane};
#pragma unroll
            for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(v, dst + base + k * 64 + lane);
        }
    }
    if constexpr (MEM == 1) {
        if (lane == 0 && lds[(wave - 4) * 2048] == 123.456f) out[0] = 1.0f;      // keep the LDS object alive
    }
}

template <int MEM, int COMP = 0, int SPLIT = 0>
static float run(const f32x4* src, f32x4* dst, float* out, long long* clocks, int wgs, int mfma_iters, long n16, int do_mfma, int do_mem, double* ghz) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((overlap_kernel<MEM, COMP, SPLIT>), dim3(wgs), dim3(512), 0, 0, src, dst, out, clocks, mfma_iters, n16, do_mfma, do_mem);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    long long hc[2] = {0, 0};
    if (do_mfma) {
        CHECK(hipMemcpy(hc, clocks, sizeof(hc), hipMemcpyDeviceToHost));
        *ghz = hc[1] > 0 ? (double)hc[0] / ((double)hc[1] * 10.0) : 0.0;      // shader clock: s_memtime ticks per 10 ns tick of the real-time counter
    } else
        *ghz = 0.0;
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return best;
}

template <int MEM, int COMP = 0, int SPLIT = 0>
static void experiment(const char* name, const f32x4* src, f32x4* dst, float* out, long long* clocks, int wg_per_cu, long bytes, int mfma_iters) {
    const int wgs = 256 * wg_per_cu;
    const long n16 = bytes / 16;
    double g0, g1, g2;
    const float t_mfma = run<MEM, COMP, SPLIT>(src, dst, out, clocks, wgs, mfma_iters, n16, 1, 0, &g0);
    const float t_mem = run<MEM, COMP, SPLIT>(src, dst, out, clocks, wgs, mfma_iters, n16, 0, 1, &g1);
    const float t_both = run<MEM, COMP, SPLIT>(src, dst, out, clocks, wgs, mfma_iters, n16, 1, 1, &g2);
    const double moved = (MEM == 0 ? 2.0 : 1.0) * (double)bytes;
    const double cwaves = (double)(SPLIT ? wgs / 2 : wgs) * 4.0;
    const double tf = cwaves * mfma_iters * (COMP == 0 ? 16.0 * 2048.0 : (COMP == 1 ? 128.0 * 128.0 : 32.0 * 16384.0)) / (t_mfma * 1e-3) / 1e12;
    printf("{\"probe\": \"overlap\", \"compute\": \"%s\", \"placement\": \"%s\", \"memory\": \"%s\", \"wg_per_cu\": %d, \"compute_alone_ms\": %.4f, \"compute_alone_TFs\": %.1f, "
           "\"mem_alone_ms\": %.4f, \"mem_alone_TBs\": %.2f, \"both_ms\": %.4f, \"sum_ms\": %.4f, \"max_ms\": %.4f, \"both_over_max\": %.3f, \"both_over_sum\": %.3f, "
           "\"shader_GHz_compute_alone\": %.3f, \"shader_GHz_both\": %.3f}\n",
           COMP == 0 ? "v_mfma_f32_16x16x4_f32" : (COMP == 1 ? "v_fma_f32" : "v_mfma_f32_16x16x32_bf16"),
           SPLIT == 1 ? "compute waves on the even XCDs, memory waves on the odd XCDs" : SPLIT == 2 ? "compute and memory waves in alternate workgroups of every XCD (different CUs)" : "4 compute + 4 memory waves per workgroup (one of each per SIMD)", name, wg_per_cu,
           t_mfma, tf, t_mem, moved / (t_mem * 1e-3) / 1e12, t_both, t_mfma + t_mem, t_mfma > t_mem ? t_mfma : t_mem,
           t_both / (t_mfma > t_mem ? t_mfma : t_mem), t_both / (t_mfma + t_mem), g0, g2);
    fflush(stdout);
}

int main() {
    const long bytes = 6L << 30;      // per direction
    f32x4 *src, *dst;
    float* out;
    long long* clocks;
    CHECK(hipMalloc(&src, bytes));
    CHECK(hipMalloc(&dst, bytes));
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    CHECK(hipMalloc(&clocks, 2 * sizeof(long long)));
    CHECK(hipMemset(src, 0, bytes));
    CHECK(hipMemset(dst, 0, bytes));
    for (int wg_per_cu : {1, 2, 4}) {
        // compute work sized to ~the memory time: one compute wave per SIMD per workgroup; a trip = 512 issue cycles of the fp32 forms
        const int it_copy = (int)(6.4e6 / 512.0 / wg_per_cu);      // copy of 6 + 6 GB at ~4.5 TB/s ~ 2.7 ms ~ 6.4 M cycles per SIMD
        const int it_read = (int)(3.0e6 / 512.0 / wg_per_cu);
        experiment<0>("copy (load -> VGPR -> store), 6 GB each way", src, dst, out, clocks, wg_per_cu, bytes, it_copy);
        experiment<1>("read only, LDS-DMA (global_load_lds 16 B), 6 GB", src, dst, out, clocks, wg_per_cu, bytes, it_read);
        experiment<2>("write only, 6 GB", src, dst, out, clocks, wg_per_cu, bytes, it_read);
    }
    {
        const int it_copy = (int)(6.4e6 / 512.0 / 2), it_read = (int)(3.0e6 / 512.0 / 2);
        // other instruction kinds in the compute waves
        experiment<0, 1>("copy (load -> VGPR -> store), 6 GB each way", src, dst, out, clocks, 2, bytes, it_copy);
        experiment<1, 1>("read only, LDS-DMA (global_load_lds 16 B), 6 GB", src, dst, out, clocks, 2, bytes, it_read);
        experiment<0, 2>("copy (load -> VGPR -> store), 6 GB each way", src, dst, out, clocks, 2, bytes, it_copy);
        experiment<1, 2>("read only, LDS-DMA (global_load_lds 16 B), 6 GB", src, dst, out, clocks, 2, bytes, it_read);
        // compute and memory waves on different CUs: 256 blocks = one per CU, even blocks compute, odd blocks stream (twice the work per wave)
        experiment<0, 0, 1>("copy (load -> VGPR -> store), 6 GB each way", src, dst, out, clocks, 1, bytes, 2 * (int)(6.4e6 / 512.0));
        experiment<1, 0, 1>("read only, LDS-DMA (global_load_lds 16 B), 6 GB", src, dst, out, clocks, 1, bytes, 2 * (int)(3.0e6 / 512.0));
        experiment<0, 0, 2>("copy (load -> VGPR -> store), 6 GB each way", src, dst, out, clocks, 1, bytes, 2 * (int)(6.4e6 / 512.0));
        experiment<1, 0, 2>("read only, LDS-DMA (global_load_lds 16 B), 6 GB", src, dst, out, clocks, 1, bytes, 2 * (int)(3.0e6 / 512.0));
    }
    return 0;
}
