// Stand-alone probe (no torch): does a gfx950 SIMD overlap the fp32 MFMAs of its waves with their (and other waves') VALU work, and
// which shader clock does the part hold under a sustained fp32-MFMA load?
//
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/calib/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
//
// Why: SQ counters of the 16 -> 16 convolution (profiles/r3_sq_pmc_conv2d_16to16_n96.txt) show the matrix pipe 58 % busy with waves
// "issue-stalled" 56 % of their life, and across the convolution shapes time ~ MFMA cycles + 4 x VALU instructions fits better than
// max(...) does.  This measures it directly: every wave loops over {4 independent v_mfma_f32_16x16x4_f32,
// M v_fma_f32 on 8 independent chains}, for M = 0..64 VALU per 4 MFMAs (128 matrix cycles) and 1 / 2 / 4 / 5 waves per SIMD.
// Output: one JSON object per line (ns per loop trip per SIMD, and the clock derived from s_memtime vs the 100 MHz real-time counter).
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

template <int NMFMA, int M>
__global__ void __launch_bounds__(256) mix_kernel(float* out, long long* clocks, int iters) {
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const float a = 1e-3f * (float)(threadIdx.x & 15), b = 1.0f + 1e-6f * (float)(threadIdx.x >> 4);
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = (float)(threadIdx.x + k);
    const long long c0 = (long long)__builtin_readcyclecounter();       // s_memtime
    const long long r0 = (long long)wall_clock64();                       // s_memrealtime: 100 MHz
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NMFMA; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < M; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(b), "v"(a));
    }
    const long long c1 = (long long)__builtin_readcyclecounter();
    const long long r1 = (long long)wall_clock64();
    float s = 0.0f;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    for (int k = 0; k < 8; ++k) s += v[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clocks[0] = c1 - c0;
        clocks[1] = r1 - r0;
    }
}

// half of the waves of a workgroup issue only MFMAs, the other half only VALU: do two WAVES of one SIMD overlap the two pipes?
template <int NMFMA, int M>
__global__ void __launch_bounds__(512) split_kernel(float* out, int iters) {
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const float a = 1e-3f * (float)(threadIdx.x & 15), b = 1.0f + 1e-6f * (float)(threadIdx.x >> 4);
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = (float)(threadIdx.x + k);
    if ((threadIdx.x >> 8) == 0) {          // waves 0..3: one per SIMD, matrix work only
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < NMFMA; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0);
    } else {                                // waves 4..7: one per SIMD, vector work only
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < M; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(b), "v"(a));
    }
    float s = 0.0f;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    for (int k = 0; k < 8; ++k) s += v[k];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// matrix instruction shapes side by side: NACC independent accumulators, NM instructions per loop trip, nothing else in the loop.
//   KIND 0: v_mfma_f32_16x16x4_f32 (8 passes = 32 cycles nominal)      KIND 1: v_mfma_f32_32x32x2_f32 (16 passes = 64 cycles)
//   KIND 2: v_mfma_f32_16x16x32_bf16 (the bf16 form of conv2d.hip)      KIND 3: v_mfma_f32_32x32x16_bf16
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND, int NACC, int NM>
__global__ void __launch_bounds__(256) shape_kernel(float* out, int iters) {
    const float a = 1e-3f * (float)(threadIdx.x & 15), b = 1.0f + 1e-6f * (float)(threadIdx.x >> 4);
    bf16x8 ha, hb;
    for (int j = 0; j < 8; ++j) {
        ha[j] = (__bf16)(a + (float)j);
        hb[j] = (__bf16)(b - (float)j);
    }
    float s = 0.0f;
    if constexpr (KIND == 0 || KIND == 2) {
        f32x4 acc[NACC];
        for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                if constexpr (KIND == 0) acc[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j % NACC], 0, 0, 0);
                else acc[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[j % NACC], 0, 0, 0);
            }
        }
        for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else {
        f32x16 acc[NACC];
        for (int j = 0; j < NACC; ++j)
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                if constexpr (KIND == 1) acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j % NACC], 0, 0, 0);
                else acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, acc[j % NACC], 0, 0, 0);
            }
        }
        for (int j = 0; j < NACC; ++j)
            for (int r = 0; r < 16; ++r) s += acc[j][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float* g_out;
static long long* g_clk;
static int g_cus;

template <int KIND, int NACC, int NM>
static void run_shape(int waves_per_simd, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((shape_kernel<KIND, NACC, NM>), dim3(g_cus * waves_per_simd), dim3(256), 0, 0, g_out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((shape_kernel<KIND, NACC, NM>), dim3(g_cus * waves_per_simd), dim3(256), 0, 0, g_out, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    static const char* names[4] = {"v_mfma_f32_16x16x4_f32", "v_mfma_f32_32x32x2_f32", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x16_bf16"};
    static const double flops[4] = {2.0 * 16 * 16 * 4, 2.0 * 32 * 32 * 2, 2.0 * 16 * 16 * 32, 2.0 * 32 * 32 * 16};
    const double n_instr = (double)iters * NM * waves_per_simd;                     // per SIMD
    const double tf = flops[KIND] * n_instr * 4.0 * g_cus / ((double)ms * 1e-3) / 1e12;
    printf("{\"probe\": \"shape\", \"instr\": \"%s\", \"accumulators\": %d, \"per_trip\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, "
           "\"ns_per_instr_per_simd\": %.3f, \"TFLOPs\": %.1f}\n", names[KIND], NACC, NM, waves_per_simd, ms, (double)ms * 1e6 / n_instr, tf);
    fflush(stdout);
}

template <int NMFMA, int M>
static void run_mix(int waves_per_simd, int iters) {
    const int grid = g_cus * waves_per_simd;        // 256-thread workgroups = one wave per SIMD each
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((mix_kernel<NMFMA, M>), dim3(grid), dim3(256), 0, 0, g_out, g_clk, iters);      // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((mix_kernel<NMFMA, M>), dim3(grid), dim3(256), 0, 0, g_out, g_clk, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long clk[2];
    CHECK(hipMemcpy(clk, g_clk, sizeof(clk), hipMemcpyDeviceToHost));
    const double ns_trip = (double)ms * 1e6 / (double)iters;          // per loop trip of a SIMD's `waves_per_simd` waves
    const double mhz = clk[1] > 0 ? (double)clk[0] / ((double)clk[1] / 100.0) : 0.0;
    printf("{\"probe\": \"mix\", \"mfma_per_trip\": %d, \"valu_per_trip\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"ns_per_trip\": %.2f, "
           "\"ns_per_trip_per_wave\": %.2f, \"memtime_ticks\": %lld, \"realtime_ticks_100mhz\": %lld, \"memtime_mhz\": %.1f}\n",
           NMFMA, M, waves_per_simd, ms, ns_trip, ns_trip / waves_per_simd, clk[0], clk[1], mhz);
    fflush(stdout);
}

template <int NMFMA, int M>
static void run_split(int wgs_per_cu, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((split_kernel<NMFMA, M>), dim3(g_cus * wgs_per_cu), dim3(512), 0, 0, g_out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((split_kernel<NMFMA, M>), dim3(g_cus * wgs_per_cu), dim3(512), 0, 0, g_out, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"probe\": \"split\", \"mfma_per_trip\": %d, \"valu_per_trip\": %d, \"wave_pairs_per_simd\": %d, \"ms\": %.4f, \"ns_per_trip\": %.2f}\n", NMFMA, M,
           wgs_per_cu, ms, (double)ms * 1e6 / (double)iters);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    printf("{\"probe\": \"device\", \"name\": \"%s\", \"cus\": %d, \"clock_rate_khz\": %d}\n", prop.name, g_cus, prop.clockRate);
    CHECK(hipMalloc(&g_out, (size_t)g_cus * 8 * 512 * sizeof(float)));
    CHECK(hipMalloc(&g_clk, 2 * sizeof(long long)));
    const int it = 20000;
    if (getenv("PROBE_SHAPES")) {      // second run: matrix instruction shapes only
        for (int w : {1, 2, 4, 5}) run_shape<0, 4, 16>(w, it);
        for (int w : {1, 4}) run_shape<0, 8, 16>(w, it);
        for (int w : {1, 2, 4}) run_shape<1, 2, 8>(w, it);
        for (int w : {1, 2, 4}) run_shape<1, 4, 8>(w, it);
        for (int w : {1, 2, 4}) run_shape<2, 4, 16>(w, it);
        for (int w : {1, 2, 4}) run_shape<3, 2, 8>(w, it);
        return 0;
    }
    // 1. the two pipes alone
    for (int w : {1, 2, 4}) run_mix<4, 0>(w, it);
    for (int w : {1, 2, 4}) run_mix<0, 32>(w, it);
    // 2. mixed in one wave: 4 MFMAs (128 matrix cycles) + M VALU (4 cycles each)
    for (int w : {1, 2, 4, 5}) {
        run_mix<4, 8>(w, it);
        run_mix<4, 16>(w, it);
        run_mix<4, 32>(w, it);
        run_mix<4, 64>(w, it);
    }
    // 3. the two kinds of work in different waves of the same SIMD
    for (int w : {1, 2}) {
        run_split<4, 16>(w, it);
        run_split<4, 32>(w, it);
        run_split<4, 64>(w, it);
    }
    // 4. the clock the part settles at under ~2 s of back-to-back fp32 MFMAs (5 waves per SIMD)
    for (int rep = 0; rep < 16; ++rep) run_mix<4, 0>(5, 400000);
    return 0;
}
