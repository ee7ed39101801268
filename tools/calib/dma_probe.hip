// Stand-alone probe (no torch) of the LDS-DMA instruction forms the convolution kernels could use next:
//
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/calib/dma_probe.hip -o /tmp/dma_probe && timeout 60 /tmp/dma_probe
//
// (written at the end of round 3 without a GPU left to run it on; a form the hardware rejects at 4-byte alignment would show up as a
// memory-access fault of this process -- run it early in a round, under `timeout`)
//
// 1. Correctness of global_load_lds with 4 / 12 / 16 bytes per lane when the GLOBAL address is only 4-byte aligned (the kernels
//    currently issue the 16-byte form with 16-byte aligned sources only, and pay for that with slack columns in LDS: a 3x3 halo
//    row of 18 floats is staged as a 24-float aligned cover.  If the 12-byte form works at 4-byte alignment, the row is exactly
//    6 pieces with no slack; if the 16-byte form does, 20 floats = 5 pieces).
// 2. Cost per wave-level instruction of each width (the texture path's issue rate: DESIGN.md 4.1 finding 3 measured ~55-60 cycles
//    per instruction whatever the width), global_load_lds vs buffer_load ... lds (scalar base + 32-bit lane offset, out-of-range
//    lanes return zeros -- which would replace the exec masks and the padding pass of conv2d_tiled.h).
// Output: one JSON object per line.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

// every lane of the 4 waves stages SIZE bytes from src + misalign_floats + lane * (SIZE / 4), REPS times into a 64 KB LDS ring, then
// the block copies the first 256 pieces back out so that the host can compare them with the source
template <int SIZE, bool BUFFER>
__global__ void __launch_bounds__(256) dma_kernel(const float* src, float* out, int misalign_floats, int reps, int n_floats) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int W = SIZE / 4;
    const float* p = src + (size_t)blockIdx.x * 4096 + misalign_floats;
    for (int r = 0; r < reps; ++r) {
        float* dst = lds + ((r & 7) * 256 + wave * 64) * W;            // wave-uniform base: the instruction adds lane * SIZE
        // (the builtins want the size as a literal)
        if constexpr (BUFFER) {
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n_floats * 4, 0x00020000);
            const int voff = (int)((blockIdx.x * 4096 + misalign_floats + (wave * 64 + lane) * W) * 4);
            if constexpr (SIZE == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDSP(dst), 4, voff, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDSP(dst), 16, voff, 0, 0, 0);
        } else {
            if constexpr (SIZE == 4) __builtin_amdgcn_global_load_lds(p + (wave * 64 + lane) * W, LDSP(dst), 4, 0, 0);
            else if constexpr (SIZE == 12) __builtin_amdgcn_global_load_lds(p + (wave * 64 + lane) * W, LDSP(dst), 12, 0, 0);
            else __builtin_amdgcn_global_load_lds(p + (wave * 64 + lane) * W, LDSP(dst), 16, 0, 0);
        }
    }
    __syncthreads();
    if (blockIdx.x == 0)
        for (int i = tid; i < 256 * W; i += 256) out[i] = lds[i];
}

// lanes past the end of the buffer: a buffer load returns zeros for them (the zero-padding idea)
__global__ void __launch_bounds__(64) oob_kernel(const float* src, float* out, int n_floats) {
    __shared__ __attribute__((aligned(16))) float lds[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) lds[i] = -1.0f;
    __syncthreads();
    const int voff = (lane & 1) ? (int)0x7ffffff0 : lane * 16;       // odd lanes far outside the buffer
    __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n_floats * 4, 0x00020000), LDSP(lds), 16, voff, 0, 0, 0);
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}

template <int SIZE, bool BUFFER>
static void run(const float* d_src, float* d_out, const std::vector<float>& h_src, int misalign, int n_floats) {
    constexpr int W = SIZE / 4;
    const int blocks = 256 * 4, reps = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((dma_kernel<SIZE, BUFFER>), dim3(blocks), dim3(256), 0, 0, d_src, d_out, misalign, 8, n_floats);
    CHECK(hipDeviceSynchronize());
    std::vector<float> got(256 * W);
    CHECK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256 * W; ++i) bad += got[i] != h_src[misalign + i];
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((dma_kernel<SIZE, BUFFER>), dim3(blocks), dim3(256), 0, 0, d_src, d_out, misalign, reps, n_floats);
    CHECK(hipEventRecord(e1, 0));
    hipError_t err = hipDeviceSynchronize();
    float ms = 0.0f;
    if (err == hipSuccess) CHECK(hipEventElapsedTime(&ms, e0, e1));
    // per CU: 4 resident blocks-worth of waves issue `reps` instructions each; 4 waves per block, blocks / 256 rounds per CU
    const double instr_per_cu = (double)reps * 4.0 * blocks / 256.0;
    printf("{\"probe\": \"lds_dma\", \"form\": \"%s\", \"bytes_per_lane\": %d, \"global_misalign_bytes\": %d, \"mismatches\": %d, \"error\": \"%s\", "
           "\"ms\": %.4f, \"ns_per_wave_instruction_per_cu\": %.2f}\n",
           BUFFER ? "buffer_load_lds" : "global_load_lds", SIZE, misalign * 4, bad, err == hipSuccess ? "" : hipGetErrorString(err), ms,
           (double)ms * 1e6 / instr_per_cu);
    fflush(stdout);
}

int main() {
    const int n_floats = 256 * 4 * 4096 + 4096;
    std::vector<float> h(n_floats);
    for (int i = 0; i < n_floats; ++i) h[i] = (float)(i % 100003) * 0.5f;
    float *d_src, *d_out;
    CHECK(hipMalloc(&d_src, (size_t)n_floats * 4));
    CHECK(hipMalloc(&d_out, 4096 * 4));
    CHECK(hipMemcpy(d_src, h.data(), (size_t)n_floats * 4, hipMemcpyHostToDevice));
    for (int mis : {0, 1, 3}) {
        run<4, false>(d_src, d_out, h, mis, n_floats);
        run<12, false>(d_src, d_out, h, mis, n_floats);
        run<16, false>(d_src, d_out, h, mis, n_floats);
        run<4, true>(d_src, d_out, h, mis, n_floats);
        run<16, true>(d_src, d_out, h, mis, n_floats);
    }
    hipLaunchKernelGGL(oob_kernel, dim3(1), dim3(64), 0, 0, d_src, d_out, n_floats);
    CHECK(hipDeviceSynchronize());
    std::vector<float> got(256);
    CHECK(hipMemcpy(got.data(), d_out, 256 * 4, hipMemcpyDeviceToHost));
    int zeros_ok = 0, data_ok = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 4; ++j) {
            const float v = got[lane * 4 + j];
            if (lane & 1) zeros_ok += v == 0.0f;
            else data_ok += v == h[lane * 4 + j];
        }
    printf("{\"probe\": \"buffer_load_lds_out_of_range\", \"out_of_range_lanes_wrote_zero\": %d, \"of\": 128, \"in_range_values_correct\": %d, \"of_\": 128}\n", zeros_ok,
           data_ok);
    return 0;
}
