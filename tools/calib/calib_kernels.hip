// Known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in the access patterns of the quad warp
// kernels (MI355X_MICROARCH.md, HBM section: only the wide coalesced streaming read is calibrated there -- FETCH_SIZE reports
// half of its bytes -- "calibrate on a known byte count in your own access pattern before trusting an absolute").
// Test / measurement infrastructure: built by tools/traffic_calib.py into its own .so, not part of libdmvs_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// (1) wide coalesced streaming read: 16 B per lane, consecutive lanes consecutive addresses; n16 = number of 16-byte pieces
__global__ void __launch_bounds__(256) calib_stream_read(const u32x4* __restrict__ src, uint32_t* __restrict__ sink, long n16) {
    uint32_t acc = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        const u32x4 v = src[i];
        acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;      // never true for the fill pattern; keeps the loads alive
}

// (2) the quad gather of getcost_quad_kernel<32>: a quad reads one 128-byte texel as two 64-byte units (lane q: 16 bytes of
// each), every quad a different texel, texels visited in a scrambled order (odd multiplier modulo a power of two: a
// permutation, every texel exactly once) -- units = 2: whole texel (= whole 128-byte line); units = 1: only its first 64 bytes
__global__ void __launch_bounds__(256) calib_quad_gather(const char* __restrict__ src, uint32_t* __restrict__ sink, long ntexels_log2, int units) {
    const long n = 1L << ntexels_log2;
    uint32_t acc = 0;
    const int q = threadIdx.x & 3;
    for (long p = ((long)blockIdx.x * 256 + threadIdx.x) >> 2; p < n; p += ((long)gridDim.x * 256) >> 2) {
        const long texel = (p * 0x9E3779B1L + 12345L) & (n - 1);
        const char* t = src + texel * 128 + q * 16;
        const u32x4 a = *reinterpret_cast<const u32x4*>(t);
        acc += a[0] ^ a[1] ^ a[2] ^ a[3];
        if (units == 2) {
            const u32x4 b = *reinterpret_cast<const u32x4*>(t + 64);
            acc += b[0] ^ b[1] ^ b[2] ^ b[3];
        }
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

// (3) the output pattern of the quad kernels: 4 bytes per lane, a wave writes 4 runs of 64 bytes (16 pixels) in 4 planes that
// lie `plane` floats apart, n planes x plane floats in total, every float written exactly once
__global__ void __launch_bounds__(256) calib_store_runs(float* __restrict__ dst, long plane, int nplanes4) {
    const int q = threadIdx.x & 3;
    for (int g = 0; g < nplanes4; ++g) {
        for (long pix = ((long)blockIdx.x * 256 + threadIdx.x) >> 2; pix < plane; pix += ((long)gridDim.x * 256) >> 2)
            dst[(long)(g * 4 + q) * plane + pix] = (float)pix;
    }
}

// (4) wide coalesced streaming write: 16 B per lane
__global__ void __launch_bounds__(256) calib_stream_write(u32x4* __restrict__ dst, long n16) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) dst[i] = u32x4{1u, 2u, 3u, (uint32_t)i};
}

extern "C" int calib_run(int which, void* buf, void* sink, long bytes, int arg, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(256 * 16), block(256);
    switch (which) {
        case 1: hipLaunchKernelGGL(calib_stream_read, grid, block, 0, st, (const u32x4*)buf, (uint32_t*)sink, bytes / 16); break;
        case 2: {
            long lg = 0;
            while ((128L << (lg + 1)) <= bytes) ++lg;
            hipLaunchKernelGGL(calib_quad_gather, grid, block, 0, st, (const char*)buf, (uint32_t*)sink, lg, arg);
            break;
        }
        case 3: hipLaunchKernelGGL(calib_store_runs, grid, block, 0, st, (float*)buf, bytes / 4 / (4 * arg), arg); break;
        case 4: hipLaunchKernelGGL(calib_stream_write, grid, block, 0, st, (u32x4*)buf, bytes / 16); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
