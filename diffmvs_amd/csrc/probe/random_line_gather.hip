// libdmvs_probe.so, part 2: random 128-byte-line gather (include/dmvs_probe.h) -- the calibration kernel for "what does this memory system
// deliver for independent scattered line requests at full occupancy, with no arithmetic at all".  GetCost on noise geometry asks for ~7.3
// distinct 128-byte texels per pixel and view, each as two 64-byte quad-coalesced requests; this kernel issues exactly that request shape
// from 8 waves per SIMD with two lines in flight per lane and trip, over a table of the same footprint, and nothing else.
#include "dmvs_common.h"
#include "dmvs_probe.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// bijection of [0, 2^bits): odd multiplies and xor-shifts are each invertible mod 2^bits
__device__ __forceinline__ uint64_t mix_bits(uint64_t x, int bits, uint32_t seed) {
    const uint64_t m = (1ull << bits) - 1ull;
    x = (x + seed) & m;
    x = (x * 0x9E3779B97F4A7C15ull) & m;
    x ^= x >> (bits / 2);
    x = (x * 0xD6E8FEB86659FD93ull) & m;
    x ^= x >> (bits / 2 + 1);
    x = (x * 0xCA5A826395121157ull) & m;
    x ^= x >> (bits / 2 - 1);
    return x;
}

__device__ __forceinline__ uint32_t hash32(uint32_t a, uint32_t b, uint32_t seed) {
    uint32_t x = a * 0x9E3779B1u ^ (b + seed) * 0x85EBCA77u;
    x ^= x >> 15; x *= 0x2C1B3C6Du;
    x ^= x >> 12; x *= 0x297A2D39u;
    x ^= x >> 15;
    return x;
}

template <int MODE>
__global__ void __launch_bounds__(DMVS_BLOCK) random_line_gather_kernel(const char* __restrict__ table, uint64_t n_lines, uint64_t n_quads, int lines_per_quad,
                                                                        int window, uint32_t seed, int bits) {
    const uint64_t quad = ((uint64_t)blockIdx.x * DMVS_BLOCK + threadIdx.x) >> 2;
    const unsigned q = threadIdx.x & 3;
    if (quad >= n_quads) return;
    auto line_of = [&](int i) -> uint64_t {
        if (MODE == DMVS_GATHER_ONCE) {
            uint64_t x = quad * (uint64_t)lines_per_quad + (uint64_t)i;
            do x = mix_bits(x, bits, seed); while (x >= n_lines);            // cycle walking keeps it a bijection of [0, n_lines)
            return x;
        }
        const uint32_t h = hash32((uint32_t)quad, (uint32_t)i, seed);
        if (MODE == DMVS_GATHER_UNIFORM) return (uint64_t)(((uint64_t)h * n_lines) >> 32);
        const uint64_t l = quad + (((uint64_t)h * (uint32_t)window) >> 32);      // (no 64-bit division in the loop)
        return l >= n_lines ? l - n_lines : l;
    };
    for (int i = 0; i < lines_per_quad; i += 2) {
        const uint64_t l0 = line_of(i), l1 = line_of(min(i + 1, lines_per_quad - 1));
        const char* p0 = table + l0 * 128ull + q * 16u;
        const char* p1 = table + l1 * 128ull + q * 16u;
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(p0), a1 = *reinterpret_cast<const u32x4*>(p0 + 64);
        const u32x4 b0 = *reinterpret_cast<const u32x4*>(p1), b1 = *reinterpret_cast<const u32x4*>(p1 + 64);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(a0[j]), "v"(a1[j]), "v"(b0[j]), "v"(b1[j]));      // waited for, not used
    }
}

}  // namespace

extern "C" int dmvs_probe_random_line_gather(const void* table, int64_t n_lines, int64_t n_quads, int32_t lines_per_quad, int32_t mode,
                                             int32_t window, uint32_t seed, void* stream) {
    if (!table || n_lines < 256 || n_quads < 1 || lines_per_quad < 1) return DMVS_EINVAL;
    if (mode == DMVS_GATHER_ONCE && n_quads * (int64_t)lines_per_quad > n_lines) return DMVS_EINVAL;
    if (mode == DMVS_GATHER_BAND && (window < 1 || window > n_lines || n_quads > n_lines)) return DMVS_EINVAL;
    int bits = 2;
    while ((1ll << bits) < n_lines) ++bits;
    if (bits > 40) return DMVS_EINVAL;
    const int64_t blocks = (n_quads * 4 + DMVS_BLOCK - 1) / DMVS_BLOCK;
    if (blocks > 0x7fffffffll) return DMVS_EINVAL;
    dim3 grid((unsigned)blocks), block(DMVS_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    const char* t = reinterpret_cast<const char*>(table);
    if (mode == DMVS_GATHER_ONCE) hipLaunchKernelGGL(random_line_gather_kernel<DMVS_GATHER_ONCE>, grid, block, 0, st, t, (uint64_t)n_lines, (uint64_t)n_quads, lines_per_quad, window, seed, bits);
    else if (mode == DMVS_GATHER_UNIFORM) hipLaunchKernelGGL(random_line_gather_kernel<DMVS_GATHER_UNIFORM>, grid, block, 0, st, t, (uint64_t)n_lines, (uint64_t)n_quads, lines_per_quad, window, seed, bits);
    else if (mode == DMVS_GATHER_BAND) hipLaunchKernelGGL(random_line_gather_kernel<DMVS_GATHER_BAND>, grid, block, 0, st, t, (uint64_t)n_lines, (uint64_t)n_quads, lines_per_quad, window, seed, bits);
    else return DMVS_EINVAL;
    return dmvs_launch_status();
}
