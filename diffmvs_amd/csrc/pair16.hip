// Two 16 -> 16 channel 3x3 convolutions (+ folded BN + ReLU each) in ONE kernel: FeatureNet conv1.1 + conv1.2 at half
// resolution (reference models/module.py:368-371, applied at :400).  As two conv2d launches the pair moves 4 x 16 planes per
// image through HBM (6 GB per launch at the bench batch, 2.4 TB/s) and sits at 0.55 of the matrix peak whatever is done to its
// matrix time, staging, tile shape or occupancy; fused, the 16-channel intermediate lives in LDS and each
// image plane is read once and written once.
//   * persistent workgroups walk 16 x 16 output tiles.  ONE input buffer: the 16 x 20 x 20 input halo of the next tile streams
//     in (LDS-DMA) while THIS tile's second convolution runs from the intermediate -- the first convolution is the only
//     reader of the input buffer;
//   * conv A on the 18 x 18 halo'd intermediate as an implicit GEMM, 21 groups of 16 intermediate pixels taken round-robin by
//     the 4 waves, B operand gathered from the input halo (lane = pixel), A = weights [k = (ci, tap)][cout]; BN + ReLU, zeroed
//     outside the image (it is conv B's zero padding), written to LDS [16][18 x 18];
//   * conv B from that LDS image, a wave = 4 output rows, pixels as the A operand (transposed accumulators: a lane ends up with 4
//     consecutive pixels of one channel): BN + ReLU + one 16-byte NCHW store per (row, lane), issued one iteration late so
//     that the barrier's vmcnt(0) does not wait them out.
// Exact-fp32 MFMA; every output sums its products in (ci-group, ci, tap) order like dmvs_conv2d_f32: the results are those of the
// two separate launches bit for bit.  1.31x the matrix work of conv A (the halo ring) for half the HBM traffic of the pair.
#include "dmvs_common.h"
#include "dmvs_lds_poison.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define DMVS_LDS_P(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int PC = 16;                             // channels in = mid = out
constexpr int PTS = 16;                            // output tile
constexpr int PMW = PTS + 2, PMP = PMW * PMW;      // intermediate tile 18 x 18 = 324
constexpr int PIW = PTS + 4, PIP = PIW * PIW;      // input tile 20 x 20 = 400
constexpr int PIPL = PIP;                          // input channel pitch: 400 = 16 mod 32 already (the 4 k-groups on disjoint banks)
constexpr int PMPL = 336;                          // intermediate channel pitch: 324 padded to 16 mod 32
constexpr int PWS = 9 * 16;                        // weight slab per input channel [tap][cout] (144 = 16 mod 32)
constexpr int PIN_FLOATS = PC * PIPL;

__device__ __attribute__((aligned(16))) const float pair_zero16[4] = {0.0f, 0.0f, 0.0f, 0.0f};

__global__ void __launch_bounds__(DMVS_BLOCK)
conv3x3_pair16_kernel(const float* __restrict__ x, const float* __restrict__ wa, const float* __restrict__ scale_a,
                      const float* __restrict__ shift_a, const float* __restrict__ wb, const float* __restrict__ scale_b,
                      const float* __restrict__ shift_b, float* __restrict__ y, int N, int H, int W, int tiles_x, int tiles_y) {
    // one LDS object (see stem.hip: separate objects make hipcc wait out the LDS-DMA before unrelated ds_reads)
    __shared__ __attribute__((aligned(16))) float lds[PIN_FLOATS + PC * PMPL + 2 * PC * PWS];
    DMVS_LDS_POISON(lds);
    float* const s_in = lds;
    float* const s_mid = lds + PIN_FLOATS;
    float* const s_wa = s_mid + PC * PMPL;
    float* const s_wb = s_wa + PC * PWS;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;
    const long plane = (long)H * W;
    const int ntiles = tiles_x * tiles_y * N;

    // weights [cin][9][16] (the kernel layout of dmvs_conv2d_f32, cout_pad = 16) -> LDS once per workgroup
    for (int e = tid; e < PC * PWS; e += DMVS_BLOCK) {
        s_wa[e] = wa[e];
        s_wb[e] = wb[e];
    }

    // input halo staging (16 x 20 x 20, zero padded), 4-byte LDS-DMA; the (channel, row, column) of a lane's pieces is
    // tile-independent: decoded once, per tile one packed border compare (guard bits as in stem.hip)
    constexpr int S_IT = (PC * PIP + DMVS_BLOCK - 1) / DMVS_BLOCK;       // 25
    constexpr unsigned kGuard = 0x8080u;
    int e_off[S_IT];                         // r * W + c from the halo origin (the channel advances by a plane per 400 pieces)
    short e_rc[S_IT];                        // r | c << 8
    // piece e = i * 256 + tid covers channel e / 400, position e % 400 (16 x 400 = 25 x 256 pieces: no tail)
#pragma unroll
    for (int i = 0; i < S_IT; ++i) {
        const int e = i * DMVS_BLOCK + tid;
        const int ci = e / PIP, rem = e - ci * PIP;
        const int r = rem / PIW, c = rem - r * PIW;
        e_off[i] = ci * (int)plane + r * W + c;
        e_rc[i] = (short)(r | (c << 8));
    }
    auto stage = [&](int tile) {
        int tq = tile;
        const int tx = tq % tiles_x; tq /= tiles_x;
        const int ty = tq % tiles_y;
        const int n = tq / tiles_y;
        const int gy0 = ty * PTS - 2, gx0 = tx * PTS - 2;
        const float* origin = x + (long)n * PC * plane + (long)gy0 * W + gx0;       // may lie outside the tensor: only in-range pieces are read
        const unsigned lo = (unsigned)(max(0, -gy0) | (max(0, -gx0) << 8));
        const unsigned him1 = (unsigned)((min(PIW, H - gy0) - 1) | ((min(PIW, W - gx0) - 1) << 8)) | kGuard;
#pragma unroll
        for (int i = 0; i < S_IT; ++i) {
            const int e0 = i * DMVS_BLOCK + wave * 64;           // first piece of this wave-instruction (wave-uniform)
            const unsigned rc = (unsigned)(unsigned short)e_rc[i];
            const bool ok = (((rc | kGuard) - lo) & (him1 - rc) & kGuard) == kGuard;
            const float* srcp = ok ? origin + e_off[i] : pair_zero16;
            __builtin_amdgcn_global_load_lds(srcp, DMVS_LDS_P(s_in + e0), 4, 0, 0);      // LDS image [ci][20 x 20] = piece order
        }
    };

    float sca[4], sha[4];       // conv A: this lane's output channels 4*kq + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sca[r] = scale_a ? scale_a[4 * kq + r] : 1.0f;
        sha[r] = shift_a ? shift_a[4 * kq + r] : 0.0f;
    }
    const bool vec = (W & 3) == 0 && ((uintptr_t)y & 15) == 0;
    const float scb = scale_b ? scale_b[m] : 1.0f, shb = shift_b ? shift_b[m] : 0.0f;       // conv B: channel m (transposed accumulators)
    auto store_tile = [&](const f32x4 (&a)[4], int n, int ox0, int oy0) {
        const int ox = ox0 + 4 * kq;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int oy = oy0 + wave * 4 + mt;
            if (ox < W && oy < H) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(a[mt][r] * scb + shb, 0.0f);
                float* dst = y + ((long)n * PC + m) * plane + (long)oy * W + ox;
                if (vec) {
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ox + r < W) dst[r] = v[r];
                }
            }
        }
    };

    int tile = blockIdx.x;
    f32x4 pend[4];                       // conv B accumulators of the previous tile, not stored yet
    int pn = -1, pox0 = 0, poy0 = 0;
    if (tile < ntiles) stage(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        int tq = tile;
        const int tx = tq % tiles_x; tq /= tiles_x;
        const int ty = tq % tiles_y;
        const int n = tq / tiles_y;
        const int ox0 = tx * PTS, oy0 = ty * PTS;
        __syncthreads();        // this tile's halo has landed (and the weights, first time); everyone is done with s_mid
        if (pn >= 0) store_tile(pend, pn, pox0, poy0);

        // ---- conv A -> s_mid: 21 groups of 16 intermediate pixels (row-major over 18 x 18), waves take groups round-robin
        for (int gidx = wave; gidx < (PMP + 15) / 16; gidx += DMVS_BLOCK / 64) {
            const int p = min(gidx * 16 + m, PMP - 1);
            const int py = p / PMW, px = p - py * PMW;
            const float* ip = s_in + py * PIW + px;
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
            for (int c4 = 0; c4 < PC / 4; ++c4) {
                const float* wp = s_wa + (c4 * 4 + kq) * PWS + m;
                const float* ipc = ip + (c4 * 4 + kq) * PIPL;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[(ky * 3 + kx) * 16], ipc[ky * PIW + kx], acc, 0, 0, 0);      // D[cout][pixel]
            }
            // this lane holds intermediate channels 4*kq + r of pixel p
            const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
            const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
            if (gidx * 16 + m < PMP) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s_mid[(4 * kq + r) * PMPL + p] = inside ? fmaxf(acc[r] * sca[r] + sha[r], 0.0f) : 0.0f;
            }
        }
        DMVS_LDS_BARRIER();     // s_mid complete (ds_writes); every wave is done with the input buffer
        if (tile + (int)gridDim.x < ntiles) stage(tile + gridDim.x);      // streams in under conv B

        // ---- conv B from s_mid: wave = output rows 4*wave .. +3, pixels as the A operand
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
        for (int c4 = 0; c4 < PC / 4; ++c4) {
            const float* wp = s_wb + (c4 * 4 + kq) * PWS + m;
            const float* mp = s_mid + (c4 * 4 + kq) * PMPL + (wave * 4) * PMW + m;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float av = wp[(ky * 3 + kx) * 16];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(mp[(mt + ky) * PMW + kx], av, acc[mt], 0, 0, 0);      // D[pixel][cout]
                }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) pend[mt] = acc[mt];
        pn = n; pox0 = ox0; poy0 = oy0;
    }
    if (pn >= 0) store_tile(pend, pn, pox0, poy0);
}

}  // namespace

extern "C" int dmvs_conv3x3_pair16_f32(const float* x, const float* wa, const float* scale_a, const float* shift_a, const float* wb,
                                       const float* scale_b, const float* shift_b, float* y, int32_t N, int32_t H, int32_t W,
                                       void* stream) {
    if (!x || !wa || !wb || !y || N <= 0 || H <= 0 || W <= 0) return DMVS_EINVAL;
    if ((long)PC * H * W >= (1L << 31)) return DMVS_EINVAL;
    const int tiles_x = (W + PTS - 1) / PTS, tiles_y = (H + PTS - 1) / PTS;
    const long ntiles = (long)tiles_x * tiles_y * N;
    if (ntiles >= (1L << 31)) return DMVS_EINVAL;
    const unsigned grid = (unsigned)(ntiles < 256 * 2 ? ntiles : 256 * 2);      // persistent: 2 workgroups per CU (65 KB of LDS each)
    hipLaunchKernelGGL(conv3x3_pair16_kernel, dim3(grid), dim3(DMVS_BLOCK), 0, (hipStream_t)stream, x, wa, scale_a, shift_a, wb,
                       scale_b, shift_b, y, N, H, W, tiles_x, tiles_y);
    return dmvs_launch_status();
}
