/* dmvs_probe.h -- C ABI of libdmvs_probe.so: MEASUREMENT probes, not part of the depth-estimation path.
 *
 * bench.py loads this library next to the product library (libdmvs_hip.so, include/dmvs.h) for its untimed roofline legs; nothing under
 * diffmvs_amd/ or models/ does.  Same conventions as dmvs.h: extern "C", raw device pointers + sizes + a hipStream_t passed as void*, int
 * return codes (0 = ok, DMVS_EINVAL, else a hipError_t), no allocation, no global state. */
#ifndef DMVS_PROBE_H
#define DMVS_PROBE_H

#include <stdint.h>

#include "dmvs.h"

#ifdef __cplusplus
extern "C" {
#endif

#define DMVS_PROBE_ABI_VERSION 2
int dmvs_probe_abi_version(void);

/* The memory-system ceiling of GetCost's address stream.  Same descriptor, same launch grid and the SAME kernel body as
 * dmvs_getcost_quad_f32 (csrc/warp_quad_core.h: hypotheses, projection, texel masks, bit scans, addresses, loads), instantiated with a texel
 * body that only waits for the loaded registers: no group dot, no hat weights, no scatter.  out_samples receives the hypotheses, out_cost
 * zeros.  Its launch time is what the L1 / L2 / HBM path alone needs for the product's line requests (reference path: models/module.py:583-667). */
int dmvs_probe_getcost_loads_f32(const dmvs_getcost_desc* d, void* stream);

/* The same stream with the texel PAIR (x, x + 1) as the load unit -- one 16-byte load per lane fetches the 64 contiguous bytes of two 16-bit
 * C = 16 texels (ABI 2; round 6: would a pair-packed 16-bit feature layout pay?  16-bit features, C = 16, n = 4, even W only; the last texel
 * of an image row pairs with the first of the next, whose bytes are read and ignored). */
int dmvs_probe_getcost_pair_loads_f32(const dmvs_getcost_desc* d, void* stream);

/* Random 128-byte-line gather: every quad of lanes requests `lines_per_quad` lines of `table` (n_lines x 128 bytes), each as two 64-byte
 * quad-coalesced pieces (one 16-byte load per lane and piece -- the request shape of a C = 32 fp32 texel in GetCost), two lines in flight per
 * lane and trip, nothing computed.
 *   mode DMVS_GATHER_ONCE   : line = bijective_hash(quad * lines_per_quad + i) -- every line of the table at most once, in scrambled order
 *                             (n_quads * lines_per_quad <= n_lines): the rate at which the memory system retires independent missing lines;
 *   mode DMVS_GATHER_UNIFORM: line = hash(quad, i) % n_lines -- uniformly random with repeats far apart in time;
 *   mode DMVS_GATHER_BAND   : line = (quad + hash(quad, i) % window) % n_lines -- a quad's lines lie in a `window`-line band that overlaps its
 *                             neighbours' bands (each line is asked for by ~lines_per_quad quads close in time): the locality of texels
 *                             scattered along epipolar segments, without any projection arithmetic. */
#define DMVS_GATHER_ONCE 0
#define DMVS_GATHER_UNIFORM 1
#define DMVS_GATHER_BAND 2
int dmvs_probe_random_line_gather(const void* table, int64_t n_lines, int64_t n_quads, int32_t lines_per_quad, int32_t mode, int32_t window,
                                  uint32_t seed, void* stream);

#ifdef __cplusplus
}
#endif
#endif
