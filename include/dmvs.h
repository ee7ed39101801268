/*
 * dmvs.h -- C ABI of libdmvs_hip.so: the MI355X (gfx950) kernels behind the DiffMVS /
 * CasDiffMVS depth-estimation path.
 *
 * The reference (cvg/diffmvs) has no native layer at all: every entry point
 * below replaces a span of PyTorch ops inside models/module.py / models/update.py, cited
 * per function as  <file>:<lines>  relative to the reference repository root.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is allocated, freed or
 *     retained by the library; no global state, no environment variables: every choice a caller can make is an argument
 *     or a descriptor field (the `tune` fields: 0 = the measured-best path; the other values exist for A/B measurements
 *     and for tests that pin a code path)  => re-entrant, graph-capturable;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing syncs;
 *   - the return value is 0 on success, a hipError_t otherwise, or DMVS_EINVAL for a
 *     descriptor the library cannot run (unsupported kernel size, channel tile, ...);
 *   - activations are fp32, planar NCHW / NCDHW unless a field says NHWC;
 *   - functions never throw.
 */
#ifndef DMVS_H
#define DMVS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): dmvs_conv2d_desc gained `arith` (round 3, without a bump) and `tune`; dmvs_conv3d_desc gained `tune`;
 * dmvs_featurenet_stem_f32 and dmvs_warp_corr_init_quad_f32 take a trailing `tune` argument; dmvs_conv3x3_pair16_f32 is gone;
 * the library no longer reads any environment variable; dmvs_conv2d_desc also gained `out_mul`, `out_mul_c0` (producer-side output
 * product, SepConvGRU's r * h) and `in0_cstride` (in0 as a channel slice) -- honoured by dmvs_conv2d_f32 only: the weight-gradient entry
 * points return DMVS_EINVAL when any of in0_cstride / gate_cstride / out_mul is set.  A caller built against version 1 passes shorter descriptors:
 * check dmvs_abi_version() == DMVS_ABI_VERSION before the first call. */
#define DMVS_ABI_VERSION 4
#define DMVS_EINVAL (-22)

/* activation codes for the fused epilogues */
enum { DMVS_ACT_NONE = 0, DMVS_ACT_RELU = 1, DMVS_ACT_SIGMOID = 2, DMVS_ACT_TANH = 3, DMVS_ACT_SILU = 4 };
/* how the logical input of a 2-D convolution is read from memory */
enum { DMVS_IN_PLAIN = 0, DMVS_IN_UPSAMPLE2 = 1 /* nearest x2, F.interpolate */, DMVS_IN_UNSHUFFLE2 = 2 /* einops 'b c (h p1) (w p2) -> b (c p1 p2) h w' */,
       DMVS_IN_ZEROINSERT2 = 3 /* logical (2y,2x) = physical (y,x), zero elsewhere: the input-gradient of a stride-2 conv as a stride-1 conv */ };
enum { DMVS_LAYOUT_NCHW = 0, DMVS_LAYOUT_NHWC = 1,
       /* channel-last output stored as 16-bit elements (round to nearest even): the reduced-precision FEATURE storage of
          BASELINE.json's bf16 / fp16 configurations; `out` then points at uint16 elements, strides / offsets count elements */
       DMVS_LAYOUT_NHWC_BF16 = 2, DMVS_LAYOUT_NHWC_F16 = 3 };
/* element type of the image-feature tensors handed to the quad warp kernels */
enum { DMVS_DTYPE_F32 = 0 /* fp32 in the NHWC-g4 channel order (see dmvs_getcost_quad_f32) */, DMVS_DTYPE_BF16 = 1, DMVS_DTYPE_F16 = 2 /* 16-bit, plain NHWC */,
       DMVS_DTYPE_F32_PLAIN = 3 /* fp32, plain NHWC: the training graph's features (its backward kernels read the same order) */ };

int dmvs_abi_version(void);

/* ---------------------------------------------------------------------------------------
 * 2-D convolution with everything around it fused.   Replaces, depending on the fields:
 *   module.Conv2d / ConvBnReLU / ConvBn (conv -> eval BN -> ReLU)      models/module.py:24-58, :279-301
 *   ResidualBlock's relu(x + y)                                        models/module.py:315-319
 *   FeatureNet's  nearest-x2(intra) + inner(conv)                      models/module.py:409-417
 *   ConditionEncoder convs, mask heads (0.25 * conv)                   models/update.py:289-297, :335-339,:473
 *   Unet init_conv / Downsample (pixel-unshuffle + 1x1) / Upsample     models/update.py:38-48, :186, :246
 *   WeightStandardizedConv2d (weights are pre-standardised by caller)  models/update.py:81-94
 *   SepConvGRU gates: z|r = sigmoid(conv[h,x]);  h' = (1-z)h + z*tanh(conv[r*h,x])
 *                                                                      models/module.py:164-177
 * Logical input = concat(in0 (* mul0), in1) along channels, read through `in_mode`.
 * Weight layout: [c0+c1][kh][kw][cout_pad] (cout fastest, zero padded to cout_pad, a
 * multiple of 8).  Epilogue, per output channel co:
 *   y = acc * scale[co] + shift[co]         (scale NULL = 1, shift NULL = 0; folded BN / bias)
 *   y += residual (read through res_mode)   if residual && !res_after_act
 *   y = act(y) * post_scale
 *   y += residual                           if residual &&  res_after_act
 *   y = (1 - gru_z) * gru_h + gru_z * y     if gru_z           (act must be TANH)
 *   y *= out_mul[co - out_mul_c0]           if out_mul && co >= out_mul_c0
 * Output is written at channel offset out_coffset of a tensor with out_cstride channels
 * (so that concatenations never have to be materialised), NCHW or NHWC.
 */
#define DMVS_ARITH_F32 0
#define DMVS_ARITH_BF16 1
#define DMVS_ARITH_SPLIT 2

/* dmvs_conv2d_desc.tune: 0 = the library's own choice (measured best on the MI355X); the bits force a code path for A/B runs.
 * Results do not depend on them (bit-identical kernels). */
#define DMVS_TUNE_TILE_WX(n) ((n) & 3)           /* 1 | 2: 16- / 32-pixel-wide workgroup tiles for the 3x3 / 5x5 layers           */
#define DMVS_TUNE_NO_WALK 0x4                     /* one tile per workgroup everywhere (no resident tile-walking workgroups)        */
#define DMVS_TUNE_PIECES4 0x8                     /* input halo staged in 4-byte LDS-DMA pieces even where 16-byte ones apply        */
#define DMVS_TUNE_TILE_MT(n) (((n) & 7) << 4)     /* A/B: 1 | 2 | 4 = tile height in units of 4 rows (4: plain 3x3 / 1xk families only; timed on the stride-2 / 5x5 / 7x7 ones in round 5: +-2 %, removed) */
#define DMVS_TUNE_1X1_TILED 0x200                 /* 1x1 layers on the LDS-tiled kernel instead of the 16-byte direct form                    */
#define DMVS_TUNE_NO_LEAN 0x100                   /* plain layers on the generic kernel (every fused path resolved at run time)     */
#define DMVS_TUNE_TALL(n) (((n) & 3) << 10)      /* 16 x 32-pixel tiles for the plain 3x3 layers: 0 = where measured better, 1 = never, 2 = wherever they apply (16 x 64 tiles: timed in round 5, 6-8 % slower, removed) */
#define DMVS_TUNE_STEM_EXACT 0x40000             /* dmvs_featurenet_stem_f32: conv0.1 in exact fp32 (0 = split-bf16 arithmetic with fp32 accuracy, the default) */
#define DMVS_TUNE_SPLIT_ALL 0x20000              /* DMVS_ARITH_SPLIT on every layer the form applies to, not only where it measured faster (A/B runs, tests) */
#define DMVS_TUNE_XCD_GROUP(n) (((n) & 7) << 14) /* tiled kernels, one tile per workgroup: which tiles share an XCD's L2.  0 = the library's choice, 1 = plain round robin, 2 | 3 | 4 = groups of 2 | 4 | 8 x-adjacent tiles, 5 = one tile row, 6 = two tile rows, 7 = one image (bit-identical results) */

typedef struct dmvs_conv2d_desc {
    const float* in0;       /* [B,c0,*,*] physical tensor                                   */
    const float* in1;       /* [B,c1,Hin,Win] or NULL; only with DMVS_IN_PLAIN               */
    const float* mul0;      /* [B,c0,Hin,Win] or NULL: in0 is multiplied element-wise       */
    const float* weight;    /* 16-byte aligned (staged in 16-byte pieces); DMVS_EINVAL otherwise */
    const float* scale;
    const float* shift;
    const float* residual;  /* [B,cout,*,*] or NULL                                         */
    const float* gru_z;     /* [B,cout,Hout,Wout] or NULL                                   */
    const float* gru_h;
    float* out;
    double* gn_stats;       /* [B,gn_groups,2] 8-byte slots or NULL (opaque: fixed-point integers, so that the accumulation
                               order of the workgroups cannot change the result -- zero-initialise, hand to
                               dmvs_groupnorm_apply_f32): += per-(b,group) sum and sum of squares of y
                               BEFORE the activation (GroupNorm statistics of the conv output, so
                               that Block.forward needs no separate reduction pass; the caller
                               zeroes the buffer).  gn_groups must be 4 and divide cout.       */
    const float* out_mul;   /* [B, cout - out_mul_c0, Hout, Wout] dense, or NULL: output channels co >= out_mul_c0 are multiplied by
                               out_mul[b, co - out_mul_c0] LAST (after activation, post-scale, residual).  SepConvGRU: the merged z|r
                               gate convolution writes [z | r * h] directly, so that the candidate convolution reads a plain input
                               instead of gating its staged tile (models/module.py:166-168)                                  */
    int32_t B, c0, c1;
    int32_t Hin, Win;       /* LOGICAL input size (after in_mode)                           */
    int32_t Hout, Wout;
    int32_t cout, cout_pad;
    int32_t kh, kw, stride, pad_h, pad_w;
    int32_t in_mode, act;
    int32_t res_mode;       /* DMVS_IN_PLAIN or DMVS_IN_UPSAMPLE2                           */
    int32_t res_after_act;
    int32_t out_layout, out_cstride, out_coffset;
    int32_t gn_groups;
    float post_scale;
    int32_t gate_cstride;   /* 0: mul0 / gru_z are dense [B,c0,..] / [B,cout,..] tensors.  > 0: both are channel slices (the pointers
                               include the channel offset) of tensors with this many channels per batch item -- SepConvGRU's z and
                               r gates computed by ONE convolution with 2x cout (models/module.py:164-177)              */
    int32_t arith;          /* DMVS_ARITH_F32 (0): exact fp32 products (v_mfma_f32_16x16x4_f32), the reference's precision.
                               DMVS_ARITH_BF16 (1): inputs and weights rounded to bf16 (nearest even) as they enter the matrix
                               cores, fp32 accumulation (v_mfma_f32_16x16x32_bf16) -- the reduced-precision configurations of
                               BASELINE.json (configs[2], [4]); tensors in memory stay fp32.  Honoured by stride-1 layers with
                               more than one tap, >= 24 input channels and an NCHW output (where it is faster); every other
                               layer computes in fp32 in either mode.
                               DMVS_ARITH_SPLIT (2): fp32 accuracy on the bf16 matrix cores -- every fp32 operand is split into three
                               bf16 values (hi + mid + lo = x to 2^-27) and a product is the sum of its six partial products down
                               to 2^-18 of it (the dropped ones are below 2^-26), fp32 accumulation.  Honoured by multi-tap layers with an NCHW
                               output on 16-byte aligned rows; the others compute in exact fp32.                           */
    int32_t tune;           /* DMVS_TUNE_* bits, 0 = automatic                                                              */
    int32_t out_mul_c0;     /* first output channel out_mul applies to                                                       */
    int32_t in0_cstride;    /* 0: in0 is a dense [B,c0,..] tensor.  > 0 (DMVS_IN_PLAIN only): in0 is a channel slice (the pointer includes
                               the channel offset) of a tensor with this many channels per batch item                         */
    const void* weight_split; /* ABI 4.  DMVS_ARITH_SPLIT only (else NULL): the layer's weights pre-split into bf16 triples in the matrix
                               cores' operand order, 16-byte aligned: [ceil(cin / 8)][ceil(kh*kw / 4)][3 planes: hi, mid, lo][4][cout_pad][8]
                               bf16 -- element (c, g, p, q, co, j) = part p of the weight of input channel 8c + j, tap 4g + q, output
                               channel co (zero beyond cin / kh*kw / cout); hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid),
                               round to nearest even.  A NULL pointer with DMVS_ARITH_SPLIT computes in exact fp32.               */
} dmvs_conv2d_desc;

/* Size limits (DMVS_EINVAL beyond them; the kernels address one batch item with 32-bit element offsets):
 *   (c0 + c1) * Hin * Win < 2^31,  max(cout, out_cstride) * Hout * Wout < 2^31,  Hout * Wout < 2^24,  out_cstride < 2^24. */
int dmvs_conv2d_f32(const dmvs_conv2d_desc* d, void* stream);

/* FeatureNet stem in one kernel: relu(bn1(conv3x3(relu(bn0(conv3x3(x)))))) with 3 -> 8 -> 8 channels, padding 1, at full
 * resolution (models/module.py:364-367 conv0, applied at :399); the 8-channel intermediate never leaves LDS.
 *   x [N,3,H,W], y [N,8,H,W] NCHW;  w0 [3][3][3][8], w1 [8][3][3][8] in the kernel weight layout of dmvs_conv2d_f32
 *   (cout_pad = 8); scale / shift [8] = folded eval BatchNorm (NULL = 1 / 0).  Agrees with the two dmvs_conv2d_f32
 *   launches it replaces to the last bits (different summation grouping of conv0.0's 27 products). */
/* tune: 0 = automatic (input halo in 16-byte LDS-DMA pieces when W % 4 == 0 and x is 16-byte aligned: 1094 -> 938 us per 96
 * images on the MI355X, profiles/r4_optins_ab.jsonl); DMVS_TUNE_PIECES4 = 4-byte pieces (bit-identical results). */
int dmvs_featurenet_stem_f32(const float* x, const float* w0, const float* scale0, const float* shift0, const float* w1,
                             const float* scale1, const float* shift1, float* y, int32_t N, int32_t H, int32_t W, int32_t tune,
                             void* stream);

/* Weight (and bias) gradient of the convolution described by `d` (its input side: in0 / in1 / mul0 / in_mode / kh /
 * kw / stride / pad / cout / cout_pad / B / Hin / Win / Hout / Wout; epilogue fields are ignored; in0 / in1 / mul0 must be DENSE tensors:
 * DMVS_EINVAL when in0_cstride, gate_cstride or out_mul is set):
 *   gw[co][ci][ky][kx] = sum_{b,y,x} grad_out[b,co,y,x] * X[b,ci,y*stride+ky-pad,x*stride+kx-pad]     (torch layout)
 *   gb[co]             = sum_{b,y,x} grad_out[b,co,y,x]                                               (gb may be NULL)
 * grad_out [B,cout,Hout,Wout] NCHW.  Every element of gw / gb is written exactly once (no pre-zeroing) and the
 * summation order is fixed: workgroups store per-slot partial sums into `workspace` (caller-owned scratch of at
 * least dmvs_conv2d_wgrad_workspace_f32() bytes) and a second kernel folds the slots -- run-to-run reproducible
 * gradients.  The input gradient needs no entry point of its own: it is dmvs_conv2d_f32 on grad_out with the
 * spatially flipped, cin<->cout transposed weights (DMVS_IN_ZEROINSERT2 for stride 2).
 * d->tune: DMVS_TUNE_PIECES4 = 4-byte staging pieces (A/B; bit-identical); DMVS_TUNE_WGRAD_ACCUMULATE = the fold kernel ADDS its sums to gw /
 * gb instead of storing them (the caller's running gradient of a weight that several layers of a step share: one read-modify-write
 * per element instead of a store here + an add kernel elsewhere; the order of the additions is the order of the calls on the stream). */
#define DMVS_TUNE_WGRAD_ACCUMULATE 0x1000
int dmvs_conv2d_wgrad_workspace_f32(const dmvs_conv2d_desc* d, int64_t* bytes);
int dmvs_conv2d_wgrad_f32(const dmvs_conv2d_desc* d, const float* grad_out, float* gw, float* gb, float* workspace,
                          int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * 3-D convolution 3x3x3, padding 1 (module.Conv3d, models/module.py:66-102) and stride-2
 * transposed convolution with output_padding 1 (module.Deconv3d, :110-144, :436-437), with
 * eval-BN folded into scale/shift, optional ReLU and an optional post-activation residual
 * (CostRegNet_small skip adds, models/module.py:445-446).
 * Weight layout: [cin][27][cout_pad].  For transposed convs the caller passes the weight of
 * the equivalent gather form: w[ci][kd][kh][kw][co] = W_torch[ci][co][kd][kh][kw].
 */
typedef struct dmvs_conv3d_desc {
    const float* in;        /* [B,cin,Din,Hin,Win]                                          */
    const float* weight;    /* 16-byte aligned (staged in 16-byte pieces); DMVS_EINVAL otherwise */
    const float* scale;
    const float* shift;
    const float* residual;  /* [B,cout,Dout,Hout,Wout] or NULL, added after the activation  */
    float* out;             /* [B,cout,Dout,Hout,Wout]                                      */
    int32_t B, cin, cout, cout_pad;
    int32_t Din, Hin, Win;
    int32_t Dout, Hout, Wout;
    int32_t stride;         /* 1 or 2                                                       */
    int32_t transposed;     /* 0 | 1 (stride must be 2)                                     */
    int32_t act;
    int32_t tune;           /* 0 = automatic; DMVS_TUNE3D_* bits force a code path (A/B runs; bit-identical results except
                               DMVS_TUNE3D_S2_DIRECT, whose summation order is the direct kernel's)                        */
} dmvs_conv3d_desc;
#define DMVS_TUNE3D_PIECES4 0x1       /* stride-1 MFMA kernels: halo tile in 4-byte LDS-DMA pieces even where 16-byte ones apply  */
#define DMVS_TUNE3D_S2_DIRECT 0x2     /* stride-2 layers on the direct (VALU) kernels of round 1 instead of the matrix cores      */
#define DMVS_TUNE3D_NO_PAIR 0x4       /* A/B: the <= 8 -> <= 8 channel stride-1 layers on the generic / streamed kernels instead of the paired ones (two output depth
                                         slices per MFMA; bit-identical results; round 5: PixelViewWeight conv0 -5 %, CostRegNet conv1 1517 -> 983 us per 96 volumes) */
#define DMVS_TUNE3D_XCD_GROUP(n) (((n) & 7) << 4) /* tiled 3-D kernels: which tiles share an XCD's L2.  0 = groups of 4 x-adjacent tiles (the default), 1 = plain round robin, 2 | 3 | 4 = groups of 2 | 4 | 8 (bit-identical results) */

/* Size limit of the stride-1 layers (DMVS_EINVAL beyond): cin * Din*Hin*Win < 2^31 and cout * Dout*Hout*Wout < 2^31
 * (one batch item is addressed with 32-bit element offsets). */
int dmvs_conv3d_f32(const dmvs_conv3d_desc* d, void* stream);

/* Weight (and bias) gradient of the (non-transposed, stride 1|2) 3x3x3 convolution described by `d`:
 *   gw[co][ci][27] = sum_{b,voxel} grad_out[b,co,voxel] * in[b,ci,voxel*stride - 1 + tap]      (torch layout)
 *   gb[co]         = sum_{b,voxel} grad_out[b,co,voxel]                                          (gb may be NULL)
 * Written once, fixed summation order, via caller-owned `workspace` as for dmvs_conv2d_wgrad_f32.
 * The transposed layers use the same entry point with the roles of `in` and `grad_out` swapped.  Input gradients
 * need no entry point: stride 1 = dmvs_conv3d_f32 on flipped/transposed weights, stride 2 <-> transposed. */
int dmvs_conv3d_wgrad_workspace_f32(const dmvs_conv3d_desc* d, int64_t* bytes);
int dmvs_conv3d_wgrad_f32(const dmvs_conv3d_desc* d, const float* grad_out, float* gw, float* gb, float* workspace,
                          int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Camera composition.  For each batch item b and source view s = 1..S:
 *   P = (K_s E_s[:3,:4] | 0001) * inverse(K_0 E_0[:3,:4] | 0001)   models/module.py:188, :520-525
 * proj: [B,V,2,4,4] (V = S+1; [:, :, 0] extrinsic, [:, :, 1, :3, :3] intrinsic).
 * out:  [B,S,12]  = rot (row-major 3x3) followed by trans (3).  Computed in fp64 from the
 * fp32 inputs and rounded once.
 */
int dmvs_compose_proj_f32(const float* proj, float* out, int32_t B, int32_t V, void* stream);

/* ---------------------------------------------------------------------------------------
 * The two fused warp kernels ("quad per pixel", warp_quad.hip): any geometry in ONE launch.
 *
 * dmvs_warp_corr_init_quad_f32 -- plane-sweep group-wise correlation volumes for depth initialisation: differentiable_warping
 * (models/module.py:181-218) + the group-wise correlation of InitialCost.forward (:514-531) for ALL source views; the hypotheses
 * are uniform in normalised inverse depth (models/diffusion.py:187-192) and generated in-kernel:
 *   depth_d = 1 / clamp(1/dmax + (1/dmin - 1/dmax) * d/(D-1), 1e-6).
 *   ref  [B,H,W,C]   src  [S][B,Hs,Ws,C] channel-last, contiguous over S     rt [B,S,12] from dmvs_compose_proj_f32
 *   disp_min/disp_max [B] (= depth_values[:,0], depth_values[:,-1])           out [B,S,G,D,H,W], cor[g] = mean over the group's channels
 *   C in {16,32,48}, G = 4.
 *
 * dmvs_getcost_quad_f32 -- GetCost.forward (models/module.py:583-667): hypothesis generation (get_cur_depth_range_samples :250-277 +
 * disp_to_depth :220-227), S homography warps, group-wise correlation and view-weighted aggregation
 * sum_s w_s cor_s / (1e-8 + sum_s w_s).
 *   inv_depth  [B,1,H,W] normalised inverse depth      confidence [B,H,W] or NULL
 *   view_w     [B,S,H>>vw_shift,W>>vw_shift]  (nearest-upsampled on the fly, models/diffusion.py:219-221)
 *   out_cost   [B,G*n,H,W]  written at channel offset cost_coffset of a tensor with cost_cstride channels; out_samples [B,n,H,W] likewise
 *   n in {4,6}; interval = depth_interval * ratio (models/diffusion.py:246).
 *
 * (Round 1's LDS-window / per-pixel-gather forward kernels -- dmvs_getcost_f32, dmvs_getcost_gather_f32, dmvs_warp_corr_init_f32,
 * dmvs_warp_corr_init_gather_f32 -- were retired in round 4 / ABI 2: the training graph runs the quad kernels too, on
 * DMVS_DTYPE_F32_PLAIN features.  `worklist` remains for the BACKWARD's hybrid window / per-pixel launch, dmvs_getcost_bwd_f32.)
 */
#define DMVS_GETCOST_TILE 16
#define DMVS_GETCOST_MAX_WINDOW_VIEWS 16      /* more source views than this: dmvs_getcost_bwd_f32 takes the per-pixel path */
#define DMVS_GETCOST_WORKLIST_INTS(B, H, W) \
    (4 + 66 * (B) * (((H) + DMVS_GETCOST_TILE - 1) / DMVS_GETCOST_TILE) * (((W) + DMVS_GETCOST_TILE - 1) / DMVS_GETCOST_TILE))
typedef struct dmvs_getcost_desc {
    const float* ref;       /* [B,H,W,C] channel-last (16-bit elements behind the pointer for feat_dtype BF16 / F16) */
    const float* src;       /* [S][B,H,W,C] NHWC */
    const float* rt;        /* [B,S,12] */
    const float* inv_depth;
    const float* confidence;
    const float* view_w;
    const float* disp_min;  /* [B] */
    const float* disp_max;  /* [B] */
    float* out_cost;
    float* out_samples;
    int32_t* worklist;      /* dmvs_getcost_bwd_f32 only: scratch of DMVS_GETCOST_WORKLIST_INTS(B,H,W) ints (per-tile fit flags, the list of
                               tiles left to the per-pixel kernel, per-view window boxes), or NULL = per-pixel kernel everywhere */
    int32_t B, S, C, G, n, H, W;
    int32_t vw_shift;
    int32_t cost_cstride, cost_coffset, samp_cstride, samp_coffset;
    float interval, min_radius, max_radius;
    int32_t feat_dtype;     /* DMVS_DTYPE_*: element type / channel order of ref / src (the backward reads plain fp32 NHWC) */
    int32_t tune;           /* ABI 3.  dmvs_getcost_bwd_f32: DMVS_TUNE_BWD_* (0 = default); the forward ignores it */
} dmvs_getcost_desc;

/* ---------------------------------------------------------------------------------------
 * Feature layouts of the quad kernels.  feat_dtype = DMVS_DTYPE_F32: `ref` / `src` are read in the GROUP-INTERLEAVED channel-last layout
 * "NHWC-g4": a texel is C/16 units of 16 floats, unit j = [ch 4j..4j+3 of group 0 | of group 1 | of group 2 | of group 3]
 * (group g = channels g*C/4 .. (g+1)*C/4 - 1 of the reference's NCHW tensor), i.e.
 *     position p of a texel holds channel  c(p) = ((p / 4) % 4) * (C / 4) + (p / 16) * 4 + p % 4,
 * so that the 4 lanes of a pixel fetch 64 contiguous bytes and each receives channels of its own correlation group.
 * The inference engine has FeatureNet's output convolutions emit this layout directly (output-channel permutation of
 * their weights); diffmvs_amd.ops.g4_channels(C) is the permutation for callers that hold plain NHWC tensors.
 * Reduced-precision feature storage (BASELINE.json's bf16 / fp16 configurations): feat_dtype = DMVS_DTYPE_BF16 | _F16 reads ref /
 * src as 16-bit elements in PLAIN NHWC order (a group's C/4 channels are then contiguous already: 8 / 16 / 24 bytes per lane);
 * values are widened to fp32 on arrival, projection, hypotheses, correlation and accumulation stay fp32.
 * DMVS_DTYPE_F32_PLAIN: fp32 in plain NHWC order (what the training graph holds and its backward kernels read). */
/* tune (plane sweep): 0 = the source band of a 16 x 4 pixel tile is staged in LDS (warp_init_band_kernel);
 * DMVS_TUNE_SWEEP_GLOBAL = every texel from global memory (warp_init_quad_kernel, the round-2 form; bit-identical results). */
#define DMVS_TUNE_SWEEP_GLOBAL 0x1
int dmvs_getcost_quad_f32(const dmvs_getcost_desc* d, void* stream);
int dmvs_warp_corr_init_quad_f32(const void* ref, const void* src, const float* rt,
                                 const float* disp_min, const float* disp_max, float* out,
                                 int32_t B, int32_t S, int32_t C, int32_t G, int32_t D,
                                 int32_t H, int32_t W, int32_t Hs, int32_t Ws, int32_t feat_dtype, int32_t tune, void* stream);

/* Stand-alone differentiable_warping (models/module.py:181-218) in the reference's layouts:
 * src [B,C,Hs,Ws] NCHW, rt [B,12] (rot row-major, trans) = src_proj * inverse(ref_proj),
 * depth [B,D,H,W] metric -> out [B,C,D,H,W].  Not used by the model's fused path. */
int dmvs_warp_volume_f32(const float* src, const float* rt, const float* depth, float* out,
                         int32_t B, int32_t C, int32_t D, int32_t H, int32_t W, int32_t Hs, int32_t Ws,
                         void* stream);

/* ---------------------------------------------------------------------------------------
 * Training step: backward of the fused warp / correlation kernels w.r.t. the image features (the
 * reference builds the sampling grid under no_grad and detaches hypotheses and GetCost's view
 * weights: models/module.py:187, :573, models/update.py:442-445).
 *   gref [B,H,W,C] is WRITTEN; gsrc [S][B,Hs,Ws,C] is ACCUMULATED with fp32 atomics (caller zeroes it,
 *   or passes the running gradient of the source features).
 */
/* Channel -> lane mapping of the per-pixel scatter kernels.  Default: a lane owns CONSECUTIVE channels (16-byte loads; each atomic
 * instruction of a flush touches 4 bytes out of every 16 of the texel).  DMVS_TUNE_BWD_INTERLEAVED (dmvs_getcost_desc.tune) /
 * gather = DMVS_BWD_GATHER_INTERLEAVED: 16 lanes per pixel, lane = channel mod 16 -- an atomic instruction covers 64 contiguous bytes of the
 * texel, C*4/64 full requests per (pixel, tap) at the L2's atomic units instead of 4x as many quarter-filled ones.  Same sums in the same
 * per-lane order; the two mappings differ only in how atomics from different pixels interleave (last-bit differences, as between any two runs). */
#define DMVS_TUNE_BWD_INTERLEAVED 0x1
#define DMVS_BWD_GATHER_INTERLEAVED 2
/* gather != 0: per-pixel atomics kernel only (1: consecutive channels per lane, 2: interleaved); 0: LDS-window kernel for C = 48 (the model's stage 1) */
int dmvs_warp_corr_init_bwd_f32(const float* ref, const float* src, const float* rt,
                                const float* disp_min, const float* disp_max, const float* gcor,
                                float* gref, float* gsrc, int32_t B, int32_t S, int32_t C, int32_t G,
                                int32_t D, int32_t H, int32_t W, int32_t Hs, int32_t Ws, int32_t gather, void* stream);
/* gcost [B,G*n,H,W] contiguous; d->out_cost / out_samples / feat_dtype are ignored (features: plain fp32 NHWC).  With d->worklist the
 * LDS-window kernel takes the 16x16 tiles whose source footprints fit its windows (C = 32 | 16) and the per-pixel kernel the rest. */
int dmvs_getcost_bwd_f32(const dmvs_getcost_desc* d, const float* gcost, float* gref, float* gsrc, void* stream);
/* backward of dmvs_view_aggregate_f32 (InitialCost, where the view weights DO require grad, :539-548):
 * gcor [B,S,GD,HW], gw [B,S,HW] are written */
int dmvs_view_aggregate_bwd_f32(const float* cor, const float* w, const float* out, const float* gout,
                                float* gcor, float* gw, int32_t B, int32_t S, int32_t GD, int32_t HW,
                                void* stream);

/* view-weighted aggregation of the per-view volumes (models/module.py:539-548):
 * out[b,g,d,p] = sum_s w[b,s,p] cor[b,s,g,d,p] / (1e-8 + sum_s w[b,s,p]) */
int dmvs_view_aggregate_f32(const float* cor, const float* w, float* out,
                            int32_t B, int32_t S, int32_t GD, int32_t HW, void* stream);

/* PixelViewWeight tail (models/module.py:460-463): out[n,p] = max_d sigmoid(x[n,d,p]) */
int dmvs_sigmoid_max_d_f32(const float* x, float* out, int32_t N, int32_t D, int32_t HW, void* stream);

/* InitialCost epilogue (models/module.py:553-571): softmax over D, expectation index,
 * normalised depth index/(D-1), metric depth via disp_to_depth, photometric confidence =
 * sum of the 4 probabilities d-1..d+2 around floor(index).
 *   logits [B,D,HW] -> norm_depth [B,HW], depth [B,HW], conf [B,HW] */
int dmvs_depth_regress_f32(const float* logits, const float* disp_min, const float* disp_max,
                           float* norm_depth, float* depth, float* conf,
                           int32_t B, int32_t D, int32_t HW, void* stream);

/* upsample_depth (models/module.py:237-248) fused with disp_to_depth (:220-227):
 * convex combination of the 3x3 neighbourhood with softmax(mask) weights.
 *   inv [B,H,W], mask [B,9*r*r,H,W] -> out_inv [B,rH,rW] (may be NULL), out_depth [B,rH,rW] */
int dmvs_convex_upsample_f32(const float* inv, const float* mask, const float* disp_min,
                             const float* disp_max, float* out_inv, float* out_depth,
                             int32_t B, int32_t H, int32_t W, int32_t ratio, void* stream);

/* The mask head's last layer + upsample_depth in one launch (DiffMVS, ratio 4): logits = post_scale * (W x + bias) -- the 1x1 convolution
 * 64 -> 144 of models/update.py:335-339 with `mask = .25 * self.mask(context)` (:473) -- then upsample_depth (models/module.py:237-248)
 * and disp_to_depth (:220-227) exactly as dmvs_convex_upsample_f32; the 144-channel mask is never written to memory.  Results are those
 * of dmvs_conv2d_f32 followed by dmvs_convex_upsample_f32 bit for bit.
 *   x [B,64,H,W] (the ReLU output of the head's 3x3 layer), weight [64][1][144] (kernel layout), bias [144] or NULL, inv [B,H,W]
 *   -> out_inv [B,4H,4W] (may be NULL), out_depth [B,4H,4W] (may be NULL; not both).  cin must be 64 and cout_pad 144 (DMVS_EINVAL otherwise:
 *   other ratios / widths take the two entry points). */
int dmvs_mask_upsample4_f32(const float* x, const float* weight, const float* bias, float post_scale, const float* inv,
                            const float* disp_min, const float* disp_max, float* out_inv, float* out_depth,
                            int32_t B, int32_t cin, int32_t cout_pad, int32_t H, int32_t W, void* stream);

/* GroupNorm statistics + fused apply of Block.forward (models/update.py:124-133):
 *   y = silu( gn(x) * (scale+1) + shift ) [+ residual]
 * x [B,C,HW]; gamma,beta [C]; scale_shift [B,2C] (scale then shift) or NULL; residual or NULL.
 * stats: caller-provided scratch of B*groups*2 8-byte slots (opaque, see gn_stats above).  In-place (y == x) is allowed. */
int dmvs_groupnorm_silu_f32(const float* x, const float* gamma, const float* beta,
                            const float* scale_shift, const float* residual, float* y,
                            double* stats, int32_t B, int32_t C, int32_t HW, int32_t groups,
                            float eps, void* stream);

/* The apply half alone, for statistics that were accumulated by dmvs_conv2d_f32 (gn_stats). */
int dmvs_groupnorm_apply_f32(const float* x, const float* gamma, const float* beta,
                             const float* scale_shift, const float* residual, float* y,
                             const double* stats, int32_t B, int32_t C, int32_t HW, int32_t groups,
                             float eps, void* stream);

/* Refinement bookkeeping (models/update.py:479-483, :496-502):
 *   new = clamp(inv + delta_in (+ update), 0, 1);  delta_out = new - inv
 * update may be NULL (first step: delta_in = scale * noise).  `new` is written with channel
 * stride/offset so that it can land directly inside the Unet input tensor. */
int dmvs_delta_update_f32(const float* inv, const float* delta_in, const float* update,
                          float delta_in_scale, float* delta_out, float* new_inv,
                          float* new_inv2, int32_t new2_cstride, int32_t new2_coffset,
                          int32_t B, int32_t HW, void* stream);

/* element-wise helpers:  depth <-> normalised inverse depth (models/module.py:220-235),
 * activations on a channel slice, nearest upsampling by an integer factor. */
enum { DMVS_EW_DEPTH_TO_DISP = 0, DMVS_EW_DISP_TO_DEPTH = 1 };
int dmvs_depth_convert_f32(const float* in, const float* disp_min, const float* disp_max,
                           float* out, int32_t mode, int32_t B, int32_t HW, void* stream);
/* out[b, out_coffset + c, p] = act(in[b, in_coffset + c, p]) for c < C */
int dmvs_act_slice_f32(const float* in, float* out, int32_t act, int32_t B, int32_t C, int32_t HW,
                       int32_t in_cstride, int32_t in_coffset, int32_t out_cstride, int32_t out_coffset,
                       void* stream);
int dmvs_upsample_nearest_f32(const float* in, float* out, int32_t N, int32_t H, int32_t W,
                              int32_t factor, void* stream);
/* Backward of dmvs_groupnorm_silu_f32 without the residual (Block.forward in training, models/update.py:124-133):
 * `stats` [B*groups*2] doubles as left by the forward; dx [B,C,HW]; dgamma / dbeta [C]; dscale_shift [B,2C] (required when
 * scale_shift is given); `workspace`: caller scratch of dmvs_groupnorm_silu_bwd_workspace_f32() bytes.  No atomics. */
int dmvs_groupnorm_silu_bwd_workspace_f32(int32_t B, int32_t C, int32_t HW, int64_t* bytes);
int dmvs_groupnorm_silu_bwd_f32(const float* x, const float* dy, const float* gamma, const float* beta, const float* scale_shift,
                                const double* stats, float* dx, float* dgamma, float* dbeta, float* dscale_shift,
                                float* workspace, int64_t workspace_bytes, int32_t B, int32_t C, int32_t HW, int32_t groups,
                                float eps, void* stream);

/* ---------------------------------------------------------------------------------------
 * Training-mode BatchNorm (+ optional ReLU) on [B, C, S] fp32 (S = H*W or D*H*W).  Replaces nn.BatchNorm2d/3d in
 * train mode inside module.Conv2d / Conv3d / ConvBnReLU / ConvBn            models/module.py:24-58, :60-96, :279-301.
 *   fwd: mean/var over (rows of the view, S) per channel (biased var for the normalisation),
 *        y = act((x-mean)*rstd*gamma+beta); running_mean/var (may both be NULL) <- (1-momentum)*running +
 *        momentum*(mean | unbiased var), once per view in view order; save_mean / save_rstd [views][C] for the backward.
 *   bwd: dz = dy * [y > 0 if act == RELU];  per view dbeta_v = sum dz, dgamma_v = sum dz*xhat,
 *        dx = gamma*rstd*(dz - dbeta_v/N - xhat*dgamma_v/N), N = (B/views)*S;  dgamma / dbeta [C] = sums over views.
 * `views` (B % views == 0) batches that many independent BatchNorm calls: the reference runs FeatureNet once per image
 * of the view stack and PixelViewWeight once per source view (diffusion.py:156-157, module.py:533); row b belongs to
 * view b / (B/views) if view_major, else b % views.  act in {DMVS_ACT_NONE, DMVS_ACT_RELU}; x / y / dy / dx 16-byte
 * aligned; `workspace`: caller-owned scratch of dmvs_batchnorm_workspace_f32() bytes (per-chunk partial sums folded
 * in double, no atomics). */
int dmvs_batchnorm_workspace_f32(int32_t B, int32_t C, int32_t S, int32_t views, int64_t* bytes);
int dmvs_batchnorm_train_fwd_f32(const float* x, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, float* y, float* save_mean, float* save_rstd, float* workspace,
                                 int64_t workspace_bytes, int32_t B, int32_t C, int32_t S, int32_t views, int32_t view_major,
                                 float momentum, float eps, int32_t act, void* stream);
int dmvs_batchnorm_train_bwd_f32(const float* x, const float* dy, const float* gamma, const float* beta,
                                 const float* save_mean, const float* save_rstd, float* dx, float* dgamma, float* dbeta,
                                 float* workspace, int64_t workspace_bytes, int32_t B, int32_t C, int32_t S, int32_t views,
                                 int32_t view_major, int32_t act, void* stream);

/* ---------------------------------------------------------------------------------------
 * Depth-map fusion: geometric consistency of one reference depth map against S source depth maps (reference
 * filter.py:8-93 reproject_with_depth + check_geometric_consistency, :230-259 the dynamic variant).  Per reference pixel and
 * source view: reference depth -> source pixel -> source depth sampled like cv2.remap(INTER_LINEAR, constant 0 border,
 * 1/32-pixel coordinate quantisation) -> back into the reference view; level l accepts the pair when
 *     reprojection distance < pix_thres[l]  and  |depth_reproj - depth_ref| / depth_ref < rel_thres[l]
 *     (and, if use_range, range_min < depth_ref < range_max -- filter.py:89).
 *   depth_ref [H,W], depth_src [S,Hs,Ws] fp32
 *   mats [S][DMVS_GEO_MATS_FLOATS] fp32, per source view row-major: inv(K_ref) 3x3 | (E_src inv(E_ref))[:3,:4] |
 *        K_src 3x3 | inv(K_src) 3x3 | (E_ref inv(E_src))[:3,:4] | K_ref 3x3, composed by the caller in fp32 as the
 *        reference's np.linalg.inv / np.matmul on its fp32 camera files do
 *   level_counts [L,H,W] int32: number of source views accepted at level l;  depth_sum [H,W]: sum over the views accepted
 *        at the LAST level of the reprojected depth (filter.py:91, :257), accumulated in view order in fp32.
 * L = 1 is filter_depth (DTU / ETH3D), L = 11 - dh_view_num the Tanks&Temples dynamic check. */
#define DMVS_GEO_MATS_FLOATS 60
#define DMVS_GEO_MAX_LEVELS 12
int dmvs_geo_consistency_f32(const float* depth_ref, const float* depth_src, const float* mats, const double* pix_thres,
                             const float* rel_thres, int32_t L, int32_t use_range, float range_min, float range_max,
                             int32_t* level_counts, float* depth_sum, int32_t S, int32_t H, int32_t W, int32_t Hs, int32_t Ws,
                             void* stream);

/* ---------------------------------------------------------------------------------------
 * Training-step tail on one flat fp32 parameter bucket (the buffer RCCL all-reduces).  Replaces
 * torch.nn.utils.clip_grad_norm_(model.parameters(), 2.0) + AdamW.step()   train.py:200-203, :321-326.
 * dmvs_sumsq_f32: *out (a device double) = sum g[i]^2;  `g` must be 16-byte aligned.
 * dmvs_adamw_step_f32: g' = g * grad_scale (1/world_size of the data-parallel average), further scaled by
 *   min(1, max_norm / (grad_scale*sqrt(*sumsq) + 1e-6)) when sumsq != NULL; decoupled weight decay,
 *   bias correction with `step` (1-based), as torch.optim.AdamW(amsgrad=False). */
int dmvs_sumsq_f32(const float* g, int64_t n, double* out, void* stream);
int dmvs_adamw_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                        float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                        const double* sumsq, float max_norm, void* stream);
/* NCHW -> NHWC for features that did not come out of dmvs_conv2d_f32 channel-last */
int dmvs_nchw_to_nhwc_f32(const float* in, float* out, int32_t B, int32_t C, int32_t HW, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DMVS_H */
