"""Training-step plumbing: torch.autograd Functions whose forward AND backward are libdmvs_hip.so kernels.

Nodes: convolution (forward, input gradient = the forward kernel on flipped weights, weight gradient = MFMA
reduction over pixels), the fused warp / correlation / aggregation kernels, training-mode BatchNorm(+ReLU) and
GroupNorm + scale/shift + SiLU (norm.hip).  The remaining element-wise glue of the training graph (head
activations, softmax regression, convex upsampling, concatenations, weight standardisation) is ATen device ops.
The train branch of the update block is diffmvs_amd/train.py, the data-parallel step diffmvs_amd/trainer.py."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops as K
from .ops import Ops, PackedConv, pack_conv2d, pack_conv3d


def _packed(cache, key, pins, make):
    """per-step cache of packed weights, keyed by tensor identity; the entry pins the tensors so that an id is not reused"""
    if cache is None:
        return make()
    hit = cache.get(key)
    if hit is None:
        hit = cache[key] = (make(), pins)
    return hit[0]


class _Conv2dFn(torch.autograd.Function):
    """conv2d(x, weight, bias) with stride 1|2 and the fused input modes of the forward kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, ops: Ops, stride, pad, in_mode, cache):
        pc = _packed(cache, ("fwd", id(weight), id(bias), stride, pad), (weight, bias),
                     lambda: pack_conv2d(weight, bias, stride=stride, pad=pad))
        out = ops.conv2d(pc, x, in_mode=in_mode)
        ctx.save_for_backward(x, weight)
        ctx.ops, ctx.pc, ctx.in_mode, ctx.has_bias, ctx.cache = ops, pc, in_mode, bias is not None, cache
        # direct accumulation (see backward): only on request of the step's owner (cache["grad_into_bucket"], set by train.forward_train for
        # a model a Trainer has re-homed into its flat bucket), only for LEAF parameters whose .grad is a preallocated contiguous tensor
        ctx.direct = None
        if cache is not None and cache.get("grad_into_bucket") and weight.is_leaf and weight.grad is not None and weight.grad.is_contiguous() and \
                (bias is None or (bias.is_leaf and bias.grad is not None and bias.grad.is_contiguous())):
            ctx.direct = (weight, bias)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        ops, pc, in_mode = ctx.ops, ctx.pc, ctx.in_mode
        g = g.contiguous()
        kh, kw = pc.k
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            # dX = conv(dY (zero-inserted for stride 2), W flipped and cin<->cout transposed), pad' = k-1-pad
            def flipped():
                wb = weight.detach().flip(2, 3).permute(1, 0, 2, 3).contiguous()
                return pack_conv2d(wb, None, stride=1, pad=(kh - 1 - pc.pad[0], kw - 1 - pc.pad[1]))
            pcb = _packed(ctx.cache, ("bwd", id(weight), pc.pad), (weight,), flipped)
            gl = ops.conv2d(pcb, g, in_mode=(K.IN_ZEROINSERT2 if pc.stride == 2 else K.IN_PLAIN))
            if in_mode == K.IN_UPSAMPLE2:
                gx = F.avg_pool2d(gl, 2) * 4.0          # adjoint of nearest x2
            elif in_mode == K.IN_UNSHUFFLE2:
                gx = F.pixel_shuffle(gl, 2)             # adjoint of 'b c (h p1) (w p2) -> b (c p1 p2) h w'
            else:
                gx = gl
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] and ctx.direct is not None:
            # the trainer's flat gradient bucket: the fold kernel adds into the parameter's .grad view (one read-modify-write per
            # element) and autograd gets None -- instead of a fresh tensor per use + one AccumulateGrad add kernel per use
            # (~1300 add launches of ~5 us per cfg4 step: the update block applies every layer 3 times)
            w_leaf, b_leaf = ctx.direct
            ops.conv2d_wgrad(pc, x, g, in_mode=in_mode, want_bias=want_b, into_gw=w_leaf.grad, into_gb=b_leaf.grad if want_b else None)
        elif ctx.needs_input_grad[1]:
            gw = ops.conv2d_wgrad(pc, x, g, in_mode=in_mode, want_bias=want_b)   # bias gradient rides in a spare MFMA column
            if want_b:
                gw, gb = gw
        elif want_b:
            gb = g.sum((0, 2, 3))
        return gx, gw, gb, None, None, None, None, None


def conv2d(ops: Ops, x, weight, bias=None, stride=1, pad=0, in_mode=K.IN_PLAIN, cache=None):
    """cache: a dict that lives for ONE training step (train._Net): the packed forward weights and the flipped / transposed
    weights of the input gradient are built once per weight tensor and step instead of once per use -- the update block applies
    the same layers in every GRU iteration (3 uses per step at cfg4: ~4 tiny permute / flip / copy kernels per use saved)."""
    pad = (pad, pad) if isinstance(pad, int) else tuple(pad)
    return _Conv2dFn.apply(x.contiguous(), weight, bias, ops, stride, pad, in_mode, cache)


class _Conv3dFn(torch.autograd.Function):
    """3x3x3 conv (stride 1|2, padding 1) or stride-2 transposed conv (padding 1, output_padding 1), optional bias."""

    @staticmethod
    def forward(ctx, x, weight, bias, ops: Ops, stride, transposed):
        pc = pack_conv3d(weight, bias, stride=stride, transposed=transposed)
        ctx.save_for_backward(x, weight)
        ctx.ops, ctx.stride, ctx.transposed, ctx.has_bias = ops, stride, transposed, bias is not None
        return ops.conv3d(pc, x)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        ops, stride, transposed = ctx.ops, ctx.stride, ctx.transposed
        g = g.contiguous()
        w = weight.detach()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if transposed:      # adjoint of the transposed conv = the stride-2 conv with the same weight tensor
                gx = ops.conv3d(pack_conv3d(w, stride=2), g)
            elif stride == 2:   # adjoint of the stride-2 conv = the transposed conv with the same weight tensor
                gx = ops.conv3d(pack_conv3d(w, stride=2, transposed=True), g)
            else:
                gx = ops.conv3d(pack_conv3d(w.flip(2, 3, 4).permute(1, 0, 2, 3, 4).contiguous()), g)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if transposed:      # roles swapped: "input" = grad_out, "output gradient" = x; result is [cin_t, cout_t, 3,3,3]
                gw = ops.conv3d_wgrad(g, x, cout=x.shape[1], stride=2)
            else:
                gw = ops.conv3d_wgrad(x, g, cout=g.shape[1], stride=stride, want_bias=want_b)
                if want_b:
                    gw, gb = gw
        if want_b and gb is None:
            gb = g.sum((0, 2, 3, 4))
        return gx, gw, gb, None, None, None


def conv3d(ops: Ops, x, weight, bias=None, stride=1, transposed=False):
    return _Conv3dFn.apply(x.contiguous(), weight, bias, ops, stride, transposed)


class _WarpCorrInitFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ref, src, rt, disp_min, disp_max, ops: Ops, D, G):
        ctx.save_for_backward(ref, src, rt, disp_min, disp_max)
        ctx.ops = ops
        return ops.warp_corr_init_quad(ref, src, rt, disp_min, disp_max, D, G, plain=True)      # plain NHWC fp32: the order the backward kernels read

    @staticmethod
    def backward(ctx, g):
        ref, src, rt, disp_min, disp_max = ctx.saved_tensors
        gref, gsrc = ctx.ops.warp_corr_init_bwd(ref, src, rt, disp_min, disp_max, g.contiguous())
        return gref, gsrc, None, None, None, None, None, None


def warp_corr_init(ops: Ops, ref, src, rt, disp_min, disp_max, D, G=4):
    """ref [B,H,W,C], src [S,B,Hs,Ws,C] (NHWC, both may require grad) -> cor [B,S,G,D,H,W]"""
    return _WarpCorrInitFn.apply(ref.contiguous(), src.contiguous(), rt, disp_min, disp_max, ops, D, G)


class _GetCostFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ref, src, rt, inv_depth, confidence, view_w, disp_min, disp_max, ops: Ops, n, interval, rmin, rmax, shift, G):
        cost, samples = ops.getcost_quad(ref, src, rt, inv_depth, confidence, view_w, disp_min, disp_max, n, interval, rmin, rmax,
                                         shift, G=G, plain=True)
        ctx.save_for_backward(ref, src, rt, inv_depth, confidence, view_w, disp_min, disp_max)
        ctx.ops, ctx.meta = ops, (n, interval, rmin, rmax, shift, G)
        ctx.mark_non_differentiable(samples)
        return cost, samples

    @staticmethod
    def backward(ctx, g, _gs):
        ref, src, rt, inv_depth, confidence, view_w, disp_min, disp_max = ctx.saved_tensors
        n, interval, rmin, rmax, shift, G = ctx.meta
        gref, gsrc = ctx.ops.getcost_bwd(ref, src, rt, inv_depth, confidence, view_w, disp_min, disp_max, n, interval, rmin,
                                         rmax, shift, g.contiguous(), G=G)
        return (gref, gsrc) + (None,) * 13


def getcost(ops: Ops, ref, src, rt, inv_depth, confidence, view_w, disp_min, disp_max, n, interval, rmin, rmax, vw_shift, G=4):
    """GetCost with gradients to the image features only (hypotheses / view weights are detached in the reference)."""
    return _GetCostFn.apply(ref.contiguous(), src.contiguous(), rt, inv_depth, confidence, view_w, disp_min, disp_max, ops, n,
                            interval, rmin, rmax, vw_shift, G)


class _ViewAggregateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cor, w, ops: Ops):
        out = ops.view_aggregate(cor, w)
        ctx.save_for_backward(cor, w, out)
        ctx.ops = ops
        return out

    @staticmethod
    def backward(ctx, g):
        cor, w, out = ctx.saved_tensors
        gcor, gw = ctx.ops.view_aggregate_bwd(cor, w, out, g.contiguous())
        return gcor, gw, None


def view_aggregate(ops: Ops, cor, w):
    return _ViewAggregateFn.apply(cor.contiguous(), w.contiguous(), ops)


class _BatchNormActFn(torch.autograd.Function):
    """BatchNorm in training mode (batch statistics per view, running stats updated in place) with the ReLU fused."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, ops: Ops, momentum, eps, relu, views, view_major):
        act = K.ACT_RELU if relu else K.ACT_NONE
        y, mean, rstd = ops.batchnorm_train_fwd(x, gamma.detach(), beta.detach(), running_mean, running_var, momentum, eps, act,
                                                views, view_major)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)     # x only: the ReLU mask is recomputed from it
        ctx.ops, ctx.act, ctx.views, ctx.view_major = ops, act, views, view_major
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        dx, dgamma, dbeta = ctx.ops.batchnorm_train_bwd(x, g.contiguous(), gamma.detach(), beta.detach(), mean, rstd, ctx.act,
                                                        ctx.views, ctx.view_major)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None


def batchnorm_act(ops: Ops, x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, relu=True, views=1,
                  view_major=True):
    """`views` independent BatchNorm calls batched in one tensor (rows view-major [V*B] or view-minor [B*V])"""
    return _BatchNormActFn.apply(x.contiguous(), gamma, beta, running_mean, running_var, ops, momentum, eps, relu, views,
                                 view_major)


class _GroupNormSiLUFn(torch.autograd.Function):
    """silu(GroupNorm(x) * (scale + 1) + shift) of the diffusion Unet's Block (update.py:124-133), HIP forward and backward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, scale_shift, ops: Ops, groups, eps):
        y, stats = ops.groupnorm_silu_train(x, gamma.detach(), beta.detach(), groups, None if scale_shift is None else scale_shift.detach(), eps)
        ctx.save_for_backward(x, gamma, beta, scale_shift, stats)
        ctx.ops, ctx.groups, ctx.eps = ops, groups, eps
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, beta, scale_shift, stats = ctx.saved_tensors
        dx, dgamma, dbeta, dss = ctx.ops.groupnorm_silu_bwd(x, g.contiguous(), gamma.detach(), beta.detach(), ctx.groups, stats,
                                                            None if scale_shift is None else scale_shift.detach(), ctx.eps)
        return dx, dgamma, dbeta, dss, None, None, None


def groupnorm_silu(ops: Ops, x, gamma, beta, groups, scale_shift=None, eps=1e-5):
    """scale_shift [B, 2C] = (scale | shift) as produced by the block's time-embedding Linear, or None"""
    ss = None if scale_shift is None else scale_shift.contiguous()
    return _GroupNormSiLUFn.apply(x.contiguous(), gamma, beta, ss, ops, groups, eps)
