"""Training forward of DiffMVS / CasDiffMVS (model.train(), reference models/diffusion.py:139-295 with the
train branches of InitialCost (module.py:539-541) and DiffusionUpdateBlockDepth (update.py:423-464)).

Built as a torch.autograd graph whose heavy nodes are libdmvs_hip.so kernels, forward AND backward
(diffmvs_amd/autograd.py): every 2-D / 3-D convolution (MFMA implicit GEMM, MFMA weight gradient), the fused
homography-warp + group-correlation volumes, GetCost, the view aggregation, training-mode BatchNorm(+ReLU) with
per-view statistics and GroupNorm + scale/shift + SiLU (norm.hip).  The element-wise glue (head activations,
softmax regression, convex upsampling, concatenations, weight standardisation) is ATen device ops.

Semantics that matter for parity with the reference's gradients:
  * FeatureNet is applied per view (diffusion.py:156-157) -> BatchNorm batch statistics per view, running stats
    updated V times; PixelViewWeight likewise per source view (module.py:533);
  * the sampling grid, the depth hypotheses, the view weights handed to GetCost and every iterate entering a GRU
    step are detached (module.py:187, :573, update.py:442-445), so gradients reach the image features, the
    context/hidden path and the networks, never the geometry;
  * random draws: one t ~ U[0,T) per batch item and one noise tensor per refinement stage (update.py:432-433),
    injectable for parity through model.t_source / model.noise_source.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import autograd as A
from . import ops as K
from .ops import Ops

_RATIOS = [4, 2, 1]
_MULTS = [(1,), (1, 2), (1, 2, 4)]


class _Net:
    """parameter / buffer lookup by the reference's checkpoint key, bound to one Ops"""

    def __init__(self, model, ops: Ops):
        self.o = ops
        self.p = dict(model.named_parameters())
        self.b = dict(model.named_buffers())
        # per-STEP cache (a _Net lives for one forward_train call and the backward of its graph): packed weights per weight tensor
        # (autograd._packed), and the sub-graphs that do not depend on the GRU iteration -- standardised weights, time embeddings --
        # built once and shared by the iterations (autograd sums their gradients before the shared node's backward runs once)
        self.cache = {}

    def has(self, k):
        return k in self.p

    # ---- convolution flavours
    def conv2(self, x, k, stride=1, pad=0, in_mode=K.IN_PLAIN):
        return A.conv2d(self.o, x, self.p[k + ".weight"], self.p.get(k + ".bias"), stride=stride, pad=pad, in_mode=in_mode, cache=self.cache)

    def bn(self, x, k, relu, views=1, view_major=True):
        """BatchNorm in training mode (+ReLU): batch statistics, running stats updated in place (momentum 0.1).
        `views` independent calls of the reference batched in one tensor (per-view statistics, `views` updates)."""
        y = A.batchnorm_act(self.o, x, self.p[k + ".weight"], self.p[k + ".bias"], self.b[k + ".running_mean"],
                            self.b[k + ".running_var"], 0.1, 1e-5, relu, views, view_major)
        self.b[k + ".num_batches_tracked"].add_(views)
        return y

    def cbr2(self, x, k, stride=1, pad=1, relu=True, views=1):
        """module.Conv2d / ConvBnReLU / ConvBn (module.py:24-58, :279-301)"""
        y = A.conv2d(self.o, x, self.p[k + ".conv.weight"], None, stride=stride, pad=pad, cache=self.cache)
        return self.bn(y, k + ".bn", relu, views)

    def cbr3(self, x, k, stride=1, transposed=False, views=1, view_major=True):
        y = A.conv3d(self.o, x, self.p[k + ".conv.weight"], None, stride=stride, transposed=transposed)
        return self.bn(y, k + ".bn", True, views, view_major)


def disp_to_depth(disp, min_depth, max_depth):
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    scaled = (min_disp + (max_disp - min_disp) * disp).clamp(min=1e-6)
    return scaled, 1 / scaled


def depth_to_disp(depth, min_depth, max_depth):
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    return (1 / depth - min_disp) / (max_disp - min_disp)


def feature_net(n: _Net, x, views, p="feature"):
    """FeatureNet on the whole view stack at once: x [V*B,3,H,W] view-major.  The reference calls it once per view
    (diffusion.py:156-157); BatchNorm keeps per-view statistics, everything else is per-sample anyway."""
    V = views
    c0 = n.cbr2(n.cbr2(x, p + ".conv0.0", views=V), p + ".conv0.1", views=V)
    c1 = n.cbr2(n.cbr2(n.cbr2(c0, p + ".conv1.0", 2, 2, views=V), p + ".conv1.1", views=V), p + ".conv1.2", views=V)
    c2 = n.cbr2(n.cbr2(n.cbr2(c1, p + ".conv2.0", 2, 2, views=V), p + ".conv2.1", views=V), p + ".conv2.2", views=V)
    c3 = n.cbr2(n.cbr2(n.cbr2(c2, p + ".conv3.0", 2, 2, views=V), p + ".conv3.1", views=V), p + ".conv3.2", views=V)
    out = {"stage1": n.conv2(c3, p + ".out1")}
    intra = F.interpolate(c3, scale_factor=2, mode="nearest") + n.conv2(c2, p + ".inner1")
    out["stage2"] = n.conv2(intra, p + ".out2", pad=1)
    if n.has(p + ".out3.weight"):
        intra = F.interpolate(intra, scale_factor=2, mode="nearest") + n.conv2(c1, p + ".inner2")
        out["stage3"] = n.conv2(intra, p + ".out3", pad=1)
    return out


def context_net(n: _Net, x, p="context"):
    def block(x, q, stride):
        y = n.cbr2(n.cbr2(x, q + ".conv1", stride), q + ".conv2", relu=False)
        if stride != 1:
            x = n.cbr2(x, q + ".downsample", stride, relu=False)
        return F.relu(x + y)
    out = {}
    x = n.cbr2(x, p + ".conv1")
    x = block(block(x, p + ".layer1.0", 2), p + ".layer1.1", 1)
    if n.has(p + ".output3.weight"):
        out["stage3"] = n.conv2(x, p + ".output3", pad=1)
    x = block(block(x, p + ".layer2.0", 2), p + ".layer2.1", 1)
    out["stage2"] = n.conv2(x, p + ".output2", pad=1)
    x = block(block(x, p + ".layer3.0", 2), p + ".layer3.1", 1)
    out["stage1"] = n.conv2(x, p + ".output1", pad=1)
    return out


def mask_head(n: _Net, context, p):
    return 0.25 * n.conv2(F.relu(n.conv2(context, p + ".0", pad=1)), p + ".2")


def upsample_depth(depth, mask, ratio):
    N, _, H, W = depth.shape
    m = torch.softmax(mask.view(N, 1, 9, ratio, ratio, H, W), dim=2)
    nb = F.unfold(depth, [3, 3], padding=1).view(N, 1, 9, 1, 1, H, W)
    return (m * nb).sum(2).permute(0, 1, 4, 2, 5, 3).reshape(N, ratio * H, ratio * W)


def _nhwc(f, B):
    """[V*B,C,h,w] view-major -> ref [B,h,w,C], src [S,B,h,w,C] (layout of the warp kernels; autograd-tracked)"""
    nhwc = f.permute(0, 2, 3, 1).contiguous()
    return nhwc[:B], nhwc[B:].view(-1, B, *nhwc.shape[1:])


def initial_cost(n: _Net, feat, B, context, rt, disp_min, disp_max, dmin, dmax, D, G, p="depthnet"):
    """InitialCost.forward, training branch (module.py:487-573)"""
    o = n.o
    ref, src = _nhwc(feat, B)
    _, H, W, _ = ref.shape
    S = src.shape[0]
    mask = mask_head(n, context, p + ".mask")
    cor = A.warp_corr_init(o, ref, src, rt, disp_min, disp_max, D, G)           # [B,S,G,D,H,W]
    # PixelViewWeight on all source views at once (rows b*S+s: view-minor); BatchNorm statistics per view as in the
    # reference's per-view calls (module.py:533)
    x = n.cbr3(cor.view(B * S, G, D, H, W), p + ".pixel_view_weight.conv.0", views=S, view_major=False)
    x = A.conv3d(o, x, n.p[p + ".pixel_view_weight.conv.1.weight"], n.p[p + ".pixel_view_weight.conv.1.bias"])
    vw = torch.sigmoid(x.squeeze(1)).max(dim=1)[0].view(B, S, H, W)
    agg = A.view_aggregate(o, cor, vw)
    r = p + ".cost_regularization"
    c1 = n.cbr3(n.cbr3(agg, r + ".conv0"), r + ".conv1")
    c3 = n.cbr3(n.cbr3(c1, r + ".conv2", 2), r + ".conv3")
    x = n.cbr3(n.cbr3(c3, r + ".conv4", 2), r + ".conv5")
    x = c3 + n.cbr3(x, r + ".conv6", 2, transposed=True)
    x = c1 + n.cbr3(x, r + ".conv7", 2, transposed=True)
    pre = A.conv3d(o, x, n.p[r + ".prob.weight"], None).squeeze(1)
    prob = F.softmax(pre, dim=1)
    idx = torch.arange(D, dtype=torch.float32, device=prob.device).view(1, D, 1, 1)
    index = (idx * prob).sum(1, keepdim=True)
    nd = index / (D - 1.0)
    depth = disp_to_depth(nd, dmin, dmax)[1].squeeze(1)
    with torch.no_grad():
        padded = F.pad(prob, (0, 0, 0, 0, 1, 2))
        sum4 = padded[:, 0:D] + padded[:, 1:D + 1] + padded[:, 2:D + 2] + padded[:, 3:D + 3]
        conf = torch.gather(sum4, 1, index.long().clamp(0, D - 1))
    return mask, nd, depth, vw.detach(), conf


# ------------------------------------------------------------------ update block nets
def _ws_conv(n: _Net, x, k):
    """WeightStandardizedConv2d (update.py:81-94); the standardisation is differentiated by autograd"""
    ws = n.cache.get(("ws", k))
    if ws is None:      # once per step: the GRU iterations share the standardised tensor (and its packed forms)
        w = n.p[k + ".weight"]
        mean = w.mean(dim=(1, 2, 3), keepdim=True)
        var = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
        ws = n.cache[("ws", k)] = (w - mean) * torch.rsqrt(var + 1e-5)
    return A.conv2d(n.o, x, ws, n.p[k + ".bias"], pad=1, cache=n.cache)


def _block(n: _Net, x, k, scale_shift=None):
    """Block.forward (update.py:124-133): WS conv -> GroupNorm(4) -> x*(scale+1)+shift -> SiLU; scale_shift [B, 2C] or None"""
    return A.groupnorm_silu(n.o, _ws_conv(n, x, k + ".proj"), n.p[k + ".norm.weight"], n.p[k + ".norm.bias"], 4, scale_shift, 1e-5)


def resnet_block(n: _Net, x, k, t_emb=None):
    ss = None
    if t_emb is not None and n.has(k + ".mlp.1.weight"):
        ss = n.cache.get(("ss", k, id(t_emb)))
        if ss is None:      # [B, 2C] = (scale | shift): a function of the time embedding only, the same in every GRU iteration
            ss = F.linear(F.silu(t_emb), n.p[k + ".mlp.1.weight"], n.p[k + ".mlp.1.bias"])
            n.cache[("ss", k, id(t_emb))] = ss
    h = _block(n, _block(n, x, k + ".block1", ss), k + ".block2")
    res = n.conv2(x, k + ".res_conv") if n.has(k + ".res_conv.weight") else x
    return h + res


def sep_conv_gru(n: _Net, k, h, x):
    for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(n.conv2(hx, f"{k}.convz{sfx}", pad=pad))
        r = torch.sigmoid(n.conv2(hx, f"{k}.convr{sfx}", pad=pad))
        q = torch.tanh(n.conv2(torch.cat([r * h, x], 1), f"{k}.convq{sfx}", pad=pad))
        h = (1 - z) * h + z * q
    return h


def unet(n: _Net, k, x, hidden, t, dim, L):
    x = n.conv2(x, k + ".init_conv", pad=3)
    r = x
    te = n.cache.get(("te", k, id(t)))
    if te is None:      # the time embedding of the stage's t (update.py:432: one draw per stage), shared by the GRU iterations
        half = dim // 2
        freqs = torch.exp(torch.arange(half, device=x.device) * -(math.log(10000) / (half - 1)))
        e = t[:, None] * freqs[None, :]
        e = torch.cat((e.sin(), e.cos()), -1)
        e = F.gelu(F.linear(e, n.p[k + ".time_mlp.1.weight"], n.p[k + ".time_mlp.1.bias"]))
        te = F.linear(e, n.p[k + ".time_mlp.3.weight"], n.p[k + ".time_mlp.3.bias"])
        n.cache[("te", k, id(t))] = te
        n.cache[("pin", id(t))] = t          # (keeps id(t) from being reused while the entry lives)
    skips = []
    for i in range(L):
        x = resnet_block(n, x, f"{k}.downs.{i}.0", te)
        skips.append(x)
        if i < L - 1:
            x = n.conv2(x, f"{k}.downs.{i}.1.1", in_mode=K.IN_UNSHUFFLE2)
        else:
            x = n.conv2(x, f"{k}.downs.{i}.1", pad=1)
    hidden = sep_conv_gru(n, k + ".gru", hidden, x)
    x = resnet_block(n, hidden, k + ".mid", te)
    for i in range(L):
        x = resnet_block(n, torch.cat((x, skips.pop()), 1), f"{k}.ups.{i}.0", te)
        if i < L - 1:
            x = n.conv2(x, f"{k}.ups.{i}.1.1", pad=1, in_mode=K.IN_UPSAMPLE2)
        else:
            x = n.conv2(x, f"{k}.ups.{i}.1", pad=1)
    x = resnet_block(n, torch.cat((x, r), 1), k + ".final_res_block", te)
    return hidden, n.conv2(x, k + ".final_conv"), torch.sigmoid(n.conv2(x, k + ".conf"))


def condition_encoder(n: _Net, k, depth, samples, cost):
    c = F.relu(n.conv2(F.relu(n.conv2(cost, k + ".convc1", pad=1)), k + ".convc2", pad=1))
    d = F.relu(n.conv2(F.relu(n.conv2(samples, k + ".convd1", pad=1)), k + ".convd2", pad=1))
    return torch.cat([F.relu(n.conv2(torch.cat([c, d], 1), k + ".output", pad=1)), depth], 1)


def update_block_train(n: _Net, k, cost_fn, inv_depth, hidden, context, gt_inv_depth, inv_init_depth, *, iters, dim, L,
                       timesteps, scale, t, noise):
    """DiffusionUpdateBlockDepth.forward, training branch (update.py:423-464)"""
    gt_inv_depth = torch.where(torch.isinf(gt_inv_depth), inv_init_depth, gt_inv_depth)
    gt_delta = (gt_inv_depth - inv_depth).detach()
    sh = (t.shape[0], 1, 1, 1)
    a = n.b[k + ".sqrt_alphas_cumprod"].gather(-1, t).reshape(sh)
    s1 = n.b[k + ".sqrt_one_minus_alphas_cumprod"].gather(-1, t).reshape(sh)
    delta = a * gt_delta + s1 * (scale * noise).float()                    # q_sample (update.py:392-399)
    new = torch.clamp(inv_depth + delta, 0, 1)
    delta = new - inv_depth
    inv_list, conf_list, confidence = [], [], None
    for _ in range(iters):
        delta = delta.detach()
        if confidence is not None:
            confidence = confidence.detach()
        new = new.detach()
        cost, samples = cost_fn(new, confidence)
        feat = condition_encoder(n, k + ".encoder", new, samples, cost)
        hidden, upd, confidence = unet(n, k + ".unet", torch.cat([context, feat], 1), hidden, t, dim, L)
        confidence = confidence.squeeze(1)
        delta = delta + upd
        conf_list.append(confidence)
        new = torch.clamp(inv_depth + delta, 0, 1)
        delta = new - inv_depth
        inv_list.append(new)
    return mask_head(n, context, k + ".mask"), hidden, inv_list, conf_list


def forward_train(model, imgs, proj_matrices, depth_values, depth_gt_ms, ops: Ops):
    """-> {"depth": [...all iterates...], "conf": [...], "photometric_confidence": [...]} with autograd history"""
    a = model.args
    n = _Net(model, ops)
    # a Trainer that owns the parameters' .grad views asks the convolution nodes to add weight gradients straight into them
    n.cache["grad_into_bucket"] = bool(getattr(model, "grad_into_flat_bucket", False))
    o = ops
    dev = o.device
    t_source = getattr(model, "t_source", None) or (lambda B, T, device: torch.randint(0, T, (B,), device=device).long())
    noise_source = getattr(model, "noise_source", None) or (lambda shape, device: torch.randn(shape, device=device))
    cas = a.stage_iters[2] != 0
    up_ratio = 2 if cas else 4
    disp_min = depth_values[:, 0].float().to(dev).view(-1, 1, 1, 1)
    disp_max = depth_values[:, -1].float().to(dev).view(-1, 1, 1, 1)
    dmax, dmin = 1.0 / disp_min, 1.0 / disp_max
    kmin, kmax = (1.0 / dmax).reshape(-1).contiguous(), (1.0 / dmin).reshape(-1).contiguous()   # kernel-side disp range
    interval = 1.0 / depth_values.size(1)

    V, B = len(imgs), imgs[0].shape[0]
    feats = feature_net(n, torch.cat([im.to(dev).float() for im in imgs], 0), V)
    ctx = context_net(n, imgs[0].to(dev).float())
    depths, confs, confs_full = [], [], []
    view_w = init_depth = None
    for s in range(3):
        if a.stage_iters[s] == 0:
            continue
        name = f"stage{s + 1}"
        inv_gt = depth_to_disp(depth_gt_ms[name].to(dev).unsqueeze(1), dmin, dmax) if s > 0 else None
        fs = feats[name]
        _, _, H, W = fs.shape
        rt = o.compose_proj(proj_matrices[name].to(dev).float().contiguous())
        if s == 0:
            mask, inv_depth, init_depth, view_w, conf = initial_cost(
                n, fs, B, torch.relu(ctx[name]), rt, kmin, kmax, dmin, dmax, a.numdepth_initial, a.cost_dim_stage[0])
            depths.append(init_depth)
            confs_full.append(F.interpolate(conf, scale_factor=8, mode="nearest").squeeze(1))
            up = upsample_depth(inv_depth, mask, 2).unsqueeze(1)
            depths.append(disp_to_depth(up, dmin, dmax)[1].squeeze(1))
            continue
        cur = depths[-1].unsqueeze(1).detach()
        inv_cur = depth_to_disp(cur, dmin, dmax)
        vw = view_w                                                     # nearest-upsampled on the fly by the kernel
        hd, cd = a.hidden_dim[s], a.context_dim[s]
        hidden, context = torch.split(ctx[name], [hd, cd], dim=1)
        hp = f"hidden_init.{s - 1}"
        hidden = n.cbr2(hidden, hp + ".0", 2)
        if s == 2:
            hidden = n.cbr2(hidden, hp + ".1", 2)
        hidden = torch.tanh(A.conv2d(o, hidden, n.p[f"{hp}.{s}.weight"], None, pad=1))
        context = torch.relu(context)
        inv_init = depth_to_disp(F.interpolate(init_depth.unsqueeze(1), scale_factor=2 ** s, mode="nearest"), dmin, dmax).detach()
        ref, src = _nhwc(fs, B)
        nsamp = a.CostNum[s]

        def cost_fn(inv, confidence, ref=ref, src=src, rt=rt, vw=vw, nsamp=nsamp, s=s):
            return A.getcost(o, ref, src, rt, inv.contiguous(), None if confidence is None else confidence.contiguous(), vw,
                             kmin, kmax, nsamp, interval * _RATIOS[s], a.min_radius, a.max_radius, s, G=a.cost_dim_stage[1])

        ub = f"update_block_depth{s + 1}"
        t = t_source(B, a.timesteps[s], dev)
        noise = noise_source((B, 1, H, W), dev)
        mask, hidden, inv_seq, conf_seq = update_block_train(
            n, ub, cost_fn, inv_cur, hidden, context, inv_gt, inv_init, iters=a.stage_iters[s], dim=a.unet_dim[s],
            L=len(_MULTS[s]), timesteps=a.timesteps[s], scale=a.scale[s], t=t, noise=noise)
        for inv_i in inv_seq:
            depths.append(disp_to_depth(inv_i, dmin, dmax)[1].squeeze(1))
        confs.extend(conf_seq)
        up = upsample_depth(inv_seq[-1], mask, up_ratio).unsqueeze(1)
        depths.append(disp_to_depth(up, dmin, dmax)[1].squeeze(1))
    return {"depth": depths, "conf": confs, "photometric_confidence": confs_full}
