"""Parameter containers + helpers mirroring the reference's models/module.py API.

The classes below exist so that (a) `load_state_dict` accepts the reference's checkpoints
key-for-key (SURVEY section 8b "Checkpoint layout") and (b) user code that pokes at
`model.feature`, `model.depthnet`, ... keeps working.  They hold parameters only; the
arithmetic is done by diffmvs_amd.engine.Engine, which reads the flat state dict and drives
the HIP kernels.  Sub-module forward()s that the reference exposes are routed through the
same engine pieces.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from diffmvs_amd import engine as E
from diffmvs_amd import ops as K
from diffmvs_amd.ops import Ops


class HipModule(nn.Module):
    """nn.Module whose forward() runs on libdmvs_hip.so.  Weights are packed into kernel layout on first
    use and re-packed when any parameter / buffer changes; the library binding is that of the parameters' HIP
    device (Ops.for_device: no CPU path)."""

    def ops(self, like=None) -> Ops:
        t = next(self.parameters(), None)
        if t is None:
            t = like
        return Ops.for_device(t.device)

    def packed(self, builder):
        key = (K.weights_generation(),) + tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))
        cache = self.__dict__.setdefault("_pack_cache", {})
        if cache.get("key") != key:
            dev = self.ops().device
            sd = {k: v.detach().to(dev) for k, v in self.state_dict().items()}
            cache.update(key=key, value=builder(sd))
        return cache["value"]

    def _eval_only(self):
        if self.training:
            raise NotImplementedError(f"{type(self).__name__}.forward() stand-alone is the eval path (folded BatchNorm, packed "
                                      "weights); training runs through CasDiffMVS.forward in train mode "
                                      "(diffmvs_amd/train.py builds the autograd graph over the whole model); call .eval()")


def _dev(o: Ops, t):
    return None if t is None else t.to(o.device).float().contiguous()


def _bn(kind, ch, momentum=0.1):
    return (nn.BatchNorm2d if kind == 2 else nn.BatchNorm3d)(ch, momentum=momentum)


class _ConvUnit(HipModule):
    """conv (+bn) holder with the reference's attribute names `conv` / `bn` / `relu`;
    forward = conv -> eval BN -> optional ReLU in one fused kernel."""

    def __init__(self, conv: nn.Module, bn: nn.Module | None, relu: bool):
        super().__init__()
        self.conv = conv
        self.bn = bn
        self.relu = relu

    def _build(self, sd):
        c = self.conv
        bn = {k: sd[f"bn.{k}"] for k in ("weight", "bias", "running_mean", "running_var")} if self.bn is not None else None
        if isinstance(c, nn.Conv2d):
            return K.pack_conv2d(sd["conv.weight"], sd.get("conv.bias"), bn=bn, stride=c.stride[0], pad=tuple(c.padding))
        return K.pack_conv3d(sd["conv.weight"], sd.get("conv.bias"), bn=bn, stride=c.stride[0],
                             transposed=isinstance(c, nn.ConvTranspose3d))

    def forward(self, x):
        self._eval_only()
        o = self.ops()
        pc = self.packed(self._build)
        act = K.ACT_RELU if self.relu else K.ACT_NONE
        if isinstance(self.conv, nn.Conv2d):
            return o.conv2d(pc, _dev(o, x), act=act)
        return o.conv3d(pc, _dev(o, x), act=act)


class Conv2d(_ConvUnit):            # reference models/module.py:24-64
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, relu=True, bn=True, bn_momentum=0.1,
                 init_method="xavier", **kwargs):
        super().__init__(nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs),
                         _bn(2, out_channels, bn_momentum) if bn else None, relu)
        self.kernel_size, self.stride = kernel_size, stride


class Conv3d(_ConvUnit):            # reference models/module.py:66-108
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, relu=True, bn=True, bn_momentum=0.1,
                 init_method="xavier", **kwargs):
        assert stride in (1, 2)
        super().__init__(nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs),
                         _bn(3, out_channels, bn_momentum) if bn else None, relu)
        self.out_channels, self.kernel_size, self.stride = out_channels, kernel_size, stride


class Deconv3d(_ConvUnit):          # reference models/module.py:110-150
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, relu=True, bn=True, bn_momentum=0.1,
                 init_method="xavier", **kwargs):
        assert stride in (1, 2)
        super().__init__(nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs),
                         _bn(3, out_channels, bn_momentum) if bn else None, relu)
        self.out_channels, self.stride = out_channels, stride


class ConvBnReLU(_ConvUnit):        # reference models/module.py:279-289
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__(nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False),
                         nn.BatchNorm2d(out_channels), True)


class ConvBn(_ConvUnit):            # reference models/module.py:291-301
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__(nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False),
                         nn.BatchNorm2d(out_channels), False)


class ResidualBlock(HipModule):     # reference models/module.py:303-319
    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = ConvBnReLU(in_planes, planes, 3, stride=stride, pad=1)
        self.conv2 = ConvBn(planes, planes, 3, stride=1, pad=1)
        self.downsample = None if stride == 1 else ConvBn(in_planes, planes, 3, stride=stride, pad=1)

    def forward(self, x):
        self._eval_only()
        o = self.ops()
        x = _dev(o, x)
        y = self.conv1(x)
        res = x if self.downsample is None else self.downsample(x)
        return o.conv2d(self.conv2.packed(self.conv2._build), y, residual=res, act=K.ACT_RELU)   # relu(x + y)


class ContextNet(HipModule):        # reference models/module.py:321-355
    def __init__(self, out_dim=[16, 16, 16]):
        super().__init__()
        self.out_dim = out_dim
        self.conv1 = ConvBnReLU(3, 8)
        planes = 8
        for i, dim in enumerate((16, 32, 48), start=1):
            setattr(self, f"layer{i}", nn.Sequential(ResidualBlock(planes, dim, stride=2), ResidualBlock(dim, dim)))
            planes = dim
        self.output1 = nn.Conv2d(48, out_dim[0], 3, stride=1, padding=1)
        self.output2 = nn.Conv2d(32, out_dim[1], 3, stride=1, padding=1)
        if out_dim[2] > 0:
            self.output3 = nn.Conv2d(16, out_dim[2], 3, stride=1, padding=1)

    def forward(self, x):
        self._eval_only()
        o = self.ops()

        def build(sd):
            heads = {i: K.pack_conv2d(sd[f"output{i + 1}.weight"], sd[f"output{i + 1}.bias"], pad=1)
                     for i in range(3) if f"output{i + 1}.weight" in sd}
            return E.pack_context_trunk({"context." + k: v for k, v in sd.items()}, "context"), heads
        trunk_pk, heads = self.packed(build)
        taps = E.run_context_trunk(o, trunk_pk, _dev(o, x))
        return {f"stage{i + 1}": o.conv2d(pc, taps[i]) for i, pc in sorted(heads.items(), reverse=True)}


class FeatureNet(HipModule):        # reference models/module.py:357-420
    def __init__(self, base_channels=8, out_channel=[32, 16, 8]):
        super().__init__()
        c = base_channels
        self.base_channels, self.out_channel = c, out_channel
        self.conv0 = nn.Sequential(Conv2d(3, c, 3, 1, padding=1), Conv2d(c, c, 3, 1, padding=1))
        for i in (1, 2, 3):
            cin, cout = c * 2 ** (i - 1), c * 2 ** i
            setattr(self, f"conv{i}", nn.Sequential(Conv2d(cin, cout, 5, stride=2, padding=2),
                                                    Conv2d(cout, cout, 3, 1, padding=1),
                                                    Conv2d(cout, cout, 3, 1, padding=1)))
        self.out1 = nn.Conv2d(c * 8, out_channel[0], 1, bias=False)
        self.inner1 = nn.Conv2d(c * 4, c * 8, 1, bias=True)
        self.out2 = nn.Conv2d(c * 8, out_channel[1], 3, padding=1, bias=False)
        if out_channel[2] > 0:
            self.inner2 = nn.Conv2d(c * 2, c * 8, 1, bias=True)
            self.out3 = nn.Conv2d(c * 8, out_channel[2], 3, padding=1, bias=False)

    def forward(self, x):
        """-> {"stage1": [B,48,H/8,W/8], "stage2": [B,32,H/4,W/4], ("stage3": [B,16,H/2,W/2])}, NCHW as the reference"""
        self._eval_only()
        o = self.ops()
        pk = self.packed(lambda sd: E.pack_feature({"feature." + k: v for k, v in sd.items()}, "feature"))
        return E.run_feature(o, pk, _dev(o, x), layout=K.LAYOUT_NCHW)


class CostRegNet_small(HipModule):  # reference models/module.py:422-448
    def __init__(self, in_channels, base_channels):
        super().__init__()
        b = base_channels
        self.conv0, self.conv1 = Conv3d(in_channels, b, padding=1), Conv3d(b, b, padding=1)
        self.conv2, self.conv3 = Conv3d(b, b * 2, stride=2, padding=1), Conv3d(b * 2, b * 2, padding=1)
        self.conv4, self.conv5 = Conv3d(b * 2, b * 4, stride=2, padding=1), Conv3d(b * 4, b * 4, padding=1)
        self.conv6 = Deconv3d(b * 4, b * 2, stride=2, padding=1, output_padding=1)
        self.conv7 = Deconv3d(b * 2, b, stride=2, padding=1, output_padding=1)
        self.prob = nn.Conv3d(b, 1, 3, stride=1, padding=1, bias=False)

    def forward(self, x):
        self._eval_only()
        o = self.ops()
        return E.run_costreg(o, self.packed(lambda sd: E.pack_costreg({"r." + k: v for k, v in sd.items()}, "r")), _dev(o, x))


class PixelViewWeight(HipModule):   # reference models/module.py:450-463
    def __init__(self, G):
        super().__init__()
        self.conv = nn.Sequential(Conv3d(G, 8, padding=1), nn.Conv3d(8, 1, 3, stride=1, padding=1))

    def forward(self, x):
        """x [B,G,D,H,W] -> [B,1,H,W]"""
        self._eval_only()
        o = self.ops()
        pk = self.packed(lambda sd: E.pack_pvw({"p." + k: v for k, v in sd.items()}, "p"))
        return E.run_pvw(o, pk, _dev(o, x)).unsqueeze(1)


def _mask_head(cin, ratio):
    return nn.Sequential(nn.Conv2d(cin, 64, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(64, ratio * ratio * 9, 1, padding=0))


def _nhwc_stack(o: Ops, features):
    """list of V [B,C,H,W] (reference layout) -> ref [B,H,W,C], src [S,B,H,W,C] for the warp kernels"""
    nhwc = [o.nchw_to_nhwc(_dev(o, f)) for f in features]
    return nhwc[0], torch.stack(nhwc[1:], 0).contiguous()


class InitialCost(HipModule):       # reference models/module.py:465-573
    def __init__(self, feature_dim, group_dim=8, ratio=2):
        super().__init__()
        self.group_dim = group_dim
        self.pixel_view_weight = PixelViewWeight(group_dim)
        self.cost_regularization = CostRegNet_small(in_channels=group_dim, base_channels=8)
        self.mask = _mask_head(feature_dim, ratio)

    def forward(self, features, context, proj_matrices, depth_values, scale_inv_depth=None):
        """Same arguments / returns as the reference (module.py:487-573).  `depth_values` [B,D,H,W] must be
        the uniform inverse-depth plane sweep the reference builds (diffusion.py:187-192): the kernel
        regenerates it from its first / last plane.  -> mask, normalized_depth [B,1,H,W], depth [B,H,W],
        view_weights [B,S,H,W], photometric_confidence [B,1,H,W]."""
        self._eval_only()
        o = self.ops()

        def build(sd):
            return (E.pack_pvw(sd, "pixel_view_weight"), E.pack_costreg(sd, "cost_regularization"), E.pack_mask(sd, "mask"))
        pvw, reg, mask_pk = self.packed(build)
        ref, src = _nhwc_stack(o, features)
        B, H, W, _ = ref.shape
        S, D, G = src.shape[0], depth_values.shape[1], self.group_dim
        dv = _dev(o, depth_values)
        disp_min = (1.0 / dv[:, 0, 0, 0]).contiguous()     # plane 0 is the farthest (normalised inverse depth 0)
        disp_max = (1.0 / dv[:, -1, 0, 0]).contiguous()
        rt = o.compose_proj(_dev(o, proj_matrices))
        cor = o.warp_corr_init_quad(ref, src, rt, disp_min, disp_max, D, G, plain=True)
        vw = E.run_pvw(o, pvw, cor.view(B * S, G, D, H, W)).view(B, S, H, W)
        logits = E.run_costreg(o, reg, o.view_aggregate(cor, vw))
        nd, depth, conf = o.depth_regress(logits.view(B, D, H, W), disp_min, disp_max)
        return E.run_mask(o, mask_pk, _dev(o, context)), nd, depth, vw, conf


class GetCost(HipModule):           # reference models/module.py:575-667 (no parameters)
    def __init__(self, group_dim=4, min_radius=0.2, max_radius=2):
        super().__init__()
        self.group_dim, self.min_radius, self.max_radius = group_dim, min_radius, max_radius

    def forward(self, inverse_depth, features, proj_matrices, depth_interval, depth_max, depth_min, CostNum=4,
                view_weights=None, confidence=None):
        """Same arguments / returns as the reference (module.py:583-667): -> cost [B,G*CostNum,H,W],
        inverse_depth_samples [B,CostNum,H,W].  features: list of V [B,C,H,W]; view_weights [B,S,H,W]."""
        o = self.ops(like=inverse_depth)
        ref, src = _nhwc_stack(o, features)
        B = ref.shape[0]
        rt = o.compose_proj(_dev(o, proj_matrices))
        dmax = torch.as_tensor(depth_max, dtype=torch.float32, device=o.device).reshape(-1).expand(B)
        dmin = torch.as_tensor(depth_min, dtype=torch.float32, device=o.device).reshape(-1).expand(B)
        return o.getcost_quad(ref, src, rt, _dev(o, inverse_depth), _dev(o, confidence), _dev(o, view_weights),
                              (1.0 / dmax).contiguous(), (1.0 / dmin).contiguous(), CostNum, float(depth_interval),
                              self.min_radius, self.max_radius, vw_shift=0, G=self.group_dim, plain=True)


class SepConvGRU(HipModule):        # reference models/module.py:152-179
    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        for n, k, p in (("1", (1, 5), (0, 2)), ("2", (5, 1), (2, 0))):
            for gate in "zrq":
                setattr(self, f"conv{gate}{n}", nn.Conv2d(hidden_dim + input_dim, hidden_dim, k, padding=p))

    def forward(self, h, x):
        o = self.ops()
        return E.run_gru(o, self.packed(lambda sd: E.pack_gru({"g." + k: v for k, v in sd.items()}, "g")), _dev(o, h), _dev(o, x))


def differentiable_warping(src_fea, src_proj, ref_proj, depth_values):
    """get warped source image features (reference models/module.py:181-218): src_fea [B,C,Hs,Ws], projs [B,4,4],
    depth_values [B,D,H,W] -> [B,C,D,H,W]."""
    o = Ops.for_device(src_fea.device)
    proj = torch.matmul(src_proj.double(), torch.linalg.inv(ref_proj.double()))      # 4x4 camera algebra
    rt = torch.cat([proj[:, :3, :3].reshape(-1, 9), proj[:, :3, 3]], 1).float().contiguous()
    return o.warp_volume(_dev(o, src_fea), rt.to(o.device), _dev(o, depth_values))


def upsample_depth(depth, mask, ratio=8):
    """upsample depth map using convex combination (reference models/module.py:237-248): [N,1,H,W] -> [N,rH,rW]."""
    o = Ops.for_device(depth.device)
    N = depth.shape[0]
    one, zero = torch.ones(N, device=o.device), torch.zeros(N, device=o.device)     # identity depth transform
    up, _ = o.convex_upsample(_dev(o, depth), _dev(o, mask), zero, one, ratio)
    return up


def get_cur_depth_range_samples(cur_depth, ndepth, depth_inteval_pixel, confidence=None, min=0.2, max=2):
    """sample new depth hypotheses in the inverse range (reference models/module.py:250-277); small tensor math,
    fused into dmvs_getcost_f32 on the model's path."""
    radius = ndepth // 2 * depth_inteval_pixel
    if confidence is not None:
        radius = min * radius + (1 - confidence) * (max * radius - min * radius)
    lo, hi = cur_depth - radius, cur_depth + radius
    step = (hi - lo) / (ndepth - 1)
    k = torch.arange(0, ndepth, device=cur_depth.device, dtype=cur_depth.dtype).reshape(1, -1, 1, 1)
    return torch.clamp(k * step.unsqueeze(1) + lo.unsqueeze(1), min=0, max=1)


# ---- scalar helpers with the reference's names (models/module.py:220-235); plain tensor math
def disp_to_depth(disp, min_depth, max_depth):
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    scaled_disp = (min_disp + (max_disp - min_disp) * disp).clamp(min=1e-6)
    return scaled_disp, 1 / scaled_disp


def depth_to_disp(depth, min_depth, max_depth):
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    return (1 / depth - min_disp) / (max_disp - min_disp)
