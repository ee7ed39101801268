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


def _bn(kind, ch, momentum=0.1):
    return (nn.BatchNorm2d if kind == 2 else nn.BatchNorm3d)(ch, momentum=momentum)


class _ConvUnit(nn.Module):
    """conv (+bn) holder with the reference's attribute names `conv` / `bn` / `relu`."""

    def __init__(self, conv: nn.Module, bn: nn.Module | None, relu: bool):
        super().__init__()
        self.conv = conv
        self.bn = bn
        self.relu = relu


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


class ResidualBlock(nn.Module):     # reference models/module.py:303-319
    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = ConvBnReLU(in_planes, planes, 3, stride=stride, pad=1)
        self.conv2 = ConvBn(planes, planes, 3, stride=1, pad=1)
        self.downsample = None if stride == 1 else ConvBn(in_planes, planes, 3, stride=stride, pad=1)


class ContextNet(nn.Module):        # reference models/module.py:321-355
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


class FeatureNet(nn.Module):        # reference models/module.py:357-420
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


class CostRegNet_small(nn.Module):  # reference models/module.py:422-448
    def __init__(self, in_channels, base_channels):
        super().__init__()
        b = base_channels
        self.conv0, self.conv1 = Conv3d(in_channels, b, padding=1), Conv3d(b, b, padding=1)
        self.conv2, self.conv3 = Conv3d(b, b * 2, stride=2, padding=1), Conv3d(b * 2, b * 2, padding=1)
        self.conv4, self.conv5 = Conv3d(b * 2, b * 4, stride=2, padding=1), Conv3d(b * 4, b * 4, padding=1)
        self.conv6 = Deconv3d(b * 4, b * 2, stride=2, padding=1, output_padding=1)
        self.conv7 = Deconv3d(b * 2, b, stride=2, padding=1, output_padding=1)
        self.prob = nn.Conv3d(b, 1, 3, stride=1, padding=1, bias=False)


class PixelViewWeight(nn.Module):   # reference models/module.py:450-463
    def __init__(self, G):
        super().__init__()
        self.conv = nn.Sequential(Conv3d(G, 8, padding=1), nn.Conv3d(8, 1, 3, stride=1, padding=1))


def _mask_head(cin, ratio):
    return nn.Sequential(nn.Conv2d(cin, 64, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(64, ratio * ratio * 9, 1, padding=0))


class InitialCost(nn.Module):       # reference models/module.py:465-573
    def __init__(self, feature_dim, group_dim=8, ratio=2):
        super().__init__()
        self.group_dim = group_dim
        self.pixel_view_weight = PixelViewWeight(group_dim)
        self.cost_regularization = CostRegNet_small(in_channels=group_dim, base_channels=8)
        self.mask = _mask_head(feature_dim, ratio)


class GetCost(nn.Module):           # reference models/module.py:575-667 (no parameters)
    def __init__(self, group_dim=4, min_radius=0.2, max_radius=2):
        super().__init__()
        self.group_dim, self.min_radius, self.max_radius = group_dim, min_radius, max_radius


class SepConvGRU(nn.Module):        # reference models/module.py:152-179
    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        for n, k, p in (("1", (1, 5), (0, 2)), ("2", (5, 1), (2, 0))):
            for gate in "zrq":
                setattr(self, f"conv{gate}{n}", nn.Conv2d(hidden_dim + input_dim, hidden_dim, k, padding=p))


# ---- scalar helpers with the reference's names (models/module.py:220-235); plain tensor math
def disp_to_depth(disp, min_depth, max_depth):
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    scaled_disp = (min_disp + (max_disp - min_disp) * disp).clamp(min=1e-6)
    return scaled_disp, 1 / scaled_disp


def depth_to_disp(depth, min_depth, max_depth):
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    return (1 / depth - min_disp) / (max_disp - min_disp)
