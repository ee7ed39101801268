"""Per-layer timing of dmvs_conv2d_f32 at the shapes of the cfg2 step (B=16, 6 views): us and TFLOP/s against the
157 TF fp32-MFMA peak."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffmvs_amd import ops as K  # noqa: E402

SHAPES = [  # name, N, cin, cout, k, stride, H, W
    ("feat conv1.1 16->16 1/2", 96, 16, 16, 3, 1, 256, 320), ("feat conv2.1 32->32 1/4", 96, 32, 32, 3, 1, 128, 160),
    ("feat conv3.1 64->64 1/8", 96, 64, 64, 3, 1, 64, 80), ("feat conv1.0 8->16 5x5s2", 96, 8, 16, 5, 2, 512, 640),
    ("feat conv2.0 16->32 5x5s2", 96, 16, 32, 5, 2, 256, 320), ("feat conv3.0 32->64 5x5s2", 96, 32, 64, 5, 2, 128, 160),
    ("unet init 64->16 7x7 1/4", 16, 64, 16, 7, 1, 128, 160), ("enc 24->32 3x3 1/4", 16, 24, 32, 3, 1, 128, 160),
    ("unet 32->32 3x3 1/8", 16, 32, 32, 3, 1, 64, 80), ("ctx 16->16 1/2", 16, 16, 16, 3, 1, 256, 320),
    ("cond 32->32 3x3 1/4 B16", 16, 32, 32, 3, 1, 128, 160), ("unet 16->16 3x3 1/4 B16", 16, 16, 16, 3, 1, 128, 160),
    ("unet 64->32 3x3 1/8 B16", 16, 64, 32, 3, 1, 64, 80),
    ("unet 32->32 3x3 1/4 B96", 96, 32, 32, 3, 1, 128, 160), ("unet 16->16 3x3 1/4 B96", 96, 16, 16, 3, 1, 128, 160),
    ("feat out 32->64 1x1 1/4", 576, 32, 64, 1, 1, 128, 160), ("1x1 64->144 1/4 B96", 96, 64, 144, 1, 1, 128, 160),
    ("1x1 32->16 1/4 B96", 96, 32, 16, 1, 1, 128, 160), ("gru 64->64 1x5 1/8 B96", 96, 64, 64, (1, 5), 1, 64, 80),
    ("feat 64->64 3x3 1/8 N576", 576, 64, 64, 3, 1, 64, 80), ("unet 64->31 3x3 1/4 B96", 96, 64, 31, 3, 1, 128, 160),
    ("unet 32->32 3x3 1/8 B96", 96, 32, 32, 3, 1, 64, 80), ("unet 32->16 3x3 1/4 B96", 96, 32, 16, 3, 1, 128, 160),
    ("unet 48->32 3x3 1/8 B96", 96, 48, 32, 3, 1, 64, 80),
]


SHAPES3D = [  # name, N, cin, cout, D, H, W, stride
    ("pvw conv0 4->8 (5 views)", 80, 4, 8, 48, 64, 80, 1), ("costreg conv0 4->8", 16, 4, 8, 48, 64, 80, 1),
    ("costreg conv1 8->8", 16, 8, 8, 48, 64, 80, 1), ("costreg conv3 16->16", 16, 16, 16, 24, 32, 40, 1),
    ("costreg conv5 32->32", 16, 32, 32, 12, 16, 20, 1), ("costreg prob 8->1 B96", 96, 8, 1, 48, 64, 80, 1),
    ("costreg conv0 4->8 B96", 96, 4, 8, 48, 64, 80, 1), ("costreg conv1 8->8 B96", 96, 8, 8, 48, 64, 80, 1),
]


def main():
    o = K.Ops.for_device("cuda:0")
    if os.environ.get("CONV_LIB"):
        from diffmvs_amd import _lib
        o = K.Ops(_lib.Lib(os.path.abspath(os.environ["CONV_LIB"])), "cuda:0")
    g3 = torch.Generator(device="cuda").manual_seed(1)      # (inputs drawn on the device: the N = 576 tensors take seconds on the host)
    for name, N, cin, cout, D, H, W, s in SHAPES3D:
        if os.environ.get("CONV_2D_ONLY") or (os.environ.get("CONV_ONLY") and os.environ["CONV_ONLY"] not in name):
            continue
        x = torch.randn(N, cin, D, H, W, generator=g3, device="cuda")
        w = torch.randn(cout, cin, 3, 3, 3, generator=g3, device="cuda") * 0.1
        pc = K.pack_conv3d(w, None, stride=s)
        for _ in range(3):
            y = o.conv3d(pc, x, act=K.ACT_RELU)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(10):
            o.conv3d(pc, x, act=K.ACT_RELU)
        en.record()
        torch.cuda.synchronize()
        us = st.elapsed_time(en) * 100.0
        flops = 2.0 * y.numel() * cin * 27
        gb = 4.0 * (x.numel() + y.numel()) / 1e9
        print(json.dumps({"layer": name, "us": round(us, 1), "TFLOPs": round(flops / us / 1e6, 1), "GB": round(gb, 3),
                          "TBs": round(gb / us * 1e3, 2)}))
    if os.environ.get("CONV_3D_ONLY"):
        return
    g = torch.Generator(device="cuda").manual_seed(0)
    only = os.environ.get("CONV_ONLY")
    arith = K.CONV_ARITH[os.environ.get("CONV_ARITH", "fp32")]      # CONV_ARITH=fp32|bf16|split
    for name, N, cin, cout, k, s, H, W in SHAPES:
        if only and only not in name:
            continue
        x = torch.randn(N, cin, H, W, generator=g, device="cuda")
        kh, kw = k if isinstance(k, tuple) else (k, k)
        w = torch.randn(cout, cin, kh, kw, generator=g, device="cuda") * 0.1
        pc = K.pack_conv2d(w, None, stride=s, pad=(kh // 2, kw // 2))
        for _ in range(3):
            y = o.conv2d(pc, x, act=K.ACT_RELU, arith=arith)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(10):
            o.conv2d(pc, x, act=K.ACT_RELU, arith=arith)
        en.record()
        torch.cuda.synchronize()
        us = st.elapsed_time(en) * 100.0
        flops = 2.0 * N * y.shape[2] * y.shape[3] * cout * cin * kh * kw
        gb = 4.0 * (x.numel() + y.numel()) / 1e9
        print(json.dumps({"layer": name, "us": round(us, 1), "TFLOPs": round(flops / us / 1e6, 1), "frac_mfma": round(flops / us / 1e6 / 157.3, 3),
                          "GB": round(gb, 3), "TBs": round(gb / us * 1e3, 2)}))


if __name__ == "__main__":
    main()
