"""A/B of the opt-in experiments that were built but not timed (DESIGN.md sections 4.1, 4.2, 9, 11), in ONE gpurun call:

    gpurun --timeout 600 -- 'python tools/ab_experiments.py > gpurun_out/ab_experiments.jsonl'

1. parity of every experiment against the default path on the GPU (bit for bit where the experiment claims it);
2. `bench.py --no-batch-sweep --no-cpu-baseline --steps 10 --warmup 2` once per arm (~70 s each), one JSON line per arm with the
   headline value, ms per step and the conv2d roofline fraction.
Arms are environment knobs read by the library at launch time, so nothing has to be rebuilt between them."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARMS = [
    ("default", {}),
    ("conv3d_16_byte_halo", {"DMVS_CONV3D_V16": "1"}),
    ("stem_16_byte_halo", {"DMVS_STEM_V16": "1"}),
    ("conv3d_and_stem_16_byte", {"DMVS_CONV3D_V16": "1", "DMVS_STEM_V16": "1"}),
    ("conv2d_4_byte_pieces", {"DMVS_CONV_V16": "0"}),
    ("feature_pair_kernel", {"DMVS_FEAT_PAIR": "1"}),
    ("bf16_matrix_arithmetic", {"DMVS_CONV_ARITH": "bf16"}),
]


def parity():
    """the stem experiment has only run on the host emulation so far: check it on the device before timing it"""
    sys.path.insert(0, ROOT)
    import torch
    from diffmvs_amd import ops as K
    o = K.Ops.for_device("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(4, 3, 512, 640, generator=g, device="cuda")
    w0, w1 = torch.randn(8, 3, 3, 3, generator=g, device="cuda") * 0.4, torch.randn(8, 8, 3, 3, generator=g, device="cuda") * 0.3
    b0, b1 = torch.randn(8, generator=g, device="cuda"), torch.randn(8, generator=g, device="cuda")
    pc0, pc1 = K.pack_conv2d(w0, b0, pad=1), K.pack_conv2d(w1, b1, pad=1)
    a = o.featurenet_stem(pc0, pc1, x)
    os.environ["DMVS_STEM_V16"] = "1"
    b = o.featurenet_stem(pc0, pc1, x)
    del os.environ["DMVS_STEM_V16"]
    ok_stem = bool(torch.equal(a, b))
    v = torch.randn(6, 4, 48, 64, 80, generator=g, device="cuda")
    w3 = torch.randn(8, 4, 3, 3, 3, generator=g, device="cuda") * 0.2
    pc3 = K.pack_conv3d(w3, None)
    a3 = o.conv3d(pc3, v, act=K.ACT_RELU)
    os.environ["DMVS_CONV3D_V16"] = "1"
    b3 = o.conv3d(pc3, v, act=K.ACT_RELU)
    del os.environ["DMVS_CONV3D_V16"]
    ok_3d = bool(torch.equal(a3, b3))
    print(json.dumps({"parity": {"stem_16_byte_bit_identical": ok_stem, "conv3d_16_byte_bit_identical": ok_3d}}), flush=True)
    return ok_stem, ok_3d


def main():
    ok_stem, ok_3d = parity()
    for name, env in ARMS:
        if ("DMVS_STEM_V16" in env and not ok_stem) or ("DMVS_CONV3D_V16" in env and not ok_3d):
            print(json.dumps({"arm": name, "skipped": "parity failed"}), flush=True)
            continue
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-batch-sweep", "--no-cpu-baseline", "--steps", "10", "--warmup", "2"],
                             env=dict(os.environ, **env), capture_output=True, text=True, cwd=ROOT)
        try:
            line = json.loads(out.stdout.strip().splitlines()[-1])
            print(json.dumps({"arm": name, "env": env, "value": line["value"], "ms_per_step": line["ms_per_step"],
                              "roofline_conv2d_frac": line["roofline_conv2d"]["frac"], "roofline_frac": line["roofline"]["frac"],
                              "roofline_warp_init_frac": line["roofline_warp_init"]["frac"]}), flush=True)
        except Exception as e:      # noqa: BLE001
            print(json.dumps({"arm": name, "error": repr(e), "stderr_tail": out.stderr[-400:]}), flush=True)


if __name__ == "__main__":
    main()
