#!/usr/bin/env python3
"""DEV-ONLY fixture generator (round 2 additions); same rules as make_golden.py: runs only where
/root/reference is mounted, writes tensors only.

 * e2e_<variant>_b2_ms2.npz      test=True, sampling_timesteps = [0, 2, 2]: the multi-step DDIM tail
                                 (reference models/update.py:504-519), two noise draws per refinement stage
 * e2e_<variant>_b2_evalall.npz  test=False in eval mode (reference models/diffusion.py:264-270): every
                                 iterate in "depth", the Unet confidences in "conf" -- what the reference's
                                 validation loop (train.py test_sample_depth) consumes
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from diffmvs_amd import synth  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models, _, _ = MG.import_reference()
    meta = {"torch": torch.__version__, "numpy": np.__version__}
    for variant in ("diffmvs", "casdiffmvs"):
        for tag, test, st in (("ms2", True, [0, 2, 2]), ("evalall", False, [0, 1, 1])):
            args = synth.make_args(variant, numdepth_initial=16, sampling_timesteps=st)
            model = ref_models.CasDiffMVS(args, test=test)
            model.load_state_dict(synth.synth_state_dict(model.state_dict(), seed=123), strict=True)
            model.eval()
            imgs, proj, dv = synth.synth_inputs(64, 96, 3, B=2, seed=2)
            out, drawn = MG.run_forward(model, imgs, proj, dv, noise_seed=9)
            e2e = {}
            MG.flat("out", out, e2e)
            MG.flat("noise", drawn, e2e)
            e2e["meta"] = np.array(json.dumps(dict(meta, H=64, W=96, B=2, n_src=3, nd_init=16, scene_seed=2, noise_seed=9,
                                                   weight_seed=123, test=test, sampling_timesteps=st)))
            np.savez_compressed(os.path.join(HERE, f"e2e_{variant}_b2_{tag}.npz"), **e2e)
            print(variant, tag, "depth maps", len(out["depth"]), "conf", len(out["conf"]), "pc", len(out["photometric_confidence"]),
                  "noise draws", len(drawn))


if __name__ == "__main__":
    main()
