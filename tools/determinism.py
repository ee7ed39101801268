"""Run-to-run reproducibility of the eval forward, kernel by kernel (JSON lines on stdout).

    python tools/determinism.py [cfg5|cfg3|cfg2] [--runs N]

DESIGN section 5's contract: same inputs + same noise => bit-identical outputs (no floating-point atomics in the eval forward).  The
round-4 driver run broke it at BASELINE.json configs[4] size.  This tool finds WHERE:
  1. `plain`     N forwards, nothing hooked: do the returned maps differ at all on this box, and in which rows
  2. `toggles`   the same with each A/B tune bit of the library forced, one at a time (a bit that makes the runs agree names the kernel)
  3. `fill`      every freshly allocated kernel output pre-filled with NaN, then with 1e30: an output that changes with the fill value reads
                 memory no kernel wrote (ragged bottom / right edges)
  4. `trace`     every Ops call hashed (exact integer checksum of each output): the first call whose checksum differs between runs
  5. `replay`    that call alone, 10 times from cloned inputs: is the kernel itself nondeterministic, where do the replays differ
DIAG_DEVICE=cpu dry-runs the script on the host emulation with a tiny geometry (no GPU in the build container).
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffmvs_amd import _lib, synth  # noqa: E402
from diffmvs_amd.ops import Ops  # noqa: E402

DEV = os.environ.get("DIAG_DEVICE", "cuda:0")
DRY = DEV == "cpu"
CFG = {"cfg5": ("casdiffmvs", 96, 1056, 1920, 11, 1, 17, 5), "cfg3": ("casdiffmvs", 48, 864, 1152, 7, 1, 9, 2),
       "cfg2": ("diffmvs", 48, 512, 640, 5, 2, 5, 1)}
HOOKED = ["conv2d", "featurenet_stem", "conv3d", "compose_proj", "warp_corr_init_quad", "getcost_quad", "view_aggregate", "sigmoid_max_d",
          "depth_regress", "convex_upsample", "groupnorm_apply", "delta_update", "depth_convert", "upsample_nearest"]


def emit(**kw):
    print(json.dumps(kw), flush=True)


def sync():
    if not DRY:
        torch.cuda.synchronize()


def checksum(t):
    if t is None:
        return None
    t = t.contiguous()
    iv = {8: torch.int64, 4: torch.int32, 2: torch.int16, 1: torch.int8}[t.element_size()]
    return int(torch.sum(t.view(iv), dtype=torch.int64).item())


def tensors_of(v):
    if torch.is_tensor(v):
        return [v]
    if isinstance(v, (list, tuple)):
        return [t for x in v for t in tensors_of(x)]
    return []


def clone_args(v):
    if torch.is_tensor(v):
        if v.is_contiguous():
            return v.clone()
        # a channel slice of a larger tensor (the merged GRU gate output): clone the storage range and re-slice it
        base = torch.empty(v.untyped_storage().nbytes() // v.element_size(), dtype=v.dtype, device=v.device)
        base.copy_(torch.as_strided(v, (base.numel(),), (1,), 0))
        return torch.as_strided(base, v.shape, v.stride(), v.storage_offset())
    if isinstance(v, list):
        return [clone_args(x) for x in v]
    if isinstance(v, tuple):
        return tuple(clone_args(x) for x in v)
    if isinstance(v, dict):
        return {k: clone_args(x) for k, x in v.items()}
    return v


class Trace:
    """wraps the Ops methods of one binding: per call, the exact checksum of every output tensor"""

    def __init__(self, ops):
        self.ops, self.orig, self.log, self.capture_at, self.captured = ops, {}, [], None, None
        for name in HOOKED:
            self.orig[name] = getattr(ops, name)
            setattr(ops, name, self._wrap(name))

    def _wrap(self, name):
        fn = self.orig[name]

        def call(*a, **kw):
            idx = len(self.log)
            if idx == self.capture_at:
                self.captured = (name, clone_args(a), clone_args(kw))
            out = fn(*a, **kw)
            self.log.append((name, [tuple(t.shape) for t in tensors_of(out)], [checksum(t) for t in tensors_of(out)]))
            return out
        return call

    def restore(self):
        for name, fn in self.orig.items():
            if name in self.ops.__dict__:
                del self.ops.__dict__[name]


def where_differs(a, b):
    a, b = a.contiguous(), b.contiguous()
    ne = (a != b) & ~(torch.isnan(a) & torch.isnan(b))
    n = int(ne.sum())
    if n == 0:
        return {"n": 0}
    idx = ne.nonzero()
    lo, hi = idx.min(0).values.tolist(), idx.max(0).values.tolist()
    d = (a.double() - b.double()).abs()
    return {"n": n, "of": a.numel(), "lo": lo, "hi": hi, "max_abs": float(d[ne].max()), "max_val": float(a.abs().max())}


def main():
    cfg = next((a for a in sys.argv[1:] if a in CFG), "cfg5")
    runs = int(sys.argv[sys.argv.index("--runs") + 1]) if "--runs" in sys.argv else 4
    variant, nd, H, W, S, B, seed, nseed = CFG[cfg]
    if DRY:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from hipemu.build import build_emu
        ops = Ops(_lib.Lib(build_emu()), "cpu")
        Ops.for_device = classmethod(lambda cls, device: ops)
        nd, H, W, S, runs = 8, 64, 96, 2, 2
    from models import CasDiffMVS
    args = synth.make_args(variant, numdepth_initial=nd)
    model = CasDiffMVS(args, test=True).eval()
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), 123), strict=True)
    model = model.to(DEV)
    imgs, proj, dv = synth.synth_inputs(H, W, S, B=B, seed=seed)
    imgs, proj, dv = [i.to(DEV) for i in imgs], {k: v.to(DEV) for k, v in proj.items()}, dv.to(DEV)

    def forward():
        model.noise_source = synth.NoiseSource(nseed)
        with torch.no_grad():
            out = model(imgs, proj, dv)
        sync()
        return [d.clone() for d in out["depth"]] + [c.clone() for c in out["photometric_confidence"]]

    ops = ops if DRY else Ops.for_device(DEV)
    base_tune = dict(ops.tune)

    def agree(n):
        first, bad = forward(), []
        for r in range(1, n):
            again = forward()
            for i, (x, y) in enumerate(zip(again, first)):
                w = where_differs(x, y)
                if w["n"]:
                    bad.append({"run": r, "output": i, "shape": list(x.shape), **w})
        return first, bad

    force = int(sys.argv[sys.argv.index("--replay-idx") + 1]) if "--replay-idx" in sys.argv else None      # (self-test of step 5)
    # 1. plain
    ref, bad = agree(runs)
    emit(step="plain", cfg=cfg, runs=runs, differing=bad[:12], n_differing=len(bad))

    T = _lib
    toggles = [("stem_pieces4", "stem", T.TUNE_PIECES4), ("conv_no_tall", "conv2d", T.TUNE_NO_TALL), ("conv_no_lean", "conv2d", T.TUNE_NO_LEAN),
               ("conv_no_walk", "conv2d", T.TUNE_NO_WALK), ("conv_pieces4", "conv2d", T.TUNE_PIECES4), ("conv_1x1_tiled", "conv2d", T.TUNE_1X1_TILED),
               ("conv3d_pieces4", "conv3d", T.TUNE3D_PIECES4), ("sweep_global", "sweep", T.TUNE_SWEEP_GLOBAL),
               ("all_conservative", None, 0)]
    # 2. toggles
    for name, key, bit in ([] if force is not None else toggles):
        ops.tune.update(base_tune)
        if key is None:
            ops.tune.update({"stem": T.TUNE_PIECES4, "conv2d": T.TUNE_NO_TALL | T.TUNE_NO_LEAN | T.TUNE_NO_WALK | T.TUNE_PIECES4 | T.TUNE_1X1_TILED,
                             "conv3d": T.TUNE3D_PIECES4, "sweep": T.TUNE_SWEEP_GLOBAL})
        else:
            ops.tune[key] = base_tune[key] | bit
        _, bad = agree(runs)
        emit(step="toggle", name=name, n_differing=len(bad), first=bad[:2])
    ops.tune.update(base_tune)

    # 3. fill
    for fill in (() if force is not None else (float("nan"), 1e30)):
        ops.debug_fill = fill
        got = forward()
        ops.debug_fill = None
        rows = []
        for i, (x, y) in enumerate(zip(got, ref)):
            w = where_differs(x, y)
            if w["n"] or not bool(torch.isfinite(x).all()):
                rows.append({"output": i, "nonfinite": int((~torch.isfinite(x)).sum()), **w})
        emit(step="fill", fill=str(fill), affected=rows)

    # 4. trace
    traces = []
    for r in range(runs):
        tr = Trace(ops)
        forward()
        tr.restore()
        traces.append(tr.log)
    first_bad, bad_calls = None, []
    for i, row in enumerate(traces[0]):
        for r in range(1, runs):
            if i >= len(traces[r]) or traces[r][i] != row:
                bad_calls.append(i)
                break
    emit(step="trace", calls=len(traces[0]), n_bad=len(bad_calls),
         first_bad=[{"idx": i, "op": traces[0][i][0], "shapes": traces[0][i][1]} for i in bad_calls[:12]])
    if not bad_calls and force is None:
        return
    first_bad = bad_calls[0] if force is None else force

    # 5. replay the first differing call from cloned inputs
    tr = Trace(ops)
    tr.capture_at = first_bad
    forward()
    tr.restore()
    name, a, kw = tr.captured
    fn = getattr(ops, name)

    def one():
        out = fn(*clone_args(a), **clone_args(kw))
        sync()
        return [t.clone() for t in tensors_of(out)]
    desc = {"op": name, "arg_shapes": [list(t.shape) for t in tensors_of(a)],
            "kw": {k: (list(v.shape) if torch.is_tensor(v) else v) for k, v in kw.items() if not isinstance(v, (list, dict))}}
    pc = a[0] if a and hasattr(a[0], "cin") else None
    if pc is not None:
        desc["conv"] = {"cin": pc.cin, "cout": pc.cout, "k": list(pc.k), "stride": pc.stride, "pad": list(pc.pad), "transposed": pc.transposed}
    emit(step="replay_target", idx=first_bad, **desc)
    for label, tune in [("default", base_tune)] + [(n, {**base_tune, k: base_tune[k] | b}) for n, k, b in toggles if k is not None]:
        ops.tune.update(tune)
        first, rows = one(), []
        for r in range(1, 10):
            for i, (x, y) in enumerate(zip(one(), first)):
                w = where_differs(x, y)
                if w["n"]:
                    rows.append({"replay": r, "output": i, **w})
        emit(step="replay", tune=label, n_differing=len(rows), first=rows[:4])
    ops.tune.update(base_tune)


if __name__ == "__main__":
    main()
