import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order: the per-kernel parity tests first, then the module / training-step parity, then the end-to-end goldens, and the
# full-size oracle comparisons and property tests LAST -- under `pytest -x` a red full-size test must not hide the kernel rows
# (round 4: one failing property test at collected position 31 left 273 op-level tests unrun).
_FILE_ORDER = ["test_abi.py", "test_isa_audit.py", "test_docs.py", "test_oracle_golden.py", "test_formats.py", "test_loss.py", "test_ops.py", "test_modules.py",
               "test_train.py", "test_trainer.py", "test_shard_gloo.py", "test_fusion.py", "test_model_cpu.py", "test_eval_gpu.py",
               "test_scene.py", "test_model_gpu.py"]
_LATE = ("full_size", "size_properties", "cfg4_full_size", "hip_graph")


def _order_key(item):
    fname = os.path.basename(str(item.fspath))
    rank = _FILE_ORDER.index(fname) if fname in _FILE_ORDER else len(_FILE_ORDER) - 1
    late = any(tag in item.name for tag in _LATE)
    return (2 * len(_FILE_ORDER) if late else 0) + rank


def pytest_collection_modifyitems(config, items):
    """Orders the suite (see _FILE_ORDER); a plain `pytest` in a container without a HIP device skips the gpu-marked tests instead
    of erroring."""
    items.sort(key=_order_key)          # stable: the order inside a file is kept
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a HIP device (MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """npz fixture with '@key' aliases resolved (see tests/golden/make_golden.py:dedupe)."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        self.files = set(self.z.files)

    def __contains__(self, k):
        return k in self.files

    def np(self, k):
        v = self.z[k]
        if v.dtype.kind == "U" and str(v).startswith("@"):
            return self.np(str(v)[1:])
        return v

    def t(self, k):
        return torch.from_numpy(np.ascontiguousarray(self.np(k)))

    def seq(self, prefix):
        n = int(self.z[prefix + ".len"])
        return [self.t(f"{prefix}.{i}") for i in range(n)]

    def meta(self):
        return json.loads(str(self.z["meta"]))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


def state_keys(variant):
    with open(os.path.join(GOLDEN, f"state_keys_{variant}.json")) as f:
        return json.load(f)


def conf_close(a, b, tol=1e-4, flips=0.01):
    """photometric confidence of stage 1 = 4 neighbouring probabilities gathered at floor(index) (reference
    module.py:569-571): discontinuous where the regressed index crosses an integer, so a small fraction of pixels may
    land in the neighbouring bin; everywhere else the maps must agree."""
    a, b = a.double().cpu(), b.double().cpu()
    assert a.shape == b.shape
    return float(((a - b).abs() > tol).double().mean()) < flips


def rel_l1(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().mean() / b.abs().mean().clamp_min(1e-30))


# ---------------------------------------------------------------------------------------
# backends for the kernel tests: "hip" = the product library on cuda:0 (GPU box, -m gpu);
# "emu" = the same kernel sources compiled for the host by tests/hipemu (CPU container).
_EMU = {}


def emu_ops():
    if "ops" not in _EMU:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from hipemu.build import build_emu
        from diffmvs_amd import _lib
        from diffmvs_amd.ops import Ops
        _EMU["ops"] = Ops(_lib.Lib(build_emu()), "cpu")
    return _EMU["ops"]


def hip_ops():
    from diffmvs_amd.ops import Ops
    assert torch.cuda.is_available(), "gpu-marked test needs a HIP device"
    return Ops.for_device("cuda:0")


def pin_ops(monkeypatch, ops):
    """Make `ops` the binding every model / module forward resolves (Ops.for_device) for the duration of a test: the CPU
    tests run the drop-in package on the host emulation this way; the product has no such switch."""
    from diffmvs_amd.ops import Ops
    monkeypatch.setattr(Ops, "for_device", classmethod(lambda cls, device: ops))


@pytest.fixture(params=[pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)])
def ops(request, monkeypatch):
    o = emu_ops() if request.param == "emu" else hip_ops()
    if request.param == "emu":
        pin_ops(monkeypatch, o)
    return o
