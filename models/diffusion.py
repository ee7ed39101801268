"""CasDiffMVS: the drop-in boundary (reference models/diffusion.py:9-295).

Constructor, forward(imgs, proj_matrices, depth_values, depth_gt_ms=None) signature, output
dict and state-dict layout are the reference's; the forward runs on the MI355X through
diffmvs_amd.engine.Engine (HIP kernels only -- no PyTorch-operator fallback, and a loud
error on a non-HIP device or a missing libdmvs_hip.so).
"""
from __future__ import annotations

import os

import torch
from torch import nn

from diffmvs_amd.engine import Engine
from diffmvs_amd.ops import Ops

from .module import ContextNet, Conv2d, FeatureNet, GetCost, InitialCost
from .update import DiffusionUpdateBlockDepth


class CasDiffMVS(nn.Module):
    """Implementation of DiffMVS (stage_iters[2] == 0) and CasDiffMVS."""

    def __init__(self, args, depth_interals_ratio=[4, 2, 1], test=False):
        super().__init__()
        self.numdepth_initial = args.numdepth_initial
        self.depth_interals_ratio = depth_interals_ratio
        self.args = args
        self.num_stage = 3
        self.cost_dim_stage = args.cost_dim_stage
        self.unet_dim = args.unet_dim
        self.unet_dim_mults = [(1,), (1, 2), (1, 2, 4)]
        self.test = test
        cas = args.stage_iters[2] != 0
        self.up_ratio = 2 if cas else 4          # final convex upsampling: 1/2 -> 1 or 1/4 -> 1
        self.CostNum = args.CostNum
        self.feat_dim_stage = [48, 32, 16 if cas else 0]
        self.hdim_stage, self.cdim_stage = args.hidden_dim, args.context_dim
        self.context_dim = [h + c for h, c in zip(self.hdim_stage, self.cdim_stage)]

        self.feature = FeatureNet(base_channels=8, out_channel=self.feat_dim_stage)
        self.context = ContextNet(self.context_dim)
        inits = [nn.Sequential(Conv2d(self.hdim_stage[1], 32, 3, 2, padding=1),
                               nn.Conv2d(32, self.hdim_stage[1], 3, 1, padding=1, bias=False))]
        if cas:
            inits.append(nn.Sequential(Conv2d(self.hdim_stage[2], 32, 3, 2, padding=1), Conv2d(32, 32, 3, 2, padding=1),
                                       nn.Conv2d(32, self.hdim_stage[2], 3, 1, padding=1, bias=False)))
        self.hidden_init = nn.ModuleList(inits)

        def block(stage):
            return DiffusionUpdateBlockDepth(
                args, dim=self.unet_dim[stage], dim_mults=self.unet_dim_mults[stage], hidden_dim=self.hdim_stage[stage],
                num_sample=self.CostNum[stage], cost_dim=self.cost_dim_stage[stage] * self.CostNum[stage],
                context_dim=self.cdim_stage[stage], stage_idx=stage, iters=args.stage_iters[stage], ratio=self.up_ratio)

        # registered twice on purpose: checkpoints carry both key sets (update_block_depthN.* and update_block.i.*)
        self.update_block_depth2 = block(1)
        blocks = [self.update_block_depth2]
        if cas:
            self.update_block_depth3 = block(2)
            blocks.append(self.update_block_depth3)
        self.update_block = nn.ModuleList(blocks)
        self.depthnet = InitialCost(self.cdim_stage[0], self.cost_dim_stage[0])
        self.GetCost = GetCost(self.cost_dim_stage[1], min_radius=args.min_radius, max_radius=args.max_radius)

        # diffusion noise: callable (shape, device) -> N(0,1) tensor.  None = device RNG (torch.randn on
        # the HIP device), the counterpart of torch.randn_like at reference update.py:472.  Parity tests
        # inject the oracle's noise stream here.
        self.noise_source = None
        # eval forward through a captured HIP graph (diffmvs_amd.engine.GraphedForward): worth it for small batches, where
        # ~350 launches per forward are host-bound; the returned tensors are then overwritten by the next call
        self.hip_graphs = os.environ.get("DMVS_GRAPHS", "0") == "1"
        self.t_source = None          # train mode: callable (B, timesteps, device) -> int64 [B]; None = torch.randint
        self._engine = None
        self._engine_key = None

    # ------------------------------------------------------------------ engine cache
    def _weights_key(self, device):
        # the tensor list is cached (walking the module tree costs more than a batch-1 forward's launches); _apply()
        # (.to / .cuda / .float) and load_state_dict(assign=True) are the calls that can replace tensors
        ts = self.__dict__.get("_key_tensors")
        if ts is None:
            ts = self.__dict__["_key_tensors"] = list(self.parameters()) + list(self.buffers())
        from diffmvs_amd.ops import weights_generation
        return (str(device), weights_generation()) + tuple((t.data_ptr(), t._version) for t in ts)

    def _apply(self, fn, *a, **k):
        self.__dict__.pop("_key_tensors", None)
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.__dict__.pop("_key_tensors", None)
        return super().load_state_dict(*a, **k)

    def engine(self, ops: Ops | None = None) -> Engine:
        """The packed inference engine for the current weights (rebuilt when they change)."""
        device = next(self.parameters()).device
        key = self._weights_key(device)
        if self._engine is None or self._engine_key != key or (ops is not None and ops is not self._engine.base_ops):
            if ops is None:
                ops = Ops.for_device(device)       # raises unless `device` is a HIP device and the .so exists
            self._engine = Engine(self.state_dict(), self.args, ops)
            self._engine_key = key
        return self._engine

    def scene_features(self, images, chunk: int = 64):
        """FeatureNet over all N images [N,3,H,W] of a scene, once -> diffmvs_amd.engine.SceneFeatureStore; pass
        `feats=store.gather(view_ids)` to forward() (view_ids [B,V], column 0 = the reference view).  Eval mode only."""
        from diffmvs_amd.engine import SceneFeatureStore
        return SceneFeatureStore(self.engine(), images, chunk)

    def forward(self, imgs, proj_matrices, depth_values, depth_gt_ms=None, feats=None):
        if self.training:
            # train branch (reference diffusion.py:167-172, update.py:423-464): an autograd graph whose convolution /
            # warp / cost-volume nodes are libdmvs_hip.so kernels in both directions (diffmvs_amd/train.py)
            if depth_gt_ms is None:
                raise ValueError("CasDiffMVS in train mode needs depth_gt_ms (reference diffusion.py:169)")
            if feats is not None:      # a scene's feature store is an eval-mode cache: the train branch differentiates through FeatureNet
                raise ValueError("CasDiffMVS.forward(feats=...) is the eval-mode scene cache; the train branch runs FeatureNet itself")
            from diffmvs_amd.train import forward_train
            ops = Ops.for_device(next(self.parameters()).device)
            return forward_train(self, imgs, proj_matrices, depth_values, depth_gt_ms, ops)
        eng = self.engine()
        if self.hip_graphs:
            return eng.forward_graphed(imgs, proj_matrices, depth_values, noise_fn=self.noise_source, test=self.test, feats=feats)
        return eng.forward(imgs, proj_matrices, depth_values, noise_fn=self.noise_source, test=self.test, feats=feats)
