"""Deterministic synthetic scenes, weights and diffusion noise.

No dataset or checkpoint travels to the GPU box, so the bench, the parity tests
and the golden-vector generator (tests/golden/make_golden.py, which feeds the
same tensors to the imported reference) all draw from here.  Everything is
seeded through numpy's frozen legacy ``RandomState`` so a fixture generated in
one container is reproducible bit-for-bit in another.

Sample contract mirrored: reference datasets/mvs.py:129-210
  imgs            list of V tensors [B,3,H,W] in [0,1], index 0 = reference view
  proj_matrices   {"stage1".."stage4": [B,V,2,4,4]}; [:, :, 0] = extrinsic 4x4,
                  [:, :, 1, :3, :3] = intrinsics with rows 0-1 scaled by
                  0.125 / 0.25 / 0.5 / 1  (datasets/mvs.py:157-185)
  depth_values    [B,numdepth] inverse depths ascending (datasets/mvs.py:163-166)
"""
from __future__ import annotations

import zlib
from types import SimpleNamespace

import numpy as np
import torch

# hyper-parameter sets: reference scripts/test/test_dtu_diffmvs.sh:13-21 and
# scripts/test/test_dtu_casdiffmvs.sh:13-21 (the only place they are written down)
_VARIANTS = {
    "diffmvs": dict(
        stage_iters=[1, 4, 0], cost_dim_stage=[4, 4, 0], CostNum=[0, 6, 0],
        hidden_dim=[0, 32, 0], context_dim=[32, 32, 0], unet_dim=[0, 16, 8],
        min_radius=0.25, max_radius=4.0,
        scale=[0.0, 0.5, 0.0], sampling_timesteps=[0, 1, 1], ddim_eta=[0.0, 1.0, 0.0],
    ),
    "casdiffmvs": dict(
        stage_iters=[1, 3, 3], cost_dim_stage=[4, 4, 4], CostNum=[0, 4, 4],
        hidden_dim=[0, 32, 20], context_dim=[32, 32, 16], unet_dim=[0, 16, 8],
        min_radius=0.125, max_radius=8.0,
        scale=[0.0, 0.5, 0.1], sampling_timesteps=[0, 1, 1], ddim_eta=[0.0, 1.0, 1.0],
    ),
}


def make_args(variant: str = "diffmvs", numdepth_initial: int = 48, numdepth: int = 384,
              **overrides) -> SimpleNamespace:
    """The argparse namespace the model constructor reads (reference test.py:20-78)."""
    a = dict(_VARIANTS[variant])
    a.update(numdepth_initial=numdepth_initial, numdepth=numdepth,
             timesteps=[1000, 1000, 1000], conf_weight=0.05)
    a.update(overrides)
    return SimpleNamespace(**a)


# --------------------------------------------------------------------------- scene
def _texture(xw: np.ndarray, yw: np.ndarray, rs: np.random.RandomState) -> np.ndarray:
    """Procedural RGB texture on the scene plane, in [0,1]."""
    out = np.zeros((3,) + xw.shape, np.float64)
    for c in range(3):
        acc = np.zeros_like(xw, dtype=np.float64)
        for _ in range(7):
            fx, fy = rs.uniform(-0.35, 0.35, 2)
            ph = rs.uniform(0, 2 * np.pi)
            acc += rs.uniform(0.3, 1.0) * np.sin(fx * xw + fy * yw + ph)
        out[c] = acc
    out = 0.5 + 0.5 * np.tanh(0.6 * out)
    return out


def synth_inputs(H: int, W: int, n_src: int, B: int = 1, seed: int = 0, numdepth: int = 384,
                 depth_min: float = 425.0, depth_max: float = 935.0, device="cpu", with_gt: bool = False):
    """One batch of B reference views, each with n_src source views.

    Geometry follows SURVEY section 8d: K = [[1.2W,0,W/2],[0,1.2W,H/2],[0,0,1]], view v rotated
    about y by 0.05*v rad and shifted t_x = -30*v (mm); DTU depth range 425..935.
    Images are renderings of one slanted textured plane, so that matching peaks exist.
    """
    V = n_src + 1
    imgs = np.zeros((V, B, 3, H, W), np.float32)
    proj = {s: np.zeros((B, V, 2, 4, 4), np.float32) for s in ("stage1", "stage2", "stage3", "stage4")}
    scales = {"stage1": 0.125, "stage2": 0.25, "stage3": 0.5, "stage4": 1.0}
    K = np.array([[1.2 * W, 0, W / 2.0], [0, 1.2 * W, H / 2.0], [0, 0, 1.0]], np.float64)
    Kinv = np.linalg.inv(K)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    pix = np.stack([xs.ravel(), ys.ravel(), np.ones(H * W)])
    for b in range(B):
        rs = np.random.RandomState(1000003 * seed + 7919 * b + 17)
        # world plane z = d0 + a x + c y, inside the depth range
        d0 = rs.uniform(560.0, 760.0)
        a, c = rs.uniform(-0.25, 0.25, 2)
        tex_seed = rs.randint(0, 2 ** 31 - 1)
        for v in range(V):
            ang = 0.05 * v * (1.0 + 0.1 * b)
            R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
            t = np.array([-30.0 * v, 4.0 * v * ((-1) ** v), 0.0])
            E = np.eye(4)
            E[:3, :3] = R
            E[:3, 3] = t
            # ray through each pixel in world coordinates: X = o + s * d
            o = -R.T @ t
            d = R.T @ (Kinv @ pix)
            # plane: X_z - a X_x - c X_y = d0
            n = np.array([-a, -c, 1.0])
            s = (d0 - n @ o) / (n @ d)
            Xw = o[:, None] + s[None, :] * d
            tex = _texture(Xw[0].reshape(H, W), Xw[1].reshape(H, W), np.random.RandomState(tex_seed))
            noise = rs.uniform(-0.03, 0.03, size=(3, H, W))
            imgs[v, b] = np.clip(tex + noise, 0.0, 1.0).astype(np.float32)
            for sname, sc in scales.items():
                Ks = K.copy()
                Ks[:2, :] *= sc
                proj[sname][b, v, 0] = E.astype(np.float32)
                proj[sname][b, v, 1, :3, :3] = Ks.astype(np.float32)
    dv = np.linspace(1.0 / depth_max, 1.0 / depth_min, numdepth, dtype=np.float32)
    depth_values = np.tile(dv[None], (B, 1))
    imgs_t = [torch.from_numpy(imgs[v]).to(device) for v in range(V)]
    proj_t = {k: torch.from_numpy(p).to(device) for k, p in proj.items()}
    if not with_gt:
        return imgs_t, proj_t, torch.from_numpy(depth_values).to(device)
    # ground-truth depth of the reference view (camera 0 = world frame) on the scene plane, per stage, with a
    # band of invalid (0) pixels, plus the validity masks (contract: reference datasets/dtu.py, train.py:183-188)
    gt, mask = {}, {}
    for sname, sc in scales.items():
        h, w = int(H * sc), int(W * sc)
        Ks = K.copy()
        Ks[:2, :] *= sc
        yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
        rays = np.linalg.inv(Ks) @ np.stack([xx.ravel(), yy.ravel(), np.ones(h * w)])
        g = np.zeros((B, h, w), np.float32)
        for b in range(B):
            rs = np.random.RandomState(1000003 * seed + 7919 * b + 17)
            d0 = rs.uniform(560.0, 760.0)
            a, c = rs.uniform(-0.25, 0.25, 2)
            n = np.array([-a, -c, 1.0])
            g[b] = (d0 / (n @ rays)).reshape(h, w).astype(np.float32)
            g[b, : max(1, h // 8)] = 0.0
        gt[sname] = torch.from_numpy(g).to(device)
        mask[sname] = torch.from_numpy(((g > depth_min) & (g < depth_max)).astype(np.float32)).to(device)
    return imgs_t, proj_t, torch.from_numpy(depth_values).to(device), gt, mask


# --------------------------------------------------------------------------- weights
SCHEDULE_BUFFERS = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas",
    "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
)


def _canon(key: str) -> str:
    """update_block.i aliases update_block_depth{i+2} (reference diffusion.py:59-71,128-129)."""
    if key.startswith("update_block."):
        i, rest = key[len("update_block."):].split(".", 1)
        return f"update_block_depth{int(i) + 2}.{rest}"
    return key


def synth_state_dict(template: dict, seed: int = 123) -> dict:
    """Fill every entry of a state-dict *template* (key -> tensor of the right shape) with a
    value that depends only on (seed, canonical key, shape).  Diffusion-schedule buffers are
    left as the model computed them.  BN running stats / affine terms are non-trivial on
    purpose so that BN folding is actually tested.
    """
    out = {}
    for key, ref in template.items():
        leaf = key.rsplit(".", 1)[-1]
        if leaf in SCHEDULE_BUFFERS:
            out[key] = ref.clone()
            continue
        shape = tuple(ref.shape)
        rs = np.random.RandomState((zlib.crc32(_canon(key).encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        if leaf == "num_batches_tracked":
            val = np.zeros(shape, np.int64)
        elif leaf == "running_var":
            val = rs.uniform(0.5, 1.5, shape)
        elif leaf == "running_mean":
            val = rs.uniform(-0.2, 0.2, shape)
        elif leaf == "bias":
            val = rs.uniform(-0.1, 0.1, shape)
            if key.endswith("unet.final_conv.bias") or key.endswith("unet.conf.bias"):
                val = val * 0.1
        elif len(shape) <= 1:                      # norm scales (BN / GroupNorm weight)
            val = rs.uniform(0.5, 1.5, shape)
        else:                                      # conv / linear weights, He-uniform
            fan_in = int(np.prod(shape[1:]))
            bound = np.sqrt(6.0 / fan_in)
            val = rs.uniform(-bound, bound, shape)
            if key.endswith("unet.final_conv.weight"):
                val = val * 0.005                  # keep the refinement delta well inside [0,1]
            if key.endswith("unet.conf.weight"):
                val = val * 0.02                   # confidences around 0.5: keeps 1/(1-conf) in the loss well conditioned
            if key.endswith("cost_regularization.prob.weight"):
                val = val * 4.0                    # make the softmax over depth peaky enough to matter
        out[key] = torch.from_numpy(np.asarray(val)).to(ref.dtype)
    return out


def synth_noise(shape, seed: int, index: int) -> torch.Tensor:
    """Standard-normal draw number *index* of a forward pass (reference update.py:472, :515)."""
    rs = np.random.RandomState((seed * 7919 + index * 104729 + 13) & 0x7FFFFFFF)
    return torch.from_numpy(rs.standard_normal(tuple(shape)).astype(np.float32))


class NoiseSource:
    """Callable handed to CasDiffMVS.noise_source for reproducible parity runs."""

    def __init__(self, seed: int = 0):
        self.seed = seed
        self.index = 0

    def reset(self):
        self.index = 0

    def __call__(self, shape, device):
        n = synth_noise(shape, self.seed, self.index)
        self.index += 1
        return n.to(device)


def getcost_scene_inputs(H, W, n_src, B, stage=2, C=32, noise=0.01, conf=0.5, seed=0, numdepth=None):
    """GetCost inputs with the geometry a trained network produces: inverse-depth hypotheses centred on the synthetic
    scene's true depth (+ Gaussian noise of `noise` in normalised inverse depth), random features / view weights.
    Returns CPU tensors: ref [B,h,w,C], src [S,B,h,w,C], proj (stage matrices [B,V,2,4,4]), inv [B,1,h,w], conf [B,h,w] or
    None, view_w [B,S,h>>vs,w>>vs], disp_min [B], disp_max [B], interval, vw_shift."""
    imgs, proj, dv, gt, _ = synth_inputs(H, W, n_src, B=B, seed=seed, with_gt=True)
    name = f"stage{stage}"
    sc = 2 ** (4 - stage)
    h, w = H // sc, W // sc
    g = torch.Generator().manual_seed(seed + 1)
    ref = torch.randn(B, h, w, C, generator=g)
    src = torch.randn(n_src, B, h, w, C, generator=g)
    kmin, kmax = dv[:, 0].contiguous(), dv[:, -1].contiguous()
    d = gt[name]
    d = torch.where(torch.isfinite(d) & (d > 0), d, torch.full_like(d, 600.0))
    inv = ((1.0 / d) - kmin.view(-1, 1, 1)) / (kmax - kmin).view(-1, 1, 1)
    inv = (inv + noise * torch.randn(inv.shape, generator=g)).clamp(0, 1).unsqueeze(1).contiguous()
    cf = torch.full((B, h, w), float(conf)) if conf is not None and conf >= 0 else None
    vshift = stage - 1
    vw = torch.rand(B, n_src, h >> vshift, w >> vshift, generator=g)
    interval = (1.0 / dv.shape[1]) * (2 if stage == 2 else 1)
    return {"ref": ref, "src": src, "proj": proj[name], "inv": inv, "conf": cf, "view_w": vw, "disp_min": kmin, "disp_max": kmax,
            "interval": interval, "vw_shift": vshift}


def synth_view_depths(H: int, W: int, n_views: int, seed: int = 0, b: int = 0) -> np.ndarray:
    """depth map [V,H,W] (z in each camera's own frame) of the scene plane of synth_inputs(seed)[batch item b] for every
    view: the input of the fusion tests (a depth map per view, as the reference's test.py leaves them on disk)"""
    K = np.array([[1.2 * W, 0, W / 2.0], [0, 1.2 * W, H / 2.0], [0, 0, 1.0]], np.float64)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    rays = np.linalg.inv(K) @ np.stack([xs.ravel(), ys.ravel(), np.ones(H * W)])
    rs = np.random.RandomState(1000003 * seed + 7919 * b + 17)
    d0 = rs.uniform(560.0, 760.0)
    a, c = rs.uniform(-0.25, 0.25, 2)
    n = np.array([-a, -c, 1.0])
    out = np.zeros((n_views, H, W), np.float32)
    for v in range(n_views):
        ang = 0.05 * v * (1.0 + 0.1 * b)
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        t = np.array([-30.0 * v, 4.0 * v * ((-1) ** v), 0.0])
        o = -R.T @ t
        d = R.T @ rays
        out[v] = ((d0 - n @ o) / (n @ d)).reshape(H, W).astype(np.float32)       # ray parameter = camera-frame depth (ray z = 1)
    return out


def synth_cameras(H: int, W: int, n_src: int, B: int = 1, numdepth: int = 384, depth_min: float = 425.0, depth_max: float = 935.0):
    """the cameras and depth range of synth_inputs(H, W, n_src, B) without rendering the images (seed-independent):
    -> proj {stage1..4: [B,V,2,4,4]}, depth_values [B,numdepth]"""
    V = n_src + 1
    proj = {s: np.zeros((B, V, 2, 4, 4), np.float32) for s in ("stage1", "stage2", "stage3", "stage4")}
    scales = {"stage1": 0.125, "stage2": 0.25, "stage3": 0.5, "stage4": 1.0}
    K = np.array([[1.2 * W, 0, W / 2.0], [0, 1.2 * W, H / 2.0], [0, 0, 1.0]], np.float64)
    for b in range(B):
        for v in range(V):
            ang = 0.05 * v * (1.0 + 0.1 * b)
            R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
            E = np.eye(4)
            E[:3, :3] = R
            E[:3, 3] = np.array([-30.0 * v, 4.0 * v * ((-1) ** v), 0.0])
            for sname, sc in scales.items():
                Ks = K.copy()
                Ks[:2, :] *= sc
                proj[sname][b, v, 0] = E.astype(np.float32)
                proj[sname][b, v, 1, :3, :3] = Ks.astype(np.float32)
    dv = np.tile(np.linspace(1.0 / depth_max, 1.0 / depth_min, numdepth, dtype=np.float32)[None], (B, 1))
    return {k: torch.from_numpy(p) for k, p in proj.items()}, torch.from_numpy(dv)


# --------------------------------------------------------------------------- a whole synthetic SCENE (scene-mode evaluation)
def synth_scene(H: int, W: int, n_views: int = 49, n_src: int = 5, seed: int = 0, numdepth: int = 384, depth_min: float = 425.0,
                depth_max: float = 935.0, grid_w: int = 7):
    """A DTU-shaped scene: n_views cameras on a grid_w-wide grid (30 mm spacing, each turned 0.05 rad per grid step towards the
    scene, like synth_inputs' rig between neighbours) looking at ONE slanted textured plane, and a pair table like the datasets'
    pair.txt (datasets/mvs.py:33-47): for every reference view its n_src nearest grid neighbours, nearest first.
    -> {"images": [N,3,H,W], "K": [N,3,3], "E": [N,4,4], "pairs": int64 [N,n_src], "depth_values": [numdepth]}"""
    rs = np.random.RandomState(1000003 * seed + 101)
    d0 = rs.uniform(560.0, 760.0)
    a, c = rs.uniform(-0.25, 0.25, 2)
    tex_seed = rs.randint(0, 2 ** 31 - 1)
    K = np.array([[1.2 * W, 0, W / 2.0], [0, 1.2 * W, H / 2.0], [0, 0, 1.0]], np.float64)
    Kinv = np.linalg.inv(K)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    pix = np.stack([xs.ravel(), ys.ravel(), np.ones(H * W)])
    imgs = np.zeros((n_views, 3, H, W), np.float32)
    Es = np.zeros((n_views, 4, 4), np.float32)
    grid = np.array([(v % grid_w, v // grid_w) for v in range(n_views)], np.float64)
    centre = grid.mean(0)
    n = np.array([-a, -c, 1.0])
    for v in range(n_views):
        gx, gy = grid[v] - centre
        ay, ax = 0.05 * gx, -0.02 * gy
        Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
        Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
        R = Rx @ Ry
        t = np.array([-30.0 * gx, -30.0 * gy, 0.0])
        E = np.eye(4)
        E[:3, :3], E[:3, 3] = R, t
        o = -R.T @ t
        d = R.T @ (Kinv @ pix)
        s = (d0 - n @ o) / (n @ d)
        Xw = o[:, None] + s[None, :] * d
        tex = _texture(Xw[0].reshape(H, W), Xw[1].reshape(H, W), np.random.RandomState(tex_seed))
        imgs[v] = np.clip(tex + rs.uniform(-0.03, 0.03, size=(3, H, W)), 0.0, 1.0).astype(np.float32)
        Es[v] = E.astype(np.float32)
    pairs = np.zeros((n_views, n_src), np.int64)
    for v in range(n_views):
        dist = np.abs(grid - grid[v]).sum(1) + 1e-3 * np.arange(n_views)      # (ties: lower view id first)
        dist[v] = 1e9
        order = np.argsort(dist, kind="stable")
        pairs[v] = order[np.arange(n_src) % max(1, n_views - 1)]
    return {"images": torch.from_numpy(imgs), "K": torch.from_numpy(K.astype(np.float32))[None].repeat(n_views, 1, 1), "E": torch.from_numpy(Es),
            "pairs": torch.from_numpy(pairs), "depth_values": torch.from_numpy(np.linspace(1.0 / depth_max, 1.0 / depth_min, numdepth, dtype=np.float32))}


def scene_batch(scene: dict, ref_ids):
    """the model inputs of reference views `ref_ids` of a synth_scene (what the dataset + collate would hand over):
    -> imgs (V tensors [B,3,H,W]), proj {stage1..4: [B,V,2,4,4]}, depth_values [B,numdepth], view_ids int64 [B,V] (column 0 = ref)"""
    ref_ids = torch.as_tensor(ref_ids, dtype=torch.int64)
    view_ids = torch.cat([ref_ids[:, None], scene["pairs"][ref_ids]], 1)
    B, V = view_ids.shape
    imgs = [scene["images"][view_ids[:, v]] for v in range(V)]
    proj = {}
    for sname, sc in (("stage1", 0.125), ("stage2", 0.25), ("stage3", 0.5), ("stage4", 1.0)):
        p = torch.zeros(B, V, 2, 4, 4)
        p[:, :, 0] = scene["E"][view_ids]
        Ks = scene["K"][view_ids].clone()
        Ks[:, :, :2, :] *= sc
        p[:, :, 1, :3, :3] = Ks
        proj[sname] = p
    return imgs, proj, scene["depth_values"][None].repeat(B, 1), view_ids
