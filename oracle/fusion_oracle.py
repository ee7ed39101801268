"""CPU restatement of the reference's depth-map fusion arithmetic (filter.py) in NumPy -- TEST INFRASTRUCTURE ONLY: imported
by tests/ to check the HIP kernel dmvs_geo_consistency_f32 and diffmvs_amd/fusion.py, never by the product.

Parity status: pinned EXCEPT the cv2.remap step.  filter.py imports cv2 and plyfile, neither of which exists in the build
container.  tests/golden/make_golden_fusion.py therefore runs the reference's own reproject_with_depth /
check_geometric_consistency / check_geometric_consistency_dynamic with cv2.remap replaced by remap_linear below (and an
empty plyfile) and commits what they return (tests/golden/fusion.npz); tests/test_fusion.py checks this restatement against
those arrays bit for bit -- the fp64 projection chain, the distance / relative-depth tests, the static and dynamic masks
are pinned to the reference's code (and tests/golden/fusion_tree.npz, the vertex tables of the reference's filter_depth /
filter_depth_dynamic on a small scene tree, pins the product's whole chain the same way).  cv2.remap(INTER_LINEAR) itself
remains UNPINNED: it is restated from OpenCV's
published implementation (imgwarp.cpp, opencv 4.x: fixed-point map conversion with INTER_BITS = 5, i.e. coordinates rounded
half-to-even to 1/32 pixel, fp32 tap weights (1-fy)(1-fx) ..., BORDER_CONSTANT 0 per tap) and there is no OpenCV here to
compare it with.  Everything follows filter.py line by line in the dtypes NumPy's promotion gives it there (fp32 camera
matrices and depth maps, int64 pixel grid => fp64 geometry)."""
import numpy as np

INTER_TAB = 32


def remap_linear(src, mapx, mapy):
    """cv2.remap(src, mapx, mapy, interpolation=cv2.INTER_LINEAR) for single-channel float32 src (filter.py:34-35)"""
    Hs, Ws = src.shape
    mx, my = mapx.astype(np.float32), mapy.astype(np.float32)
    far = ~((np.abs(mx) < np.float32(1e7)) & (np.abs(my) < np.float32(1e7)))
    sx = np.rint(np.where(far, 0, mx) * np.float32(INTER_TAB)).astype(np.int64)
    sy = np.rint(np.where(far, 0, my) * np.float32(INTER_TAB)).astype(np.int64)
    ix, iy = sx >> 5, sy >> 5
    fx = ((sx & 31).astype(np.float32)) * np.float32(1.0 / INTER_TAB)
    fy = ((sy & 31).astype(np.float32)) * np.float32(1.0 / INTER_TAB)
    one = np.float32(1.0)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < Ws) & (yy >= 0) & (yy < Hs)
        return np.where(ok, src[np.clip(yy, 0, Hs - 1), np.clip(xx, 0, Ws - 1)], np.float32(0.0)).astype(np.float32)

    out = tap(iy, ix) * ((one - fy) * (one - fx)) + tap(iy, ix + 1) * ((one - fy) * fx) + tap(iy + 1, ix) * (fy * (one - fx)) + \
        tap(iy + 1, ix + 1) * (fy * fx)
    return np.where(far, np.float32(0.0), out).astype(np.float32)


def reproject_with_depth(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src):
    """filter.py:8-53"""
    width, height = depth_ref.shape[1], depth_ref.shape[0]
    x_ref, y_ref = np.meshgrid(np.arange(0, width), np.arange(0, height))
    x_ref, y_ref = x_ref.reshape([-1]), y_ref.reshape([-1])
    xyz_ref = np.matmul(np.linalg.inv(intrinsics_ref), np.vstack((x_ref, y_ref, np.ones_like(x_ref))) * depth_ref.reshape([-1]))
    xyz_src = np.matmul(np.matmul(extrinsics_src, np.linalg.inv(extrinsics_ref)), np.vstack((xyz_ref, np.ones_like(x_ref))))[:3]
    K_xyz_src = np.matmul(intrinsics_src, xyz_src)
    with np.errstate(divide="ignore", invalid="ignore"):
        xy_src = K_xyz_src[:2] / K_xyz_src[2:3]
    x_src = xy_src[0].reshape([height, width]).astype(np.float32)
    y_src = xy_src[1].reshape([height, width]).astype(np.float32)
    sampled_depth_src = remap_linear(depth_src, x_src, y_src)
    xyz_src = np.matmul(np.linalg.inv(intrinsics_src), np.vstack((xy_src, np.ones_like(x_ref))) * sampled_depth_src.reshape([-1]))
    xyz_reprojected = np.matmul(np.matmul(extrinsics_ref, np.linalg.inv(extrinsics_src)), np.vstack((xyz_src, np.ones_like(x_ref))))[:3]
    depth_reproj = xyz_reprojected[2].reshape([height, width]).astype(np.float32)
    K_xyz_reprojected = np.matmul(intrinsics_ref, xyz_reprojected)
    K_xyz_reprojected = np.where(K_xyz_reprojected == 0, 1e-5, K_xyz_reprojected)
    with np.errstate(divide="ignore", invalid="ignore"):
        xy_reprojected = K_xyz_reprojected[:2] / K_xyz_reprojected[2:3]
    xy_reprojected = np.clip(xy_reprojected, -1e8, 1e8)
    x_reprojected = xy_reprojected[0].reshape([height, width]).astype(np.float32)
    y_reprojected = xy_reprojected[1].reshape([height, width]).astype(np.float32)
    return depth_reproj, x_reprojected, y_reprojected, x_src, y_src


def _dist_rel(depth_ref, depth_reproj, x2d_reproj, y2d_reproj):
    width, height = depth_ref.shape[1], depth_ref.shape[0]
    x_ref, y_ref = np.meshgrid(np.arange(0, width), np.arange(0, height))
    dist = np.sqrt((x2d_reproj - x_ref) ** 2 + (y2d_reproj - y_ref) ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        relative_depth_diff = np.abs(depth_reproj - depth_ref) / depth_ref
    return dist, relative_depth_diff


def check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src, ref_depth_max,
                                ref_depth_min, geo_pixel_thres=1.0, geo_depth_thres=0.01):
    """filter.py:56-93"""
    depth_reproj, x2d_reproj, y2d_reproj, x2d_src, y2d_src = reproject_with_depth(depth_ref, intrinsics_ref, extrinsics_ref, depth_src,
                                                                                  intrinsics_src, extrinsics_src)
    dist, rel = _dist_rel(depth_ref, depth_reproj, x2d_reproj, y2d_reproj)
    mask = np.logical_and(dist < geo_pixel_thres, rel < geo_depth_thres)
    mask = np.logical_and(mask, np.logical_and(depth_ref > ref_depth_min, depth_ref < ref_depth_max))
    depth_reproj[~mask] = 0
    return mask, depth_reproj, x2d_src, y2d_src


def check_geometric_consistency_dynamic(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src, dh):
    """filter.py:230-259; dh = [view_num, dist denominator, rel-diff denominator]"""
    depth_reproj, x2d_reproj, y2d_reproj, x2d_src, y2d_src = reproject_with_depth(depth_ref, intrinsics_ref, extrinsics_ref, depth_src,
                                                                                  intrinsics_src, extrinsics_src)
    dist, rel = _dist_rel(depth_ref, depth_reproj, x2d_reproj, y2d_reproj)
    masks = []
    for i in range(dh[0], 11):
        mask = np.logical_and(dist < i / dh[1], rel < i / dh[2])
        masks.append(mask)
    depth_reproj[~mask] = 0
    return masks, mask, depth_reproj, x2d_src, y2d_src


def fuse_view(ref_depth, ref_K, ref_E, depth_max, depth_min, confs, srcs, photo_thres, geo_mask_thres=3, geo_pixel_thres=1.0,
              geo_depth_thres=0.01, method="casdiffmvs"):
    """the per-reference-view arithmetic of filter_depth (filter.py:110-190) -> photo_mask, geo_mask, final_mask, averaged depth.
    confs: the conf0.. maps; srcs: [(depth, K, E)]"""
    if method == "casdiffmvs":
        photo_mask = (confs[0] > photo_thres[0]) & (confs[1] > photo_thres[1]) & (confs[2] > photo_thres[2])
    else:
        photo_mask = (confs[0] > photo_thres[0]) & (confs[1] > photo_thres[1])
    all_depth, geo_mask_sum = [], 0
    for d, k, e in srcs:
        geo_mask, depth_reproj, _, _ = check_geometric_consistency(ref_depth, ref_K, ref_E, d, k, e, depth_max, depth_min,
                                                                   geo_pixel_thres, geo_depth_thres)
        geo_mask_sum = geo_mask_sum + geo_mask.astype(np.int32)
        all_depth.append(depth_reproj)
    depth_est_averaged = (sum(all_depth) + ref_depth) / (geo_mask_sum + 1)
    geo_mask = geo_mask_sum >= geo_mask_thres
    return photo_mask, geo_mask, np.logical_and(photo_mask, geo_mask), depth_est_averaged


def fuse_view_dynamic(ref_depth, ref_K, ref_E, depth_max, depth_min, confs, srcs, photo_thres, dh, method="casdiffmvs"):
    """filter_depth_dynamic's per-view arithmetic (filter.py:319-385); dh = [view_num, dist, rel_diff]"""
    if method == "casdiffmvs":
        photo_mask = (confs[0] > photo_thres[0]) & (confs[1] > photo_thres[1]) & (confs[2] > photo_thres[2])
    else:
        photo_mask = (confs[0] > photo_thres[0]) & (confs[1] > photo_thres[2])
    all_depth, geo_mask_sum, sums = [], 0, None
    for d, k, e in srcs:
        masks, geo_mask, depth_reproj, _, _ = check_geometric_consistency_dynamic(ref_depth, ref_K, ref_E, d, k, e, dh)
        sums = [m.astype(np.int32) for m in masks] if sums is None else [s + m.astype(np.int32) for s, m in zip(sums, masks)]
        geo_mask_sum = geo_mask_sum + geo_mask.astype(np.int32)
        all_depth.append(depth_reproj)
    geo_mask = geo_mask_sum >= 10
    for i in range(dh[0], 11):
        geo_mask = np.logical_or(geo_mask, sums[i - dh[0]] >= i)
    depth_est_averaged = (sum(all_depth) + ref_depth) / (geo_mask_sum + 1)
    maskdepth = np.logical_and(depth_est_averaged >= depth_min, depth_est_averaged <= depth_max)
    final = np.logical_and(np.logical_and(photo_mask, geo_mask), maskdepth)
    return photo_mask, geo_mask, final, depth_est_averaged


def unproject(depth, K, E, mask):
    """filter.py:192-205: valid pixels -> world points [N,3]"""
    height, width = depth.shape[:2]
    x, y = np.meshgrid(np.arange(0, width), np.arange(0, height))
    x, y, d = x[mask], y[mask], depth[mask]
    xyz_ref = np.matmul(np.linalg.inv(K), np.vstack((x, y, np.ones_like(x))) * d)
    xyz_world = np.matmul(np.linalg.inv(E), np.vstack((xyz_ref, np.ones_like(x))))[:3]
    return xyz_world.transpose((1, 0))
