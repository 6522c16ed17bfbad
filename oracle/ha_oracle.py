"""
TEST INFRASTRUCTURE ONLY (see oracle/mp_oracle.py for the rules: the product never imports this).

CPU restatement of the homographic adaptation of ethz-asl/multipoint
(multipoint/utils/homographies.py:38-189, 361-433; driver export_keypoints.py:64-103).

The reference's own code on this path is the driver loop (pinned: tests/test_oracle_vs_reference.py runs the imported
reference driver, with the third-party calls below supplied by this file, against `homographic_adaptation` here) and
`sample_homography` (pinned: the product's host function is compared with the imported reference draw for draw, and
golden matrices are committed).  The pixel arithmetic lives in third-party packages that are absent from
/root/reference and from this image -> "PARITY UNPINNED" at these boundaries, published algorithms restated:

  * kornia (unpinned, optional import, homographies.py:5-10): `warp_perspective` =
    `homography_warp(src, inv(dst_norm_to_dst_norm(M)), dsize)` = a [-1,1] mesh grid pushed through the normalised
    homography and `F.grid_sample(..., align_corners=True)`            -> `kornia_homography_warp`, `dst_norm_to_dst_norm`
  * opencv-python==4.2.0.34 (requirements.txt:1): `cv2.getPerspectiveTransform` (8x8 system, h33 = 1),
    `cv2.warpPerspective(..., INTER_NEAREST)` (inverts M, cvRound of the source coordinate, constant border 0),
    `cv2.erode` (minimum over the kernel window, border = +inf)         -> `cv2_*`
  * the dataset's homographic augmentation (multipoint/datasets/augmentation/augmentation.py:25-54):
    `cv2.warpPerspective(image, M, size, borderMode=...)` with INTER_LINEAR (1/32-pixel fixed-point source
    coordinates, float32 weight table)                                   -> `cv2_warp_perspective_linear`
"""
import copy
from math import pi

import numpy as np
import torch
import torch.nn.functional as F

# homographies.py:17-36
DEFAULT_CONFIG = {
    'num': 100,
    'aggregation': 'prod',
    'homographies': {'translation': True, 'rotation': True, 'scaling': True, 'perspective': True,
                     'scaling_amplitude': 0.15, 'perspective_amplitude_x': 0.15, 'perspective_amplitude_y': 0.15,
                     'patch_ratio': 0.9, 'max_angle': pi, 'allow_artifacts': True},
    'erosion_radius': 5,
    'mask_border': True,
    'min_count': 2,
    'filter_size': 0,
}


# augmentation.homographic block of the reference's configs/config_image_pair_dataset_prediction.yaml:20-36
PREDICTION_AUGMENTATION = {
    'enable': True,
    'params': {'translation': True, 'rotation': True, 'scaling': True, 'perspective': True,
               'scaling_amplitude': 0.2, 'perspective_amplitude_x': 0.2, 'perspective_amplitude_y': 0.2,
               'patch_ratio': 0.85, 'max_angle': 1.57, 'allow_artifacts': True, 'translation_overflow': 0.05},
    'valid_border_margin': 0, 'border_reflect': True, 'mask_border': True}


def full_config(cfg=None):
    out = copy.deepcopy(DEFAULT_CONFIG)
    for k, v in (cfg or {}).items():
        if isinstance(v, dict):
            out[k].update(v)
        else:
            out[k] = v
    return out


# ------------------------------------------------------------------------------------------------------------------
# OpenCV stand-ins
# ------------------------------------------------------------------------------------------------------------------
def cv2_get_perspective_transform(src, dst):
    """cv2.getPerspectiveTransform(src, dst) for float32 (4,2) points (call site homographies.py:326): solves
    u = (c00 x + c01 y + c02) / (c20 x + c21 y + 1), v likewise, as an 8x8 linear system in float64."""
    src = np.asarray(src, np.float32).astype(np.float64)
    dst = np.asarray(dst, np.float32).astype(np.float64)
    A = np.zeros((8, 8))
    b = np.zeros(8)
    for i in range(4):
        A[i, 0], A[i, 1], A[i, 2] = src[i, 0], src[i, 1], 1
        A[i, 6], A[i, 7] = -src[i, 0] * dst[i, 0], -src[i, 1] * dst[i, 0]
        A[i + 4, 3], A[i + 4, 4], A[i + 4, 5] = src[i, 0], src[i, 1], 1
        A[i + 4, 6], A[i + 4, 7] = -src[i, 0] * dst[i, 1], -src[i, 1] * dst[i, 1]
        b[i], b[i + 4] = dst[i, 0], dst[i, 1]
    x = np.linalg.solve(A, b)
    return np.concatenate([x, [1.0]]).reshape(3, 3)


def cv2_warp_perspective_nearest(src, M, dsize):
    """cv2.warpPerspective(src, M, dsize, flags=cv2.INTER_NEAREST) (call site homographies.py:377): dsize = (W, H);
    dst(x, y) = src(round(M^-1 (x, y, 1))) with cvRound (half to even) and a zero constant border."""
    src = np.asarray(src)
    W, H = int(dsize[0]), int(dsize[1])
    Mi = np.linalg.inv(np.asarray(M, np.float64))
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    den = Mi[2, 0] * xs + Mi[2, 1] * ys + Mi[2, 2]
    with np.errstate(divide='ignore', invalid='ignore'):
        iw = np.where(den != 0, 1.0 / den, 0.0)
    fx = np.clip((Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]) * iw, -2147483648.0, 2147483647.0)
    fy = np.clip((Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]) * iw, -2147483648.0, 2147483647.0)
    rx, ry = np.rint(fx).astype(np.int64), np.rint(fy).astype(np.int64)
    ok = (rx >= 0) & (rx < src.shape[1]) & (ry >= 0) & (ry < src.shape[0])
    out = np.zeros((H, W), dtype=src.dtype)
    out[ok] = src[ry[ok], rx[ok]]
    return out


def cv2_invert3(M):
    """cv::invert of a 3x3 float64 matrix (DECOMP_LU takes the closed-form adjugate path for n <= 3), which
    cv2.warpPerspective applies to M before mapping destination to source pixels.  Singular -> zeros."""
    S = np.asarray(M, np.float64).reshape(3, 3)
    d = (S[0, 0] * (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) - S[0, 1] * (S[1, 0] * S[2, 2] - S[1, 2] * S[2, 0]) +
         S[0, 2] * (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]))
    if d == 0.0:
        return np.zeros((3, 3))
    d = 1.0 / d
    t = [(S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) * d, (S[0, 2] * S[2, 1] - S[0, 1] * S[2, 2]) * d,
         (S[0, 1] * S[1, 2] - S[0, 2] * S[1, 1]) * d, (S[1, 2] * S[2, 0] - S[1, 0] * S[2, 2]) * d,
         (S[0, 0] * S[2, 2] - S[0, 2] * S[2, 0]) * d, (S[0, 2] * S[1, 0] - S[0, 0] * S[1, 2]) * d,
         (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]) * d, (S[0, 1] * S[2, 0] - S[0, 0] * S[2, 1]) * d,
         (S[0, 0] * S[1, 1] - S[0, 1] * S[1, 0]) * d]
    return np.array(t, np.float64).reshape(3, 3)


def _cv2_border_101(p, n):
    """cv::borderInterpolate(p, n, BORDER_REFLECT_101) on an int64 array."""
    if n == 1:
        return np.zeros_like(p)
    p = p.copy()
    while True:
        bad = (p < 0) | (p >= n)
        if not bad.any():
            return p
        p = np.where(p < 0, -p, np.where(p >= n, 2 * n - 2 - p, p))


def cv2_warp_perspective_linear(src, M, dsize, border='reflect101'):
    """cv2.warpPerspective(src, M, dsize, borderMode=BORDER_REFLECT_101 | BORDER_CONSTANT) with the default
    INTER_LINEAR on a float32 image (call site multipoint/datasets/augmentation/augmentation.py:33-36), restated
    from OpenCV 4.2's WarpPerspectiveInvoker + remapBilinear<float>:

      M <- invert(M);  destination blocks are bw0 = min(1024 / min(16, H), W) wide; for the block starting at xb,
      X0 = M00*xb + M01*y + M02 (Y0, W0 alike, float64);  per pixel x1 = x - xb:  W = 32 / (W0 + M20*x1) (0 if the
      denominator is 0),  X = cvRound(clamp((X0 + M00*x1) * W, INT_MIN, INT_MAX))  -- coordinates in 1/32 px;
      source pixel (X >> 5, Y >> 5) saturated to int16, table index (Y & 31, X & 31);
      value = v00*w0 + v01*w1 + v10*w2 + v11*w3 in float32, left to right, weights
      (1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy*fx computed in float32 from f = idx/32.
    Only dsize == src size is restated (the block width depends on the destination size)."""
    src = np.ascontiguousarray(src, dtype=np.float32)
    H, W = src.shape
    assert (int(dsize[0]), int(dsize[1])) == (W, H)
    Mi = cv2_invert3(M)
    bh0 = min(16, H)
    bw0 = min(1024 // bh0, W)
    ys, xs = np.mgrid[0:H, 0:W]
    xb = ((xs // bw0) * bw0).astype(np.float64)
    x1 = xs.astype(np.float64) - xb
    yf = ys.astype(np.float64)
    X0 = (Mi[0, 0] * xb + Mi[0, 1] * yf) + Mi[0, 2]
    Y0 = (Mi[1, 0] * xb + Mi[1, 1] * yf) + Mi[1, 2]
    W0 = (Mi[2, 0] * xb + Mi[2, 1] * yf) + Mi[2, 2]
    den = W0 + Mi[2, 0] * x1
    with np.errstate(divide='ignore', invalid='ignore'):
        w = np.where(den != 0, 32.0 / den, 0.0)
    with np.errstate(invalid='ignore', over='ignore'):
        fX = np.clip((X0 + Mi[0, 0] * x1) * w, -2147483648.0, 2147483647.0)
        fY = np.clip((Y0 + Mi[1, 0] * x1) * w, -2147483648.0, 2147483647.0)
    X, Y = np.rint(fX).astype(np.int64), np.rint(fY).astype(np.int64)
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    fx = ((X & 31).astype(np.float32) * np.float32(1.0 / 32)).astype(np.float32)
    fy = ((Y & 31).astype(np.float32) * np.float32(1.0 / 32)).astype(np.float32)
    ax, ay = (np.float32(1) - fx).astype(np.float32), (np.float32(1) - fy).astype(np.float32)
    w0, w1, w2, w3 = ay * ax, ay * fx, fy * ax, fy * fx

    def tap(yy, xx):
        if border == 'constant':
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            return np.where(ok, src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], np.float32(0)).astype(np.float32)
        assert border == 'reflect101'
        return src[_cv2_border_101(yy, H), _cv2_border_101(xx, W)]

    out = tap(sy, sx) * w0
    out = (out + tap(sy, sx + 1) * w1).astype(np.float32)
    out = (out + tap(sy + 1, sx) * w2).astype(np.float32)
    out = (out + tap(sy + 1, sx + 1) * w3).astype(np.float32)
    return out


def warp_keypoints_int(keypoints, homography):
    """warp_keypoints (homographies.py:329-346): cv2.perspectiveTransform in float64, truncated to int."""
    keypoints = np.asarray(keypoints)
    if len(keypoints) == 0:
        return keypoints
    h = np.asarray(homography, np.float64)
    xy = keypoints[:, ::-1].astype(np.float64)
    w = h[2, 0] * xy[:, 0] + h[2, 1] * xy[:, 1] + h[2, 2]
    with np.errstate(divide='ignore', invalid='ignore'):
        iw = np.where(w != 0, 1.0 / w, 0.0)
    x = (h[0, 0] * xy[:, 0] + h[0, 1] * xy[:, 1] + h[0, 2]) * iw
    y = (h[1, 0] * xy[:, 0] + h[1, 1] * xy[:, 1] + h[1, 2]) * iw
    return np.stack([y, x], axis=1).astype(int)


def homographic_augmentation(image, keypoints, homography, border_reflect=True, valid_border_margin=0,
                             mask_border=True):
    """multipoint/datasets/augmentation/augmentation.py:25-54 for an already-sampled homography: warped image,
    warped + in-frame keypoints (or None), valid mask."""
    H, W = image.shape
    warped = cv2_warp_perspective_linear(image, homography, (W, H), 'reflect101' if border_reflect else 'constant')
    mask = compute_valid_mask((H, W), homography, valid_border_margin * 2, mask_border)
    pts = None
    if keypoints is not None:
        pts = keypoints
        if keypoints.size > 0:
            pts = warp_keypoints_int(keypoints, homography)
            pts = pts[(pts[:, 0] >= 0) & (pts[:, 1] >= 0) & (pts[:, 0] < H) & (pts[:, 1] < W)]
    return warped, pts, mask


def cv2_erode(src, kernel, iterations=1):
    """cv2.erode(src, ones((k,k)), iterations=1) (call site homographies.py:385): minimum over the kernel window
    anchored at its centre; pixels outside the image do not take part (default border value +inf)."""
    assert iterations == 1
    src = np.asarray(src)
    kh, kw = kernel.shape
    assert np.all(kernel != 0)
    ry, rx = kh // 2, kw // 2
    pad = np.pad(src.astype(np.float64), ((ry, ry), (rx, rx)), constant_values=np.inf)
    out = np.full(src.shape, np.inf)
    for dy in range(kh):
        for dx in range(kw):
            out = np.minimum(out, pad[dy:dy + src.shape[0], dx:dx + src.shape[1]])
    return out.astype(src.dtype)


def compute_valid_mask(image_shape, homography, erosion_radius=0, mask_border=False):
    """homographies.py:361-389."""
    H, W = int(image_shape[0]), int(image_shape[1])
    mask = cv2_warp_perspective_nearest(np.ones((H, W)), homography, (W, H))
    if erosion_radius > 0:
        if mask_border:
            ring = np.zeros((H + 2, W + 2))
            ring[1:-1, 1:-1] = mask
            mask = ring
        mask = cv2_erode(mask, np.ones((erosion_radius * 2 + 1,) * 2, np.float32))
        if mask_border:
            mask = mask[1:-1, 1:-1]
    return mask


# ------------------------------------------------------------------------------------------------------------------
# kornia stand-ins
# ------------------------------------------------------------------------------------------------------------------
def _normal_transform_pixel(height, width):
    # kornia.geometry.conversions.normal_transform_pixel: pixel -> [-1, 1] with the corners on pixel centres
    return torch.tensor([[2.0 / (width - 1), 0.0, -1.0], [0.0, 2.0 / (height - 1), -1.0], [0.0, 0.0, 1.0]])[None]


def dst_norm_to_dst_norm(dst_pix_trans_src_pix, dsize_src, dsize_dst):
    """kornia.geometry.transform.imgwarp.dst_norm_to_dst_norm (call site homographies.py:424): the pixel homography
    expressed between normalised source and destination coordinates."""
    src_norm = _normal_transform_pixel(*dsize_src).to(dst_pix_trans_src_pix)
    dst_norm = _normal_transform_pixel(*dsize_dst).to(dst_pix_trans_src_pix)
    return dst_norm @ (dst_pix_trans_src_pix @ torch.inverse(src_norm))


def kornia_homography_warp(patch_src, src_homo_dst, dsize, mode='bilinear', padding_mode='zeros'):
    """kornia.geometry.warp.homography_warper.homography_warp (call site homographies.py:425): a normalised mesh grid
    of the destination is mapped by the normalised dst -> src homography and sampled with align_corners=True."""
    B = patch_src.shape[0]
    Ho, Wo = int(dsize[0]), int(dsize[1])
    xs = torch.linspace(-1, 1, Wo, dtype=patch_src.dtype)
    ys = torch.linspace(-1, 1, Ho, dtype=patch_src.dtype)
    gy, gx = torch.meshgrid(ys, xs, indexing='ij')
    pts = torch.stack([gx, gy, torch.ones_like(gx)], -1).reshape(1, -1, 3).expand(B, -1, -1)
    w = torch.bmm(pts, src_homo_dst.to(patch_src.dtype).transpose(1, 2))
    z = w[..., 2:]
    scale = torch.where(z.abs() > 1e-8, 1.0 / z, torch.ones_like(z))      # kornia.convert_points_from_homogeneous
    flow = (w[..., :2] * scale).reshape(B, Ho, Wo, 2)
    return F.grid_sample(patch_src, flow, mode=mode, padding_mode=padding_mode, align_corners=True)


def warp_perspective(src, M, dsize, mode='bilinear', padding_mode='zeros'):
    """homographies.py:404-425 (warp_perspective_tensor)."""
    M_norm = dst_norm_to_dst_norm(M, tuple(src.shape[-2:]), tuple(dsize))
    return kornia_homography_warp(src, torch.inverse(M_norm), dsize, mode, padding_mode)


# ------------------------------------------------------------------------------------------------------------------
# reference code restated
# ------------------------------------------------------------------------------------------------------------------
def gaussian_weights(kernel_size, sigma=None):
    """multipoint/utils/utils.py:124-150."""
    if sigma is None:
        sigma = 0.3 * ((kernel_size - 1) * 0.5 - 1) + 0.8
    c = torch.arange(kernel_size)
    xg = c.repeat(kernel_size).view(kernel_size, kernel_size)
    xy = torch.stack([xg, xg.t()], dim=-1)
    mean, var = (kernel_size - 1) / 2., sigma ** 2.
    k = (1. / (2. * pi * var)) * torch.exp(-torch.sum((xy - mean) ** 2., dim=-1) / (2 * var))
    return (k / torch.sum(k)).view(1, 1, kernel_size, kernel_size)


def smooth(prob, kernel_size):
    """filter(pad(prob)), homographies.py:55-58."""
    r = int((kernel_size - 1) / 2)
    return F.conv2d(F.pad(prob, (r, r, r, r), mode='reflect'), gaussian_weights(kernel_size).to(prob.dtype))


def homographic_adaptation(images, forward_fn, config=None, homographies=None, aggregation=None):
    """homographies.py:38-127 (two spectra: images = [optical, thermal], aggregation 'prod' / 'sum') and :129-189
    (one image stream: images = [image], aggregation None).

    forward_fn(stream_index, image_batch) -> prob (B,1,H,W); homographies: the num-1 (3,3) float64 matrices the
    reference would draw from sample_homography.  Returns (out, count)."""
    cfg = full_config(config)
    if cfg['num'] < 1:
        raise ValueError('num must be larger than 0 for the homographic adaptation')
    if cfg['filter_size'] % 2 == 0 and cfg['filter_size'] != 0:
        raise ValueError('The filter_size must be uneven')
    shape = images[0].shape
    B, _, H, W = shape

    def combined(batches):
        maps = [forward_fn(i, x) for i, x in enumerate(batches)]
        if cfg['filter_size'] > 0:
            maps = [smooth(m, cfg['filter_size']) for m in maps]
        if aggregation is None:
            return maps[0]
        if aggregation == 'prod':
            return maps[0] * maps[1]
        if aggregation == 'sum':
            return maps[0] + maps[1]
        raise ValueError('Unknown aggregation: ' + aggregation)

    count = torch.ones(shape)
    prob = combined(images).clone()
    assert len(homographies) == cfg['num'] - 1
    for h64 in homographies:
        mask = compute_valid_mask((H, W), h64, cfg['erosion_radius'], cfg['mask_border'])
        mask = torch.from_numpy(mask.astype(np.float32))[None, None].repeat(B, 1, 1, 1)
        hom = torch.from_numpy(np.asarray(h64).astype(np.float32))[None].repeat(B, 1, 1)
        warped = [warp_perspective(x, hom, (H, W), 'bilinear', 'reflection') for x in images]
        prob_w = combined(warped)
        count_sample = warp_perspective(mask, torch.inverse(hom), (H, W), 'nearest')
        count = count + count_sample
        prob = prob + warp_perspective(prob_w, torch.inverse(hom), (H, W), 'bilinear') * count_sample
    out = prob / count
    if aggregation == 'prod':
        out = out.sqrt()
    elif aggregation == 'sum':
        out = out * 0.5
    if cfg['min_count'] > 0:
        out[count < cfg['min_count']] = 0.0
    return out, count
