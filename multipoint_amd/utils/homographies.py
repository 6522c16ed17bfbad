"""Host-side mirror of the homographic-adaptation code of the reference (multipoint/utils/homographies.py), the
label-export path that export_keypoints.py:64-103 drives.

Sampling a random homography is host arithmetic on 4 points, as in the reference, and draws from numpy's global
generator in the reference's order (so `np.random.seed(s)` selects the same homographies in both code bases); the
3x3 matrix comes from an 8x8 linear solve in place of cv2.getPerspectiveTransform.  Everything that touches pixels
runs in HIP behind the C ABI: the perspective warp (kornia's warp_perspective in the reference), the valid mask
(cv2.warpPerspective + cv2.erode), the optional Gaussian filter and the aggregation of the warped-back heat maps.

Unlike the reference, which runs `num - 1` forwards of B images one after the other, the driver below stacks the
warped copies of several homographies into one forward of up to `max_images` images: a 288 GB MI355X holds them
easily and the persistent convolution kernels want >= 256 work items per launch.  The sums are formed in the
reference's order, so the result does not depend on the grouping."""
import copy
from math import pi

import numpy as np
import torch

from .. import _lib
from .utils import dict_update

__all__ = ['homography_adaptation_default_config', 'homographic_adaptation', 'homographic_adaptation_multispectral',
           'sample_homography', 'get_perspective_transform', 'compute_valid_mask', 'warp_perspective_tensor',
           'WarpingModule', 'get_gaussian_filter', 'gaussian_filter', 'warp_keypoints', 'warp_points_pytorch',
           'filter_points']

# homographies.py:17-36
homography_adaptation_default_config = {
    'num': 100,
    'aggregation': 'prod',
    'homographies': {
        'translation': True,
        'rotation': True,
        'scaling': True,
        'perspective': True,
        'scaling_amplitude': 0.15,
        'perspective_amplitude_x': 0.15,
        'perspective_amplitude_y': 0.15,
        'patch_ratio': 0.9,
        'max_angle': pi,
        'allow_artifacts': True,
    },
    'erosion_radius': 5,
    'mask_border': True,
    'min_count': 2,
    'filter_size': 0,
}

_AGG = {None: 0, 'prod': 1, 'sum': 2}


# ------------------------------------------------------------------------------------------------------------------
# host arithmetic
# ------------------------------------------------------------------------------------------------------------------
def get_perspective_transform(src, dst):
    """3x3 homography mapping the four (x, y) points `src` onto `dst` with h33 = 1: the linear system that
    cv2.getPerspectiveTransform (homographies.py:326) solves, in float64 on the float32-rounded points."""
    src = np.asarray(src, dtype=np.float32).astype(np.float64)
    dst = np.asarray(dst, dtype=np.float32).astype(np.float64)
    if src.shape != (4, 2) or dst.shape != (4, 2):
        raise ValueError('get_perspective_transform needs two (4,2) point arrays')
    a = np.zeros((8, 8))
    b = np.zeros(8)
    for i in range(4):
        x, y = src[i]
        u, v = dst[i]
        a[i] = [x, y, 1, 0, 0, 0, -x * u, -y * u]
        a[i + 4] = [0, 0, 0, x, y, 1, -x * v, -y * v]
        b[i], b[i + 4] = u, v
    return np.append(np.linalg.solve(a, b), 1.0).reshape(3, 3)


def sample_homography(image_shape, perspective=True, scaling=True, rotation=True, translation=True,
                      n_scales=10, n_angles=25, scaling_amplitude=0.2, perspective_amplitude_x=0.1,
                      perspective_amplitude_y=0.1, patch_ratio=0.8, max_angle=pi / 2,
                      allow_artifacts=True, translation_overflow=0.1):
    """Random homography of the reference (homographies.py:191-327): a centred patch of relative size `patch_ratio`
    is perturbed by the enabled transformations in random order and the homography maps the image corners onto it.
    Same arguments, same draws from np.random in the same order."""
    rng = np.random
    unit = np.array([[0., 0.], [0., 1.], [1., 1.], [1., 0.]])
    patch = (1 - patch_ratio) * 0.5 + patch_ratio * unit

    def perspective_step(p):
        lo, hi = -p.min(axis=0), 1.0 - p.max(axis=0)
        hi[1] = min(abs(lo[1]), abs(hi[1]))
        lo[1] = -hi[1]
        amp = np.array([perspective_amplitude_x, perspective_amplitude_y])
        amp_lo, amp_hi = (-amp, amp) if allow_artifacts else (np.maximum(-amp, lo), np.minimum(amp, hi))
        dy = rng.uniform(amp_lo[1], amp_hi[1])
        dx_left = rng.uniform(amp_lo[0], amp_hi[0])
        dx_right = rng.uniform(amp_lo[0], amp_hi[0])
        return p + np.array([[dx_left, dy], [dx_left, -dy], [dx_right, dy], [dx_right, -dy]])

    def inside(q):
        return q.max() < 1.0 and q.min() >= 0.0

    def scale_step(p):
        factors = rng.uniform(-scaling_amplitude, scaling_amplitude, n_scales) + 1.0
        c = p.mean(axis=0)
        cand = (p - c)[None] * factors[:, None, None] + c
        ok = np.arange(n_scales) if allow_artifacts else [i for i in range(n_scales) if inside(cand[i])]
        return cand[rng.choice(ok)]

    def translation_step(p):
        lo, hi = -p.min(axis=0), 1.0 - p.max(axis=0)
        if allow_artifacts:
            lo, hi = lo - translation_overflow, hi + translation_overflow
        tx = rng.uniform(lo[0], hi[0])
        ty = rng.uniform(lo[1], hi[1])
        return p + np.array([tx, ty])

    def rotation_step(p):
        ang = np.append(rng.uniform(-max_angle, max_angle, n_angles), 0.0)   # the trailing 0: "no rotation"
        c = p.mean(axis=0)
        cs, sn = np.cos(ang), np.sin(ang)
        rot = np.stack([cs, -sn, sn, cs], axis=1).reshape(-1, 2, 2)
        cand = np.matmul((p - c)[None].repeat(n_angles + 1, axis=0), rot) + c
        ok = np.arange(n_angles) if allow_artifacts else [i for i in range(n_angles + 1) if inside(cand[i])]
        return cand[rng.choice(ok)]

    steps = [f for on, f in ((perspective, perspective_step), (scaling, scale_step),
                             (translation, translation_step), (rotation, rotation_step)) if on]
    order = np.arange(len(steps))
    rng.shuffle(order)
    for i in order:
        patch = steps[i](patch)

    wh = np.asarray(image_shape)[::-1]          # image_shape is (H, W); points are (x, y)
    return get_perspective_transform(unit * wh, patch * wh)


def warp_keypoints(keypoints, homography, return_type=int):
    """(N,2) (y,x) keypoints through the homography (homographies.py:329-346; cv2.perspectiveTransform in float64)."""
    keypoints = np.asarray(keypoints)
    if len(keypoints) == 0:
        return keypoints
    h = np.asarray(homography, dtype=np.float64)
    xy1 = np.concatenate([keypoints[:, ::-1].astype(np.float64), np.ones((len(keypoints), 1))], axis=1)
    w = xy1 @ h.T
    with np.errstate(divide='ignore', invalid='ignore'):
        xy = np.where(w[:, 2:] != 0, w[:, :2] / w[:, 2:], 0.0)
    return xy[:, ::-1].astype(return_type)


def warp_points_pytorch(points, homography):
    """(B,N,2) (y,x) points through (B,3,3) homographies (homographies.py:348-357)."""
    xy1 = torch.cat([points.flip(-1), torch.ones(points.shape[:2] + (1,), dtype=torch.float32,
                                                 device=points.device)], -1)
    w = torch.bmm(homography, xy1.permute(0, 2, 1)).permute(0, 2, 1)
    return (w[:, :, :2] / w[:, :, 2:]).flip(-1)


def filter_points(points, shape):
    """Drop the (y,x) points outside an image of the given shape (homographies.py:359-374)."""
    keep = (points[:, 0] >= 0) & (points[:, 1] >= 0) & (points[:, 0] < shape[0]) & (points[:, 1] < shape[1])
    return points[keep]


# ------------------------------------------------------------------------------------------------------------------
# GPU pieces
# ------------------------------------------------------------------------------------------------------------------
def _hom_tensor(h, dev):
    """(n,3,3) array / tensor -> contiguous device float64 (n,9)."""
    if torch.is_tensor(h):
        h = h.detach().to('cpu', torch.float64).numpy()
    h = np.ascontiguousarray(np.asarray(h, dtype=np.float64).reshape(-1, 9))
    return torch.from_numpy(h).to(dev)


def _maps(t):
    """(B,1,H,W) tensor -> contiguous fp32 device tensor, device, B, H, W."""
    if not torch.is_tensor(t):
        raise TypeError('Input src type is not a torch.Tensor. Got {}'.format(type(t)))
    if t.dim() != 4:
        raise ValueError('Input src must be a BxCxHxW tensor. Got {}'.format(t.shape))
    if t.shape[1] != 1:
        raise ValueError('the HIP warp handles single-channel maps (B,1,H,W); got {}'.format(tuple(t.shape)))
    dev = _lib.require_cuda(t.device if t.device.type == 'cuda' else None)
    return t.to(dev, torch.float32).contiguous(), dev, t.shape[0], t.shape[2], t.shape[3]


def _warp(src, dst_to_src, n_out, dsize, mode, padding_mode):
    s, dev, B, H, W = _maps(src)
    modes, pads = {'bilinear': 0, 'nearest': 1}, {'zeros': 0, 'reflection': 1}
    if mode not in modes or padding_mode not in pads:
        raise ValueError("warp: mode must be 'bilinear' or 'nearest' and padding_mode 'zeros' or 'reflection'")
    Ho, Wo = int(dsize[0]), int(dsize[1])
    out = torch.empty((n_out, 1, Ho, Wo), dtype=torch.float32, device=dev)
    hom = _hom_tensor(dst_to_src, dev)
    if hom.shape[0] != n_out:
        raise ValueError('warp: need one homography per output map')
    h = _lib.get_handle(dev)
    with torch.cuda.device(dev):
        h.check(h.lib.mp_warp_perspective(h.ptr, _lib.ptr(s), B, H, W, _lib.ptr(hom), n_out, Ho, Wo, modes[mode],
                                          pads[padding_mode], _lib.ptr(out), _lib.stream_ptr(dev)))
    return out


def warp_perspective_tensor(src, M, dsize, mode='bilinear', padding_mode='zeros'):
    """kornia.warp_perspective as the reference calls it (homographies.py:404-425): src (B,1,H,W), M (B,3,3) mapping
    source pixels to destination pixels, output (B,1,*dsize).  The 3x3 inverse is taken on the host in float64."""
    if not torch.is_tensor(M):
        raise TypeError('Input M type is not a torch.Tensor. Got {}'.format(type(M)))
    if not (M.dim() == 3 and M.shape[-2:] == (3, 3)):
        raise ValueError('Input M must be a Bx3x3 tensor. Got {}'.format(M.shape))
    if M.shape[0] != src.shape[0]:
        raise ValueError('warp_perspective_tensor: one homography per image expected')
    inv = np.linalg.inv(M.detach().to('cpu', torch.float64).numpy())
    return _warp(src, inv, src.shape[0], dsize, mode, padding_mode).to(src.device)


class WarpingModule(torch.nn.Module):
    """homographies.py:427-433."""

    def forward(self, src, M, dsize, mode='bilinear', padding_mode='zeros'):
        return warp_perspective_tensor(src, M, dsize, mode, padding_mode)


def _valid_masks(hom_inv, shape, erosion_radius, mask_border, dev):
    H, W = int(shape[0]), int(shape[1])
    hi = _hom_tensor(hom_inv, dev)
    G = hi.shape[0]
    mask = torch.empty((G, H, W), dtype=torch.uint8, device=dev)
    h = _lib.get_handle(dev)
    with torch.cuda.device(dev):
        h.check(h.lib.mp_ha_valid_mask(h.ptr, _lib.ptr(hi), G, H, W, int(erosion_radius), int(bool(mask_border)),
                                       _lib.ptr(mask), _lib.stream_ptr(dev)))
    return mask


def compute_valid_mask(image_shape, homography, erosion_radius=0, mask_border=False, device=None):
    """Pixels of the warped image that come from inside the source frame, eroded by `erosion_radius`
    (homographies.py:361-389).  Returns an (H,W) float64 numpy array of 0/1 like the reference; computed on the GPU."""
    dev = _lib.require_cuda(device)
    inv = np.linalg.inv(np.asarray(homography, dtype=np.float64))
    return _valid_masks(inv[None], image_shape, erosion_radius, mask_border, dev)[0].cpu().numpy().astype(np.float64)


def get_gaussian_filter(kernel_size, sigma=None, channels=1):
    """Weights of the reference's Gaussian smoothing filter (multipoint/utils/utils.py:124-160) as a
    (channels,1,k,k) fp32 tensor: normalised samples of exp(-(dx^2+dy^2)/(2 sigma^2)),
    sigma = 0.3*((k-1)*0.5 - 1) + 0.8 by default."""
    if sigma is None:
        sigma = 0.3 * ((kernel_size - 1) * 0.5 - 1) + 0.8
    ax = torch.arange(kernel_size, dtype=torch.float32) - (kernel_size - 1) / 2.
    d2 = ax[None, :] ** 2 + ax[:, None] ** 2
    var = float(sigma) ** 2
    k = (1. / (2. * pi * var)) * torch.exp(-d2 / (2 * var))
    k = k / k.sum()
    return k.view(1, 1, kernel_size, kernel_size).repeat(channels, 1, 1, 1)


def gaussian_filter(prob, kernel_size, weights=None):
    """filter(pad(prob)) of homographies.py:55-58 on (B,1,H,W) maps: ReflectionPad2d((k-1)/2) + k x k filter."""
    p, dev, B, H, W = _maps(prob)
    w = (get_gaussian_filter(kernel_size) if weights is None else weights).to(dev, torch.float32).contiguous()
    out = torch.empty_like(p)
    h = _lib.get_handle(dev)
    with torch.cuda.device(dev):
        h.check(h.lib.mp_gaussian_filter(h.ptr, _lib.ptr(p), B, H, W, int(kernel_size), _lib.ptr(w), _lib.ptr(out),
                                         _lib.stream_ptr(dev)))
    return out


# ------------------------------------------------------------------------------------------------------------------
# the adaptation drivers
# ------------------------------------------------------------------------------------------------------------------
def _check_config(config):
    if config['num'] < 1:
        raise ValueError('num must be larger than 0 for the homographic adaptation')
    if config['filter_size'] % 2 == 0 and config['filter_size'] != 0:
        raise ValueError('The filter_size must be uneven')


def _adapt(streams, net, config, aggregation, homographies=None, max_images=64):
    """streams: list of data dicts (one per spectrum) with 'image' (B,1,H,W) and optionally 'is_optical'."""
    if aggregation not in _AGG:
        raise ValueError('Unknown aggregation: ' + str(aggregation))
    agg = _AGG[aggregation]
    images = []
    for d in streams:
        img, dev, B, H, W = _maps(d['image'])
        images.append(img)
    if any(i.shape != images[0].shape for i in images):
        raise ValueError('homographic adaptation: the images of both spectra must have the same shape')
    ksize = int(config['filter_size'])
    wgt = get_gaussian_filter(ksize).to(dev) if ksize > 0 else None
    h = _lib.get_handle(dev)
    stream = _lib.stream_ptr(dev)

    def heat_maps(batch_images, g):
        outs = []
        for d, img in zip(streams, batch_images):
            inp = {'image': img}
            if 'is_optical' in d:
                inp['is_optical'] = d['is_optical'].repeat(g, 1)
            p = net(inp)['prob']
            outs.append(gaussian_filter(p, ksize, wgt) if ksize > 0 else p.contiguous())
        return outs

    prob = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
    count = torch.empty_like(prob)
    first = heat_maps(images, 1)
    with torch.cuda.device(dev):
        h.check(h.lib.mp_ha_begin(h.ptr, _lib.ptr(first[0]), _lib.ptr(first[1]) if agg else None, B, H, W, agg,
                                  _lib.ptr(prob), _lib.ptr(count), stream))

    n_h = int(config['num']) - 1
    if homographies is None:
        # the only consumer of np.random in the reference's loop: drawing them up front keeps its sequence
        homographies = [sample_homography(np.array([H, W]), **config['homographies']) for _ in range(n_h)]
    homographies = np.asarray(homographies, dtype=np.float64).reshape(-1, 3, 3)
    if len(homographies) != n_h:
        raise ValueError('homographic adaptation: need num - 1 = %d homographies' % n_h)
    # the valid mask is built from the float64 matrix, the warps from its float32 rounding (homographies.py:79-82)
    homographies32 = homographies.astype(np.float32).astype(np.float64)

    group = max(1, int(max_images) // max(1, B))
    for g0 in range(0, n_h, group):
        hom = homographies32[g0:g0 + group]
        G = len(hom)
        inv = np.linalg.inv(hom)
        hom_d = _hom_tensor(hom, dev)
        # image g*B + b = image b warped by homography g   (warper(image, homography, ..., 'bilinear', 'reflection'))
        per_image = np.repeat(inv, B, axis=0)
        warped = [_warp(img, per_image, G * B, (H, W), 'bilinear', 'reflection') for img in images]
        maps = heat_maps(warped, G)
        mask = _valid_masks(np.linalg.inv(homographies[g0:g0 + group]), (H, W), config['erosion_radius'], config['mask_border'], dev)
        with torch.cuda.device(dev):
            h.check(h.lib.mp_ha_accumulate(h.ptr, _lib.ptr(maps[0]), _lib.ptr(maps[1]) if agg else None, _lib.ptr(mask),
                                           _lib.ptr(hom_d), G, B, H, W, agg, _lib.ptr(prob), _lib.ptr(count), stream))
    out = torch.empty_like(prob)
    with torch.cuda.device(dev):
        h.check(h.lib.mp_ha_finalize(h.ptr, _lib.ptr(prob), _lib.ptr(count), B, H, W, agg, float(config['min_count']),
                                     _lib.ptr(out), stream))
    return out


def homographic_adaptation(data, net, homographic_adaptation_config={}, homographies=None, max_images=64):
    """Heat map averaged over `num` random homographic views of each image (homographies.py:129-189).
    `homographies` (extension): the num-1 matrices to use instead of sampling them; `max_images`: images per forward."""
    config = dict_update(copy.deepcopy(homography_adaptation_default_config), homographic_adaptation_config)
    _check_config(config)
    return _adapt([data], net, config, None, homographies, max_images)


def homographic_adaptation_multispectral(data, net, homographic_adaptation_config={}, homographies=None,
                                         max_images=64):
    """As above for optical/thermal pairs: the two heat maps of every view are multiplied ('prod', geometric mean in
    the end) or added ('sum') before aggregation (homographies.py:38-127)."""
    config = dict_update(copy.deepcopy(homography_adaptation_default_config), homographic_adaptation_config)
    _check_config(config)
    if config['aggregation'] not in ('prod', 'sum'):
        raise ValueError('Unknown aggregation: ' + str(config['aggregation']))
    return _adapt([data['optical'], data['thermal']], net, config, config['aggregation'], homographies, max_images)
