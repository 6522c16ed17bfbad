"""Homographic augmentation of the dataset samples (multipoint/datasets/augmentation/augmentation.py:25-61), the step
that gives the prediction / evaluation configs their ground-truth homography and valid mask
(configs/config_image_pair_dataset_prediction.yaml: augmentation.homographic.enable = true).

The homography is sampled on the host exactly as the reference does (`utils.sample_homography`, numpy's global
generator); the pixel work runs in HIP behind the C ABI: `mp_warp_perspective_cv` restates
cv2.warpPerspective(INTER_LINEAR, BORDER_REFLECT_101 | BORDER_CONSTANT), `mp_ha_valid_mask` restates
compute_valid_mask.  No CPU fallback.  Photometric augmentation is a training-time feature and is rejected."""
import numpy as np
import torch

from .. import _lib
from ..utils import homographies as hom

__all__ = ['homographic_augmentation', 'homographic_augmentation_batch', 'warp_perspective_cv', 'dummy_valid_mask', 'cv_invert3',
           'photometric_augmentation']


def cv_invert3(m):
    """cv::invert of a 3x3 float64 matrix (closed-form adjugate / determinant, what cv2.warpPerspective applies to
    M before it maps destination to source pixels).  A singular matrix gives zeros, as OpenCV does."""
    s = np.asarray(m, dtype=np.float64).reshape(3, 3)
    c00 = s[1, 1] * s[2, 2] - s[1, 2] * s[2, 1]
    c01 = s[1, 0] * s[2, 2] - s[1, 2] * s[2, 0]
    c02 = s[1, 0] * s[2, 1] - s[1, 1] * s[2, 0]
    det = s[0, 0] * c00 - s[0, 1] * c01 + s[0, 2] * c02
    if det == 0.0:
        return np.zeros((3, 3))
    d = 1.0 / det
    return np.array([[c00 * d, (s[0, 2] * s[2, 1] - s[0, 1] * s[2, 2]) * d, (s[0, 1] * s[1, 2] - s[0, 2] * s[1, 1]) * d],
                     [(s[1, 2] * s[2, 0] - s[1, 0] * s[2, 2]) * d, (s[0, 0] * s[2, 2] - s[0, 2] * s[2, 0]) * d,
                      (s[0, 2] * s[1, 0] - s[0, 0] * s[1, 2]) * d],
                     [c02 * d, (s[0, 1] * s[2, 0] - s[0, 0] * s[2, 1]) * d, (s[0, 0] * s[1, 1] - s[0, 1] * s[1, 0]) * d]])


def warp_perspective_cv(images, homographies, border_reflect=False):
    """cv2.warpPerspective(image, H, (W, H), borderMode=BORDER_CONSTANT | BORDER_REFLECT_101) with INTER_LINEAR for a
    batch of fp32 images (B,1,H,W) on the GPU (mp_warp_perspective_cv); homographies (B,3,3) map source to destination
    pixels.  Used by the dataset augmentation below and by predict_align_image_pair.py to warp the optical image onto
    the thermal one with the estimated homography (reference predict_align_image_pair.py:218)."""
    if not torch.is_tensor(images) or images.dim() != 4 or images.shape[1] != 1:
        raise ValueError('warp_perspective_cv: images must be a (B,1,H,W) tensor')
    dev = _lib.require_cuda(images.device if images.device.type == 'cuda' else None)
    src = images.to(dev, torch.float32).contiguous()
    B, _, H, W = src.shape
    h_np = np.asarray(homographies, dtype=np.float64).reshape(-1, 3, 3)
    if h_np.shape[0] != B:
        raise ValueError('warp_perspective_cv: one homography per image expected')
    inv_cv = torch.from_numpy(np.stack([cv_invert3(m) for m in h_np]).reshape(B, 9)).to(dev)
    out = torch.empty_like(src)
    h = _lib.get_handle(dev)
    with torch.cuda.device(dev):
        h.check(h.lib.mp_warp_perspective_cv(h.ptr, _lib.ptr(src), B, H, W, _lib.ptr(inv_cv),
                                             1 if border_reflect else 0, _lib.ptr(out), _lib.stream_ptr(dev)))
    return out


def homographic_augmentation_batch(images, homographies, border_reflect=True, valid_border_margin=0,
                                   mask_border=True):
    """images (B,1,H,W) fp32 on the GPU, homographies (B,3,3) source -> destination pixel maps (host).
    Returns the warped images (B,1,H,W) fp32 and the valid masks (B,1,H,W) bool, both on the GPU."""
    out = warp_perspective_cv(images, homographies, border_reflect)
    dev = out.device
    B, _, H, W = out.shape
    h_np = np.asarray(homographies, dtype=np.float64).reshape(-1, 3, 3)
    # compute_valid_mask(image_shape, homography, valid_border_margin * 2, mask_border)  (augmentation.py:38-40)
    mask = hom._valid_masks(np.linalg.inv(h_np), (H, W), int(valid_border_margin) * 2, mask_border, dev)
    return out, mask.view(B, 1, H, W).bool()


def homographic_augmentation(image, keypoints=None, return_homography=False, **config):
    """augmentation.py:25-54: warp an (H,W) image with a random homography drawn from config['params'], warp and
    filter the (N,2) (y,x) keypoints, compute the valid mask.  numpy in, numpy out, like the reference; a CUDA
    tensor image is accepted too and then image / mask stay on the GPU (warped_image (H,W) fp32, valid_mask (H,W))."""
    on_gpu = torch.is_tensor(image)
    image_shape = tuple(image.shape)
    if len(image_shape) != 2:
        raise ValueError('homographic_augmentation: expected an (H,W) image, got shape {}'.format(image_shape))
    homography = hom.sample_homography(image_shape, **config['params'])
    img = image if on_gpu else torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32))
    dev = _lib.require_cuda(img.device if img.device.type == 'cuda' else None)
    warped, mask = homographic_augmentation_batch(img.to(dev)[None, None], homography[None],
                                                  config['border_reflect'], config['valid_border_margin'],
                                                  config['mask_border'])
    if on_gpu:
        warped_image, valid_mask = warped[0, 0], mask[0, 0]
    else:
        warped_image = warped[0, 0].cpu().numpy()
        valid_mask = mask[0, 0].cpu().numpy().astype(np.float64)
    if keypoints is not None:
        if keypoints.size > 0:
            warped_points = hom.filter_points(hom.warp_keypoints(keypoints, homography), image_shape)
        else:
            warped_points = keypoints
    else:
        warped_points = None
    if return_homography:
        return warped_image, warped_points, valid_mask, homography
    return warped_image, warped_points, valid_mask


def dummy_valid_mask(image_shape):
    """augmentation.py:56-61."""
    return np.ones(image_shape)


def photometric_augmentation(image, **config):
    raise NotImplementedError('photometric augmentation is a training-time feature outside the accelerated '
                              'inference path (SURVEY.md section 2); set augmentation.photometric.enable to false')
