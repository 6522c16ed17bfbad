#!/usr/bin/env python3
"""Generates tests/golden/detector_metrics.npz and tests/golden/dataset_augmentation.npz by running the IMPORTED
reference on PyTorch-CPU.  Build container only:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eval.py

detector_metrics.npz   multipoint/utils/evaluation.py:56-97 `compute_tp_fp_dist` (pure torch/numpy, runs unmodified) on
                       seeded heat maps / label maps with distinct scores, and the precision / recall / prob / dist
                       of the `compute_detector_metrics` tail (:33-54) over them.
dataset_augmentation.npz   multipoint/datasets/ImagePairDataset.py `__getitem__` (pair and single-image mode) with the
                       homographic augmentation block of configs/config_image_pair_dataset_prediction.yaml.  h5py is
                       not installed: an in-memory stand-in serves the seeded arrays (the file format is not what is
                       pinned here); cv2 is not installed: warpPerspective / getPerspectiveTransform / erode /
                       perspectiveTransform are the oracle's restatements (oracle/ha_oracle.py) -- so the fixture
                       pins the reference's OWN dataset + augmentation driver code (draw order of random / np.random,
                       which image is warped, keypoint warping and filtering, label maps, dict schema), not OpenCV's
                       pixel arithmetic.
The fixtures hold data only (inputs, seeds, outputs)."""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from oracle import ha_oracle as HA  # noqa: E402
from oracle import mp_oracle as O  # noqa: E402

HCFG = {'enable': True,
        'params': {'translation': True, 'rotation': True, 'scaling': True, 'perspective': True,
                   'scaling_amplitude': 0.2, 'perspective_amplitude_x': 0.2, 'perspective_amplitude_y': 0.2,
                   'patch_ratio': 0.85, 'max_angle': 1.57, 'allow_artifacts': True, 'translation_overflow': 0.05},
        'valid_border_margin': 0, 'border_reflect': True}


def detector_cases():
    rng = np.random.default_rng(17)
    cases = []
    for n, g in ((80, 50), (300, 200), (6, 3), (500, 12)):
        H, W = 40, 56
        prob = np.zeros((H, W), np.float32)
        idx = rng.choice(H * W, n, replace=False)
        prob.flat[idx] = rng.permutation(n).astype(np.float32) / n * 0.9 + 0.01
        km = np.zeros((H, W), bool)
        near = rng.choice(idx, min(g // 2, n), replace=False)
        yy, xx = np.unravel_index(near, (H, W))
        yy = np.clip(yy + rng.integers(-2, 3, len(near)), 0, H - 1); xx = np.clip(xx + rng.integers(-2, 3, len(near)), 0, W - 1)
        km[yy, xx] = True
        km.flat[rng.choice(H * W, g - len(near), replace=False)] = True
        cases.append((prob, km))
    return cases


def make_detector(ev):
    out = {}
    tp, fp, prob, n_gt, dist = [], [], [], 0, []
    for i, (p, km) in enumerate(detector_cases()):
        r = ev.compute_tp_fp_dist(torch.from_numpy(p), torch.from_numpy(km))
        out['prob_%d' % i], out['keypoints_%d' % i] = p, km
        out['tp_%d' % i], out['sorted_prob_%d' % i] = np.asarray(r[0]), np.asarray(r[2])
        out['n_gt_%d' % i], out['dist_%d' % i] = np.int64(r[3]), np.asarray(r[4])
        tp.append(r[0].tolist()); fp.append(r[1].tolist()); prob.append(r[2].tolist()); n_gt += r[3]; dist.append(r[4].tolist())
    # tail of compute_detector_metrics (:33-54) through the reference's own helpers
    tp, fp, prob, dist = np.concatenate(tp), np.concatenate(fp), np.concatenate(prob), np.concatenate(dist)
    sort_idx = np.argsort(prob)[::-1]
    tp, fp, prob = tp[sort_idx], fp[sort_idx], prob[sort_idx]
    tp_cum, fp_cum = np.cumsum(tp), np.cumsum(fp)
    recall = ev.div0(tp_cum, n_gt)
    precision = ev.div0(tp_cum, tp_cum + fp_cum)
    recall = np.concatenate([[0], recall, [1]])
    precision = np.concatenate([[0], precision, [0]])
    precision = np.maximum.accumulate(precision[::-1])[::-1]
    out.update(precision=precision, recall=recall, all_prob=prob, all_dist=dist, mAP=np.float64(ev.compute_mAP(precision, recall)))
    return out


class _FakeH5:
    """h5py.File stand-in over a dict {group: {dataset: array}} (only what ImagePairDataset.py touches)."""
    store = {}

    def __init__(self, filename, mode='r', swmr=False):
        self._g = _FakeH5.store[filename]

    def keys(self):
        return self._g.keys()

    def __getitem__(self, k):
        return self._g[k]

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def make_dataset():
    ref_shim.install_homographies()
    import cv2
    import h5py
    cv2.BORDER_REFLECT_101, cv2.BORDER_CONSTANT, cv2.INTER_LINEAR = 4, 0, 1

    def warpPerspective(src, M, dsize, flags=1, borderMode=0):
        if flags == cv2.INTER_NEAREST:
            return HA.cv2_warp_perspective_nearest(src, M, dsize)
        return HA.cv2_warp_perspective_linear(src, M, dsize, 'reflect101' if borderMode == 4 else 'constant')
    cv2.warpPerspective = warpPerspective
    cv2.perspectiveTransform = lambda pts, h: O.warp_keypoints(pts[0][:, ::-1], h)[None, :, ::-1]
    h5py.File = _FakeH5
    from multipoint.datasets.ImagePairDataset import ImagePairDataset
    H, W = 48, 64
    rng = np.random.default_rng(23)
    images, labels = {}, {}
    for i in range(3):
        images['s%d' % i] = {'optical': rng.random((H, W), dtype=np.float32), 'thermal': rng.random((H, W), dtype=np.float32)}
        labels['s%d' % i] = {'keypoints': np.stack([rng.integers(0, H, 30), rng.integers(0, W, 30)], axis=1)}
    _FakeH5.store = {'images': images, 'labels': labels}
    out = {}
    for i in range(3):
        out['in_optical_%d' % i], out['in_thermal_%d' % i] = images['s%d' % i]['optical'], images['s%d' % i]['thermal']
        out['in_keypoints_%d' % i] = labels['s%d' % i]['keypoints']
    pair = ImagePairDataset({'filename': 'images', 'keypoints_filename': 'labels', 'single_image': False,
                             'augmentation': {'homographic': HCFG}})
    for i in range(3):
        random.seed(302 + i); np.random.seed(400 + i)
        s = pair[i]
        for side in ('optical', 'thermal'):
            for k in ('image', 'valid_mask', 'keypoints', 'homography', 'is_optical'):
                out['pair_%d_%s_%s' % (i, side, k)] = s[side][k].numpy()
        assert s['name'] == 's%d' % i
    single = ImagePairDataset({'filename': 'images', 'keypoints_filename': 'labels', 'single_image': True,
                               'augmentation': {'homographic': HCFG}})
    for i in range(3):
        random.seed(502 + i); np.random.seed(600 + i)
        s = single[i]
        for k in ('image', 'valid_mask', 'keypoints', 'is_optical'):
            out['single_%d_%s' % (i, k)] = s[k].numpy()
        assert 'homography' not in s
    return out


if __name__ == '__main__':
    models, utils = ref_shim.install()
    import multipoint.utils.evaluation as ev
    np.savez_compressed(os.path.join(HERE, 'detector_metrics.npz'), **make_detector(ev))
    np.savez_compressed(os.path.join(HERE, 'dataset_augmentation.npz'), **make_dataset())
    for f in ('detector_metrics.npz', 'dataset_augmentation.npz'):
        print(f, os.path.getsize(os.path.join(HERE, f)), 'bytes')
