#!/usr/bin/env python3
"""Generates tests/golden/homographic_adaptation.npz by running the IMPORTED reference
(/root/reference/multipoint/utils/homographies.py) on PyTorch-CPU.  Build container only:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ha.py

cv2 and kornia are not installed: ref_shim.install_homographies() supplies the oracle's restatements of the five
third-party calls, everything else that runs is the reference's code.  The fixture holds data only:
  sample_*      sample_homography(image_shape, **kwargs) matrices after np.random.seed(seed), for several kwargs
  ha_*          homographic_adaptation / homographic_adaptation_multispectral outputs at 2x1x64x64 with the
                homographies the reference drew (weights / images come from seeds, oracle.make_weights/make_images)
"""
import copy
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from oracle import mp_oracle as O  # noqa: E402

SAMPLE_CASES = [
    {},
    {'allow_artifacts': False, 'max_angle': 0.5, 'patch_ratio': 0.7},
    {'translation': True, 'rotation': True, 'scaling': True, 'perspective': True, 'scaling_amplitude': 0.2,
     'perspective_amplitude_x': 0.2, 'perspective_amplitude_y': 0.2, 'patch_ratio': 0.85, 'max_angle': 1.57,
     'allow_artifacts': True},                                   # configs/config_export_keypoints.yaml:26-36
    {'rotation': False, 'perspective': False},
]
HA_CASES = {
    # name: (multispectral, image seed, np.random seed, homographic_adaptation config)
    'single': (False, 31, 7, {'num': 4, 'erosion_radius': 3, 'mask_border': True, 'min_count': 2, 'filter_size': 0}),
    'single_filter': (False, 32, 8, {'num': 3, 'erosion_radius': 2, 'mask_border': False, 'min_count': 0,
                                     'filter_size': 3}),
    'pair_prod': (True, 33, 9, {'num': 4, 'aggregation': 'prod', 'erosion_radius': 3, 'mask_border': True,
                                'min_count': 2, 'filter_size': 0}),
    'pair_sum': (True, 34, 10, {'num': 3, 'aggregation': 'sum', 'erosion_radius': 5, 'mask_border': True,
                                'min_count': 3, 'filter_size': 5}),
}
HOMOGRAPHIES = SAMPLE_CASES[2]
MODEL_CFG_PAIR = {'multispectral': True, 'descriptor_size': 64}
WEIGHT_SEED = 0


def main():
    RH = ref_shim.install_homographies()
    models, utils = ref_shim.install()
    pristine = copy.deepcopy(RH.homography_adaptation_default_config)
    out = {'sample_cases': json.dumps(SAMPLE_CASES), 'ha_cases': json.dumps(HA_CASES),
           'ha_homographies_cfg': json.dumps(HOMOGRAPHIES), 'model_cfg_pair': json.dumps(MODEL_CFG_PAIR),
           'weight_seed': WEIGHT_SEED, 'sample_shape': np.array([240, 320])}
    for i, kw in enumerate(SAMPLE_CASES):
        mats = []
        for seed in range(6):
            np.random.seed(seed)
            mats.append(RH.sample_homography(np.array([240, 320]), **kw))
        out['sample_%d' % i] = np.stack(mats)
    for name, (pair, img_seed, rng_seed, hc) in HA_CASES.items():
        cfg = dict(MODEL_CFG_PAIR) if pair else dict(O.SHIPPED_MODEL_CONFIG)
        sd = O.make_weights(WEIGHT_SEED, cfg)
        net = models.MultiPoint(dict(cfg)).eval()
        net.load_state_dict(sd)
        img = O.make_images(img_seed, 4 if pair else 2, 64, 64)
        hc = dict(hc, homographies=dict(HOMOGRAPHIES))
        # the reference's dict_update writes into its module-level default: restore it for every case
        RH.homography_adaptation_default_config.clear()
        RH.homography_adaptation_default_config.update(copy.deepcopy(pristine))
        np.random.seed(rng_seed)
        with torch.no_grad():
            if pair:
                data = {'optical': {'image': img[:2], 'is_optical': torch.ones(2, 1, dtype=torch.bool)},
                        'thermal': {'image': img[2:], 'is_optical': torch.zeros(2, 1, dtype=torch.bool)}}
                res = RH.homographic_adaptation_multispectral(data, net, hc)
            else:
                res = RH.homographic_adaptation({'image': img}, net, hc)
        np.random.seed(rng_seed)
        homs = np.stack([RH.sample_homography(np.array([64, 64]), **HOMOGRAPHIES) for _ in range(hc['num'] - 1)])
        out['ha_%s_out' % name] = res.numpy()
        out['ha_%s_homographies' % name] = homs
    path = os.path.join(HERE, 'homographic_adaptation.npz')
    np.savez_compressed(path, **out)
    print('%s  %.1f KB' % (path, os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
