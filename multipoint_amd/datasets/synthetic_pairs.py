"""Synthetic optical/thermal pairs with the sample-dict schema of the reference's ImagePairDataset
(multipoint/datasets/ImagePairDataset.py:173-241) -- the 36 GB HDF5 dataset (README.md:30) and h5py
are not available offline, so benchmarks and the CLIs can run on this dataset type instead."""
import copy

import numpy as np
import torch
from torch.utils.data.dataset import Dataset

from ..utils.utils import dict_update
from .image_pair_dataset import build_sample, check_augmentation_config


class SyntheticPairs(Dataset):
    default_config = {
        'num_samples': 32,
        'height': 480,
        'width': 640,
        'seed': 0,
        'single_image': False,
        'random_pairs': False,
        'return_name': True,
        # same block as ImagePairDataset (ImagePairDataset.py:26-39); homographic.enable gives every pair a
        # ground-truth homography + valid mask the way the reference's prediction config does
        'augmentation': {
            'photometric': {'enable': False, 'primitives': 'all', 'params': {}, 'random_order': True},
            'homographic': {'enable': False, 'params': {}, 'border_reflect': True,
                            'valid_border_margin': 0, 'mask_border': True},
        },
    }

    def __init__(self, config=None):
        self.config = dict_update(copy.deepcopy(self.default_config), config or {})
        if self.config['height'] % 8 or self.config['width'] % 8:
            raise ValueError('SyntheticPairs: height and width must be divisible by 8')
        check_augmentation_config(self.config, 'SyntheticPairs')

    @staticmethod
    def make_pair(seed, index, H, W):
        """grayscale fp32 images uniform[0,1) from numpy default_rng (reproducible on any box)."""
        rng = np.random.default_rng([int(seed), int(index)])
        optical = rng.random((1, H, W), dtype=np.float32)
        thermal = rng.random((1, H, W), dtype=np.float32)
        return optical, thermal

    def __getitem__(self, index):
        if index < 0 or index >= len(self):
            raise IndexError(index)
        H, W = self.config['height'], self.config['width']
        optical, thermal = self.make_pair(self.config['seed'], index, H, W)
        if self.config['augmentation']['homographic']['enable'] or self.config['random_pairs']:
            return build_sample(optical[0], thermal[0], None, self.config, self.get_name(index))
        ones = torch.ones((1, H, W), dtype=torch.bool)
        if self.config['single_image']:
            out = {'image': torch.from_numpy(optical), 'valid_mask': ones,
                   'is_optical': torch.BoolTensor([True])}
        else:
            out = {'optical': {'image': torch.from_numpy(optical), 'valid_mask': ones,
                               'is_optical': torch.BoolTensor([True])},
                   'thermal': {'image': torch.from_numpy(thermal), 'valid_mask': ones.clone(),
                               'is_optical': torch.BoolTensor([False])}}
        if self.config['return_name']:
            out['name'] = self.get_name(index)
        return out

    def get_name(self, index):
        return 'synthetic_%06d' % index

    def returns_pair(self):
        return not self.config['single_image']

    def __len__(self):
        return self.config['num_samples']
