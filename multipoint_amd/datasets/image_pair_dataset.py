"""Reader for the reference's HDF5 pair files (multipoint/datasets/ImagePairDataset.py:13-252),
prediction subset: each HDF5 group holds 'optical', 'thermal' (and 'thermal_raw') images; samples are
emitted with the reference's dict schema.  Training-time augmentation is out of scope (SURVEY.md
section 2, rows 13-14) and rejected explicitly.  h5py is imported lazily."""
import copy
import random

import numpy as np
import torch
from torch.utils.data.dataset import Dataset

from ..utils.utils import dict_update


class ImagePairDataset(Dataset):
    default_config = {
        'filename': None,
        'keypoints_filename': None,
        'height': -1,
        'width': -1,
        'raw_thermal': False,
        'single_image': True,
        'random_pairs': False,
        'return_name': True,
        'augmentation': {
            'photometric': {'enable': False, 'primitives': 'all', 'params': {}, 'random_order': True},
            'homographic': {'enable': False, 'params': {}, 'border_reflect': True,
                            'valid_border_margin': 0, 'mask_border': True},
        }
    }

    def __init__(self, config):
        self.config = dict_update(copy.deepcopy(self.default_config), config or {})
        if self.config['filename'] is None:
            raise ValueError('ImagePairDataset: The dataset filename needs to be present in the config file')
        aug = self.config['augmentation']
        if aug['photometric']['enable'] or aug['homographic']['enable']:
            raise NotImplementedError('ImagePairDataset: augmentation is a training-time feature outside the '
                                      'accelerated inference path; set augmentation.*.enable to false')
        try:
            import h5py
        except ImportError as e:
            raise ImportError('ImagePairDataset needs h5py to read %s; use dataset type '
                              "'SyntheticPairs' when h5py is unavailable" % self.config['filename']) from e
        self._h5py = h5py
        with h5py.File(self.config['filename'], 'r') as f:
            self.memberslist = list(f.keys())
        self.num_files = len(self.memberslist)
        print('The dataset ' + self.config['filename'] + ' contains {} samples'.format(self.num_files))

    def __getitem__(self, index):
        with self._h5py.File(self.config['filename'], 'r', swmr=True) as f:
            sample = f[self.memberslist[index]]
            optical = sample['optical'][...]
            thermal = sample['thermal_raw'][...] if self.config['raw_thermal'] else sample['thermal'][...]
        if thermal.shape != optical.shape:
            raise ValueError('ImagePairDataset: The optical and thermal image must have the same shape')
        keypoints = None
        if self.config['keypoints_filename'] is not None:
            with self._h5py.File(self.config['keypoints_filename'], 'r', swmr=True) as kf:
                keypoints = np.array(kf[self.memberslist[index]]['keypoints'])
        h, w = thermal.shape[:2]
        if self.config['height'] > 0 or self.config['width'] > 0:
            h = self.config['height'] if self.config['height'] > 0 else thermal.shape[0]
            w = self.config['width'] if self.config['width'] > 0 else thermal.shape[1]
            if w > thermal.shape[1] or h > thermal.shape[0]:
                raise ValueError('ImagePairDataset: Requested height/width exceeds original image size')
            i_h = random.randint(0, thermal.shape[0] - h)
            i_w = random.randint(0, thermal.shape[1] - w)
            optical = optical[i_h:i_h + h, i_w:i_w + w]
            thermal = thermal[i_h:i_h + h, i_w:i_w + w]
            if keypoints is not None:
                keypoints = keypoints - np.array([[i_h, i_w]])
                keypoints = keypoints[(keypoints[:, 0] >= 0) & (keypoints[:, 0] < h) &
                                      (keypoints[:, 1] >= 0) & (keypoints[:, 1] < w)]

        def entry(img, is_optical):
            e = {'image': torch.from_numpy(np.expand_dims(img, 0).astype(np.float32)),
                 'valid_mask': torch.ones((1, h, w), dtype=torch.bool),
                 'is_optical': torch.BoolTensor([is_optical])}
            if keypoints is not None:
                km = np.zeros((h, w), dtype=bool)
                kk = keypoints.astype(np.int64)
                km[kk[:, 0], kk[:, 1]] = True
                e['keypoints'] = torch.from_numpy(km)
            return e

        out = {}
        if self.config['single_image']:
            is_optical = bool(random.randint(0, 1))
            out.update(entry(optical if is_optical else thermal, is_optical))
        else:
            o_opt, t_opt = True, False
            if self.config['random_pairs']:
                src_o, src_t = optical, thermal
                if bool(random.randint(0, 1)):
                    optical, o_opt = src_t, False
                if bool(random.randint(0, 1)):
                    thermal, t_opt = src_o, True
            out['optical'] = entry(optical, o_opt)
            out['thermal'] = entry(thermal, t_opt)
        if self.config['return_name']:
            out['name'] = self.memberslist[index]
        return out

    def get_name(self, index):
        return self.memberslist[index]

    def returns_pair(self):
        return not self.config['single_image']

    def __len__(self):
        return self.num_files
