"""Reader for the reference's HDF5 pair files (multipoint/datasets/ImagePairDataset.py:13-252): each HDF5 group
holds 'optical', 'thermal' (and 'thermal_raw') images, an optional second file holds the exported 'keypoints' labels
per group; samples are emitted with the reference's dict schema, including the homographic augmentation that the
prediction configs enable (ground-truth 'homography' + 'valid_mask', :209-227) -- its pixel work runs on the GPU
(datasets/augmentation.py).  Photometric augmentation is training-only and rejected.

h5py is imported lazily; a file name ending in '.npz' selects a flat numpy archive with the same layout
('<sample>/optical', '<sample>/thermal', '<sample>/keypoints'), which needs no h5py."""
import copy
import random

import numpy as np
import torch
from torch.utils.data.dataset import Dataset

from ..utils.utils import dict_update
from . import augmentation


class _NpzStore:
    """'<group>/<dataset>' keys of an .npz archive behind the little of the h5py.File interface used below."""

    def __init__(self, filename):
        self._z = np.load(filename)
        self._groups = {}
        for key in self._z.files:
            g, _, d = key.partition('/')
            self._groups.setdefault(g, {})[d] = key

    def keys(self):
        return self._groups.keys()

    def __getitem__(self, group):
        return {d: self._z[k] for d, k in self._groups[group].items()}

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self._z.close()


def _open_store(filename):
    if str(filename).endswith('.npz'):
        return _NpzStore(filename)
    try:
        import h5py
    except ImportError as e:
        raise ImportError("ImagePairDataset needs h5py to read %s (or an '.npz' archive with the same layout); use "
                          "dataset type 'SyntheticPairs' when neither is available" % filename) from e
    return h5py.File(filename, 'r', swmr=True)


def generate_keypoint_map(keypoints, image_shape):
    """utils.generate_keypoint_map (multipoint/utils/utils.py:58-62)."""
    tmp = np.asarray(keypoints).astype(np.int64)
    keypoint_map = np.zeros(image_shape, dtype=bool)
    keypoint_map[tmp[:, 0], tmp[:, 1]] = True
    return keypoint_map


def check_augmentation_config(config, who):
    if config['augmentation']['photometric']['enable']:
        raise NotImplementedError(who + ': photometric augmentation is a training-time feature outside the accelerated '
                                  'inference path; set augmentation.photometric.enable to false')


def build_sample(optical, thermal, keypoints, config, name):
    """ImagePairDataset.py:138-241: the sample dict from two registered (H,W) images and optional (N,2) labels.
    Draws from `random` / `np.random` in the reference's order."""
    h, w = thermal.shape[:2]
    hcfg = config['augmentation']['homographic']
    out = {}

    def entry(image, valid_mask, is_optical, kp):
        if torch.is_tensor(image):                      # warped on the GPU
            image, valid_mask = image.cpu()[None], valid_mask.cpu()[None]
        else:
            image = torch.from_numpy(np.expand_dims(image, 0).astype(np.float32))
            valid_mask = torch.from_numpy(np.expand_dims(valid_mask, 0).astype(bool))
        e = {'image': image.to(torch.float32), 'valid_mask': valid_mask.to(torch.bool),
             'is_optical': torch.BoolTensor([is_optical])}
        if kp is not None:
            e['keypoints'] = torch.from_numpy(generate_keypoint_map(kp, (h, w)))
        return e

    if config['single_image']:
        is_optical = bool(random.randint(0, 1))
        image = optical if is_optical else thermal
        if hcfg['enable']:
            image, keypoints, valid_mask = augmentation.homographic_augmentation(_gpu_image(image), keypoints, **hcfg)
        else:
            valid_mask = augmentation.dummy_valid_mask(image.shape)
        out.update(entry(image, valid_mask, is_optical, keypoints))
    else:
        optical_is_optical, thermal_is_optical = True, False
        if config['random_pairs']:
            tmp_optical, tmp_thermal = optical, thermal
            if bool(random.randint(0, 1)):
                optical, optical_is_optical = tmp_thermal, False
            if bool(random.randint(0, 1)):
                thermal, thermal_is_optical = tmp_optical, True
        hom_optical = hom_thermal = None
        if hcfg['enable']:
            # randomly pick one image to warp (:209-227)
            if bool(random.randint(0, 1)):
                valid_mask_thermal, keypoints_thermal = augmentation.dummy_valid_mask(thermal.shape), keypoints
                optical, keypoints_optical, valid_mask_optical, H = augmentation.homographic_augmentation(
                    _gpu_image(optical), keypoints, return_homography=True, **hcfg)
                hom_optical, hom_thermal = torch.from_numpy(H.astype(np.float32)), torch.eye(3, dtype=torch.float32)
            else:
                valid_mask_optical, keypoints_optical = augmentation.dummy_valid_mask(optical.shape), keypoints
                thermal, keypoints_thermal, valid_mask_thermal, H = augmentation.homographic_augmentation(
                    _gpu_image(thermal), keypoints, return_homography=True, **hcfg)
                hom_thermal, hom_optical = torch.from_numpy(H.astype(np.float32)), torch.eye(3, dtype=torch.float32)
        else:
            keypoints_optical = keypoints_thermal = keypoints
            valid_mask_optical = valid_mask_thermal = augmentation.dummy_valid_mask(optical.shape)
        out['optical'] = entry(optical, valid_mask_optical, optical_is_optical, keypoints_optical)
        out['thermal'] = entry(thermal, valid_mask_thermal, thermal_is_optical, keypoints_thermal)
        if hom_optical is not None:
            out['optical']['homography'], out['thermal']['homography'] = hom_optical, hom_thermal
    if config['return_name']:
        out['name'] = name
    return out


def _gpu_image(image):
    return torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32)).cuda()


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
        check_augmentation_config(self.config, 'ImagePairDataset')
        if self.config['single_image'] and self.config['random_pairs']:
            print('INFO: random_pairs has no influence if single_image is true')
        with _open_store(self.config['filename']) as f:
            self.memberslist = list(f.keys())
        self.num_files = len(self.memberslist)
        if self.config['keypoints_filename'] is not None:
            with _open_store(self.config['keypoints_filename']) as kf:
                have = set(kf.keys())
            missing_labels = [m for m in self.memberslist if m not in have]
            if len(missing_labels) > 0:
                raise IndexError('Labels for the following samples not available: {}'.format(missing_labels))
        print('The dataset ' + self.config['filename'] + ' contains {} samples'.format(self.num_files))

    def __getitem__(self, index):
        with _open_store(self.config['filename']) as f:
            sample = f[self.memberslist[index]]
            optical = sample['optical'][...]
            thermal = sample['thermal_raw'][...] if self.config['raw_thermal'] else sample['thermal'][...]
        if thermal.shape != optical.shape:
            raise ValueError('ImagePairDataset: The optical and thermal image must have the same shape')
        keypoints = None
        if self.config['keypoints_filename'] is not None:
            with _open_store(self.config['keypoints_filename']) as kf:
                keypoints = np.array(kf[self.memberslist[index]]['keypoints'])
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
        return build_sample(optical, thermal, keypoints, self.config, self.memberslist[index])

    def get_name(self, index):
        return self.memberslist[index]

    def returns_pair(self):
        return not self.config['single_image']

    def __len__(self):
        return self.num_files
