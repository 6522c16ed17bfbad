#!/usr/bin/env python3
"""Drop-in for the reference's export_keypoints.py (flags -y -o -m -v -snms -skip): keypoint labels of a dataset by
homographic adaptation of a base detector (reference export_keypoints.py:64-103).  Every view of every sample is
warped, predicted, warped back and aggregated on the GPU (multipoint_amd.utils.homographic_adaptation*), followed by
box-NMS / top-k and the row-major keypoint list; the output file has the reference's layout, one group per sample
name holding an int64 (N,2) 'keypoints' dataset of (y,x) rows (:100-103).

h5py is needed for the HDF5 output; when it is not installed an output name ending in '.npz' selects a numpy archive
{sample name: (N,2) int64} instead (extension).  -s seeds numpy's generator, which draws the homographies (extension;
the reference leaves it unseeded)."""
import argparse
import os

import numpy as np
import torch
import yaml

import multipoint_amd.datasets as datasets
import multipoint_amd.utils as utils
from predict_align_image_pair import load_network, select_device


def build_parser():
    parser = argparse.ArgumentParser(description='Script to export the keypoints for images in a dataset using a base detector')
    parser.add_argument('-y', '--yaml-config', default='configs/config_export_keypoints.yaml', help='YAML config file')
    parser.add_argument('-o', '--output_file', required=True, help='Output file name')
    parser.add_argument('-m', '--model-dir', default='model_weights/multipoint', help='Directory of the model')
    parser.add_argument('-v', '--version', default='none', help='Model version (name of the .model file)')
    parser.add_argument('-snms', '--single-nms', action='store_true', help='Do the nms calculation for each sample separately')
    parser.add_argument('-skip', dest='skip_processed', action='store_true', help='Skip already processed samples')
    parser.add_argument('-s', '--seed', default=None, type=int, help='Seed of the homography sampler (numpy)')
    return parser


class _NpzStore:
    """'<sample>/keypoints' arrays in a numpy archive: the group / dataset layout of the HDF5 label file, readable by
    multipoint_amd.datasets.ImagePairDataset(keypoints_filename=...); used when h5py is unavailable."""

    def __init__(self, path):
        self.path = path
        self.data = {}
        if os.path.exists(path):
            with np.load(path) as f:
                self.data = {k: f[k] for k in f.files}

    def keys(self):
        return [k.rpartition('/')[0] for k in self.data if k.endswith('/keypoints')]

    def put(self, name, keypoints):
        self.data[name + '/keypoints'] = keypoints

    def close(self):
        np.savez_compressed(self.path, **self.data)


class _Hdf5Store:
    def __init__(self, path, h5py):
        self.f = h5py.File(path, 'a')

    def keys(self):
        return self.f.keys()

    def put(self, name, keypoints):
        self.f.create_group(name)
        self.f[name].create_dataset('keypoints', data=keypoints)

    def close(self):
        self.f.close()


def open_store(path):
    try:
        import h5py
    except ImportError:
        if path.endswith('.npz'):
            return _NpzStore(path)
        raise ImportError("export_keypoints: h5py is needed to write %s; give an output name ending in '.npz' for a "
                          'numpy archive instead' % path)
    return _NpzStore(path) if path.endswith('.npz') else _Hdf5Store(path, h5py)


def main(argv=None):
    args = build_parser().parse_args(argv)
    with open(args.yaml_config, 'r') as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    with open(os.path.join(args.model_dir, 'params.yaml'), 'r') as f:
        config['model'] = yaml.load(f, Loader=yaml.FullLoader)['model']      # overwrite the model params
    if args.seed is not None:
        np.random.seed(args.seed)

    output = open_store(args.output_file)
    device = select_device(config)
    print('Predicting on device: {}'.format(device))
    dataset = getattr(datasets, config['dataset']['type'])(config['dataset'])
    loader_dataset = torch.utils.data.DataLoader(dataset, batch_size=config['prediction']['batchsize'],
                                                 shuffle=False, num_workers=datasets.loader_num_workers(dataset, config['prediction']['num_worker']))
    net = load_network(config, args.model_dir, args.version, device, 0)
    pred = config['prediction']
    n_samples = n_keypoints = 0

    with torch.no_grad():
        for batch in loader_dataset:
            names = list(batch['name'])
            if args.skip_processed and all(name in output.keys() for name in names):
                continue
            batch = utils.data_to_device(batch, device)
            if dataset.returns_pair():
                prob_ha = utils.homographic_adaptation_multispectral(batch, net, pred['homographic_adaptation'])
            else:
                prob_ha = utils.homographic_adaptation(batch, net, pred['homographic_adaptation'])
            # box_nms + torch.nonzero(prob > detection_threshold) (reference :82-100), fused; images are independent in
            # the suppression, so -snms gives the same lists as the batched call
            if pred['nms'] > 0:
                kp, _, cnt = utils.detect_keypoints(prob_ha, pred['nms'], pred['detection_threshold'],
                                                    keep_top_k=pred['topk'])
            else:
                kp, _, cnt = utils.extract_keypoints(prob_ha, pred['detection_threshold'])
            kp, cnt = kp.cpu().numpy(), cnt.cpu().numpy()
            for i, name in enumerate(names):
                if args.skip_processed and name in output.keys():
                    continue
                output.put(name, kp[i, :cnt[i]].astype(np.int64))
                n_samples += 1
                n_keypoints += int(cnt[i])
    output.close()
    print('Exported {} keypoints of {} samples to {}'.format(n_keypoints, n_samples, args.output_file))
    return n_samples, n_keypoints


if __name__ == '__main__':
    main()
