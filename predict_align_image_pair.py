#!/usr/bin/env python3
"""Drop-in for the hot path of the reference's predict_align_image_pair.py: same flags
(-y -m -v -i -r -p -e -tk -th -s), same yaml keys, same model_weights/<name>/{params.yaml,<version>.model}
format, same three timing prints (reference predict_align_image_pair.py:141-143) -- computed on an
MI355X through libmultipoint_hip.so.  -e computes NN-mAP, M-score and homography correctness like the
reference (utils.compute_descriptor_metrics; per-sample arithmetic and a batched RANSAC on the GPU -- the
RANSAC is the product's own, not OpenCV's RNG).  The matplotlib/cv2 visualisation of -p is replaced by a
text summary of keypoints/matches, the estimated and the ground-truth homography (and --save-npz)."""
import argparse
import os
import random
import time

import numpy as np
import torch
import yaml

import multipoint_amd.datasets as datasets
import multipoint_amd.models as models
import multipoint_amd.utils as utils
from multipoint_amd.pipeline import PairPipeline


def synchronize():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def build_parser():
    parser = argparse.ArgumentParser(description='Predict the keypoints of an image')
    parser.add_argument('-y', '--yaml-config', default='configs/config_image_pair_dataset_prediction.yaml', help='YAML config file')
    parser.add_argument('-m', '--model-dir', default='model_weights/multipoint', help='Directory of the model')
    parser.add_argument('-v', '--version', default='latest', help='Model version (name of the param file), none for no weights')
    parser.add_argument('-i', '--index', default=0, type=int, help='Index of the sample to predict and show')
    parser.add_argument('-r', '--radius', default=4, type=int, help='Radius of the keypoint circle')
    parser.add_argument('-p', dest='plot', action='store_true', help='If set the prediction the results are displayed')
    parser.add_argument('-e', dest='evaluation', action='store_true', help='If set the evaluation metrics are computed')
    parser.add_argument('-tk', dest='threshold_keypoints', default=4, type=int, help='Distance below which two keypoints are considered a match')
    parser.add_argument('-th', dest='threshold_homography', default=1, type=int, help='Homography correctness threshold')
    parser.add_argument('-s', '--seed', default=0, type=int, help='Seed of the random generators')
    parser.add_argument('--save-npz', default=None, help='(extension) write keypoints/descriptors/matches of the sample here')
    return parser


def load_network(config, model_dir, version, device, seed=0):
    net = getattr(models, config['model']['type'])(config['model'])
    if version != 'none':
        weights = torch.load(os.path.join(model_dir, version + '.model'), map_location=torch.device('cpu'))
        weights = utils.fix_model_weigth_keys(weights)
        net.load_state_dict(weights)
        del weights
    else:
        net.init_random_weights(seed)
    net.to(device)
    net.eval()
    return net


def select_device(config):
    if not config['prediction']['allow_gpu'] or not torch.cuda.is_available():
        raise RuntimeError('this implementation runs on an MI355X only: prediction.allow_gpu must be true and '
                           'a GPU must be visible (there is no CPU fallback; the reference runs on CPU)')
    return torch.device('cuda:0')


def main(argv=None):
    args = build_parser().parse_args(argv)
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)

    with open(args.yaml_config, 'r') as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    with open(os.path.join(args.model_dir, 'params.yaml'), 'r') as f:
        config['model'] = yaml.load(f, Loader=yaml.FullLoader)['model']      # overwrite the model params

    device = select_device(config)
    print('Predicting on device: {}'.format(device))

    dataset = getattr(datasets, config['dataset']['type'])(config['dataset'])
    loader_dataset = torch.utils.data.DataLoader(dataset, batch_size=config['prediction']['batchsize'],
                                                 shuffle=False, num_workers=datasets.loader_num_workers(dataset, config['prediction']['num_worker']))
    net = load_network(config, args.model_dir, args.version, device, args.seed)
    pred = config['prediction']

    with torch.no_grad():
        if args.evaluation:
            # reference predict_align_image_pair.py:69-88: utils.compute_descriptor_metrics over the whole loader; the
            # per-sample arithmetic (mp_pair_metrics) and the RANSAC homography estimate (mp_find_homography) run on the GPU
            synchronize(); t0 = time.time()
            results = utils.compute_descriptor_metrics(net, loader_dataset, device, pred, args.threshold_keypoints,
                                                       args.threshold_homography)
            synchronize(); dt = time.time() - t0
            print('NN-mAP: {}'.format(results['nn_map']))
            print('M-Score: {}'.format(results['m_score']))
            print('Homography Correctness: {}'.format(results['h_correctness']))
            print('Matches: {}  ({:.1f} pairs/s)'.format(len(results['tp_optical']), len(dataset) / dt))
            results['config'] = config
            results['threshold_keypoints'] = args.threshold_keypoints
            results['threshold_homography'] = args.threshold_homography
            target_dir = os.path.join(args.model_dir, 'descriptor_evaluation')
            os.makedirs(target_dir, exist_ok=True)
            np.save(os.path.join(target_dir, os.path.split(args.model_dir.strip('/'))[-1] + '_' +
                                 time.strftime('%Y-%m-%d_%H-%M-%S', time.gmtime())), results)

        # get the sample and move it to the right device
        synchronize()
        t_start = time.time()
        data = dataset[args.index]
        data = utils.data_to_device(data, device)
        data = utils.data_unsqueeze(data, 0)

        synchronize()
        t_1 = time.time()
        out_optical = net(data['optical'])
        out_thermal = net(data['thermal'])
        synchronize()
        t_2 = time.time()

        if pred['nms'] > 0:
            # (box_nms with the top-k tie guard: an image whose top-k cut falls inside a plateau of tied scores is re-evaluated
            # with the tie-exact convolution algorithm, so the kept indices follow the reference's exact score order)
            out_optical['prob'] = utils.box_nms_tie_robust(net, data['optical'], out_optical, pred['nms'], pred['detection_threshold'],
                                                           keep_top_k=pred['topk'], on_cpu=pred['cpu_nms'],
                                                           valid_mask=data['optical']['valid_mask'])
            out_thermal['prob'] = utils.box_nms_tie_robust(net, data['thermal'], out_thermal, pred['nms'], pred['detection_threshold'],
                                                           keep_top_k=pred['topk'], on_cpu=pred['cpu_nms'],
                                                           valid_mask=data['thermal']['valid_mask'])
        else:
            out_optical['prob'] = out_optical['prob'] * data['optical']['valid_mask']
            out_thermal['prob'] = out_thermal['prob'] * data['thermal']['valid_mask']
        synchronize()
        t_3 = time.time()
        print('Loading the data took: {} s'.format(t_1 - t_start))
        print('Two forward passes took: {} s'.format(t_2 - t_1))
        print('Box nms: {} s'.format(t_3 - t_2))

        if args.plot or args.save_npz:
            H, W = data['optical']['image'].shape[2:]
            thr = pred['detection_threshold']
            pred_optical = torch.nonzero((out_optical['prob'][0].squeeze() > thr).float())
            pred_thermal = torch.nonzero((out_thermal['prob'][0].squeeze() > thr).float())
            desc_optical = utils.interpolate_descriptors(pred_optical, out_optical['desc'][0], H, W)
            desc_thermal = utils.interpolate_descriptors(pred_thermal, out_thermal['desc'][0], H, W)
            matches = utils.get_matches(desc_optical.cpu().numpy(), desc_thermal.cpu().numpy(),
                                        pred['matching']['method'], pred['matching']['knn_matches'],
                                        **pred['matching']['method_kwargs'])
            print('--------------------------------------------------------')
            print('Optical keypoints: {}'.format(pred_optical.shape[0]))
            print('Thermal keypoints: {}'.format(pred_thermal.shape[0]))
            print('Matches ({}): {}'.format(pred['matching']['method'], len(matches)))
            if matches:
                d = np.array([m.distance for m in matches])
                print('Match distance: min {:.4f} mean {:.4f} max {:.4f}'.format(d.min(), d.mean(), d.max()))
            print('--------------------------------------------------------')
            # align the images: homography from the matches (reference :209-216) next to the ground truth (:252-257)
            kpo = pred_optical.cpu().numpy(); kpt = pred_thermal.cpu().numpy()
            optical_pts = np.array([kpo[m.queryIdx][::-1] for m in matches]).reshape(-1, 2)
            thermal_pts = np.array([kpt[m.trainIdx][::-1] for m in matches]).reshape(-1, 2)
            H_est, mask = utils.find_homography_points(optical_pts, thermal_pts, pred['reprojection_threshold'], device=device)
            if H_est is None:
                H_est = np.eye(3, 3)
            eye = torch.eye(3)
            H_gt = np.matmul(data['thermal'].get('homography', eye[None])[0].cpu().numpy(),
                             np.linalg.inv(data['optical'].get('homography', eye[None])[0].cpu().numpy()))
            print('Estimated Homography:')
            print(H_est)
            print('RANSAC inliers: {} of {} matches'.format(int(np.sum(mask)), len(matches)))
            print('Ground Truth Homography:')
            print(H_gt)
            print('--------------------------------------------------------')
            # the aligned image: optical warped onto the thermal frame by the estimate (reference :218,
            # cv2.warpPerspective(im_optical, H_est, size, borderMode=cv2.BORDER_CONSTANT)), on the GPU
            from multipoint_amd.datasets.augmentation import warp_perspective_cv
            warped_image = warp_perspective_cv(data['optical']['image'][:1], H_est[None], border_reflect=False)
            inside = warp_perspective_cv(torch.ones_like(data['optical']['image'][:1]), H_est[None]) > 0.999
            if bool(inside.any()):
                resid = (warped_image - data['thermal']['image'][:1]).abs()[inside].mean().item()
                print('Aligned optical vs thermal: mean |diff| {:.4f} over {:.1f} % of the frame'.format(
                    resid, 100.0 * inside.float().mean().item()))
            if args.save_npz:
                np.savez_compressed(args.save_npz, kp_optical=pred_optical.cpu().numpy(), kp_thermal=pred_thermal.cpu().numpy(),
                                    desc_optical=desc_optical.cpu().numpy(), desc_thermal=desc_thermal.cpu().numpy(),
                                    match_query=np.array([m.queryIdx for m in matches]),
                                    match_train=np.array([m.trainIdx for m in matches]),
                                    match_distance=np.array([m.distance for m in matches], dtype=np.float32),
                                    homography_estimated=H_est, homography_ground_truth=H_gt,
                                    warped_optical=warped_image[0, 0].cpu().numpy())


if __name__ == "__main__":
    main()
