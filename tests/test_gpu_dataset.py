"""GPU (MI355X): the dataset's homographic augmentation (multipoint/datasets/augmentation/augmentation.py:25-54,
ImagePairDataset.py:173-241) -- mp_warp_perspective_cv + mp_ha_valid_mask behind
multipoint_amd.datasets.augmentation -- against the oracle's restatement of cv2.warpPerspective(INTER_LINEAR).

The restated arithmetic is float64 coordinates -> 1/32-pixel fixed point -> float32 table weights, evaluated without
contraction on both sides: the comparison is BIT-EXACT."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import ha_oracle as HA

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
HCFG = HA.PREDICTION_AUGMENTATION          # the reference's prediction-config augmentation block


def _homs(seed, n, H, W):
    from multipoint_amd.utils.homographies import sample_homography
    np.random.seed(seed)
    return np.stack([sample_homography((H, W), **HCFG['params']) for _ in range(n)])


@pytest.mark.parametrize('H,W', [(240, 320), (50, 70), (40, 48), (8, 200)])
@pytest.mark.parametrize('reflect', [True, False])
def test_cv_warp_bit_exact(H, W, reflect):
    from multipoint_amd.datasets.augmentation import homographic_augmentation_batch
    rng = np.random.default_rng(H * 1000 + W)
    imgs = rng.random((5, 1, H, W), dtype=np.float32)
    homs = _homs(H + W, 5, H, W)
    homs[0] = np.eye(3)
    homs[1] = [[1, 0, 3], [0, 1, -2], [0, 0, 1]]
    for margin in (0, 3):
        got, mask = homographic_augmentation_batch(torch.from_numpy(imgs).to(DEV), homs, reflect, margin, True)
        assert got.shape == (5, 1, H, W) and mask.shape == (5, 1, H, W) and mask.dtype == torch.bool
        got, mask = got.cpu().numpy(), mask.cpu().numpy()
        for i in range(5):
            want = HA.cv2_warp_perspective_linear(imgs[i, 0], homs[i], (W, H), 'reflect101' if reflect else 'constant')
            assert np.array_equal(got[i, 0], want), (i, np.abs(got[i, 0] - want).max())
            wm = HA.compute_valid_mask((H, W), homs[i], margin * 2, True)
            assert np.array_equal(mask[i, 0], wm.astype(bool)), i
    assert np.array_equal(got[0, 0], imgs[0, 0])                                    # identity copies the image


def test_cv_warp_extreme_homographies():
    """points at infinity inside the frame, far out-of-frame sources, singular matrix: same pixels as the oracle."""
    from multipoint_amd.datasets.augmentation import homographic_augmentation_batch
    H, W = 64, 96
    rng = np.random.default_rng(5)
    imgs = rng.random((4, 1, H, W), dtype=np.float32)
    homs = np.stack([np.array([[1, 0, 0], [0, 1, 0], [0.02, 0.01, 1.0]]),           # horizon crosses the frame
                     np.array([[1e-3, 0, 5e4], [0, 1e-3, -7e4], [0, 0, 1]]),        # saturating source coordinates
                     np.array([[40.0, 0, -900], [0, 40.0, -700], [0, 0, 1]]),       # 40x zoom
                     np.array([[0.5, 0.2, 3], [-0.1, 0.7, 9], [1e-3, -2e-3, 1.0]])])
    for reflect in (True, False):
        got, _ = homographic_augmentation_batch(torch.from_numpy(imgs).to(DEV), homs, reflect, 0, False)
        got = got.cpu().numpy()
        for i in range(4):
            want = HA.cv2_warp_perspective_linear(imgs[i, 0], homs[i], (W, H), 'reflect101' if reflect else 'constant')
            assert np.array_equal(got[i, 0], want), (reflect, i)


def test_dataset_homographic_augmentation_matches_oracle(tmp_path):
    """ImagePairDataset over an .npz archive with labels, the reference's prediction augmentation block: same draws
    (random / np.random), same warped image, mask, warped keypoints and homography as the oracle."""
    from multipoint_amd.datasets import ImagePairDataset
    from multipoint_amd.utils.homographies import sample_homography
    H, W = 96, 128
    rng = np.random.default_rng(11)
    arrays, labels = {}, {}
    for i in range(3):
        arrays['s%d/optical' % i] = rng.random((H, W), dtype=np.float32)
        arrays['s%d/thermal' % i] = rng.random((H, W), dtype=np.float32)
        labels['s%d/keypoints' % i] = np.stack([rng.integers(0, H, 40), rng.integers(0, W, 40)], axis=1)
    fn, kfn = str(tmp_path / 'pairs.npz'), str(tmp_path / 'labels.npz')
    np.savez(fn, **arrays); np.savez(kfn, **labels)
    ds = ImagePairDataset({'filename': fn, 'keypoints_filename': kfn, 'single_image': False,
                           'augmentation': {'homographic': HCFG}})
    assert len(ds) == 3 and ds.returns_pair()
    warped_sides = set()
    for i in range(3):
        random.seed(100 + i); np.random.seed(200 + i)
        s = ds[i]
        random.seed(100 + i); np.random.seed(200 + i)
        warp_optical = bool(random.randint(0, 1))
        hom = sample_homography((H, W), **HCFG['params'])
        side, other = ('optical', 'thermal') if warp_optical else ('thermal', 'optical')
        warped_sides.add(side)
        img, pts, mask = HA.homographic_augmentation(arrays['s%d/%s' % (i, side)], labels['s%d/keypoints' % i], hom)
        assert s['name'] == 's%d' % i
        assert np.array_equal(s[side]['homography'].numpy(), hom.astype(np.float32))
        assert torch.equal(s[other]['homography'], torch.eye(3))
        assert s[side]['image'].dtype == torch.float32 and s[side]['image'].shape == (1, H, W)
        assert np.array_equal(s[side]['image'][0].numpy(), img)
        assert np.array_equal(s[side]['valid_mask'][0].numpy(), mask.astype(bool)) and not s[side]['valid_mask'].all()
        km = np.zeros((H, W), bool); km[pts[:, 0], pts[:, 1]] = True
        assert np.array_equal(s[side]['keypoints'].numpy(), km)
        assert np.array_equal(s[other]['image'][0].numpy(), arrays['s%d/%s' % (i, other)])
        assert s[other]['valid_mask'].all()
        k0 = labels['s%d/keypoints' % i]
        km = np.zeros((H, W), bool); km[k0[:, 0], k0[:, 1]] = True
        assert np.array_equal(s[other]['keypoints'].numpy(), km)
        assert bool(s['optical']['is_optical'][0]) and not bool(s['thermal']['is_optical'][0])
    # single_image mode: image, mask and labels of the randomly chosen spectrum
    ds1 = ImagePairDataset({'filename': fn, 'keypoints_filename': kfn, 'single_image': True,
                            'augmentation': {'homographic': HCFG}})
    random.seed(7); np.random.seed(8)
    s = ds1[1]
    random.seed(7); np.random.seed(8)
    is_optical = bool(random.randint(0, 1))
    hom = sample_homography((H, W), **HCFG['params'])
    img, pts, mask = HA.homographic_augmentation(arrays['s1/' + ('optical' if is_optical else 'thermal')],
                                                 labels['s1/keypoints'], hom)
    assert bool(s['is_optical'][0]) is is_optical and 'homography' not in s
    assert np.array_equal(s['image'][0].numpy(), img) and np.array_equal(s['valid_mask'][0].numpy(), mask.astype(bool))


def test_descriptor_metrics_over_augmented_pairs():
    """The evaluation driver over a loader whose ground-truth homographies come from the dataset augmentation, as in
    the reference's prediction config: warping a pair's thermal image = its optical image by H must make the
    keypoints of the two sides repeat under H (high repeatability with the net applied to both)."""
    import multipoint_amd.models as models
    import multipoint_amd.utils as U
    from multipoint_amd.datasets import SyntheticPairs
    from oracle import mp_oracle as O
    cfg = dict(O.SHIPPED_MODEL_CONFIG)
    net = models.MultiPoint(cfg); net.load_state_dict(O.make_weights(0, cfg)); net.to(DEV); net.eval()
    ds = SyntheticPairs({'num_samples': 4, 'height': 120, 'width': 160, 'augmentation': {'homographic': HCFG}})
    random.seed(1); np.random.seed(2)
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=0)
    batch = next(iter(loader))
    assert batch['optical']['homography'].shape == (2, 3, 3) and batch['thermal']['homography'].shape == (2, 3, 3)
    assert batch['optical']['valid_mask'].shape == (2, 1, 120, 160)
    pred = {'nms': 4, 'detection_threshold': 0.015, 'topk': 300, 'cpu_nms': True, 'reprojection_threshold': 3,
            'matching': {'method': 'bfmatcher', 'method_kwargs': {'crossCheck': True}, 'knn_matches': False}}
    random.seed(1); np.random.seed(2)
    res = U.compute_descriptor_metrics(net, loader, DEV, pred, 4, 3)
    assert set(res) >= {'nn_map', 'm_score', 'h_correctness'}
    assert np.isfinite(res['nn_map']) and np.isfinite(res['m_score'])


def test_dataset_against_reference_golden(tmp_path, golden_dir):
    """The product's ImagePairDataset (GPU warp + mask) vs the samples the REFERENCE's ImagePairDataset emitted for the
    same arrays and seeds (tests/golden/dataset_augmentation.npz, make_golden_eval.py): schema, which image is
    warped, homography, warped image, valid mask and label maps all equal."""
    from multipoint_amd.datasets import ImagePairDataset
    g = np.load(os.path.join(golden_dir, 'dataset_augmentation.npz'))
    arrays, labels = {}, {}
    for i in range(3):
        arrays['s%d/optical' % i], arrays['s%d/thermal' % i] = g['in_optical_%d' % i], g['in_thermal_%d' % i]
        labels['s%d/keypoints' % i] = g['in_keypoints_%d' % i]
    fn, kfn = str(tmp_path / 'pairs.npz'), str(tmp_path / 'labels.npz')
    np.savez(fn, **arrays); np.savez(kfn, **labels)
    hcfg = {k: v for k, v in HCFG.items() if k != 'mask_border'}           # as in the reference's yaml (default true)
    pair = ImagePairDataset({'filename': fn, 'keypoints_filename': kfn, 'single_image': False,
                             'augmentation': {'homographic': hcfg}})
    for i in range(3):
        random.seed(302 + i); np.random.seed(400 + i)
        s = pair[i]
        assert s['name'] == 's%d' % i and set(s) == {'optical', 'thermal', 'name'}
        for side in ('optical', 'thermal'):
            assert set(s[side]) == {'image', 'valid_mask', 'keypoints', 'homography', 'is_optical'}
            for k in ('image', 'valid_mask', 'keypoints', 'homography', 'is_optical'):
                want = g['pair_%d_%s_%s' % (i, side, k)]
                got = s[side][k].numpy()
                assert got.dtype == want.dtype and got.shape == want.shape, (i, side, k, got.dtype, want.dtype)
                assert np.array_equal(got, want), (i, side, k)
    single = ImagePairDataset({'filename': fn, 'keypoints_filename': kfn, 'single_image': True,
                               'augmentation': {'homographic': hcfg}})
    for i in range(3):
        random.seed(502 + i); np.random.seed(600 + i)
        s = single[i]
        assert set(s) == {'image', 'valid_mask', 'keypoints', 'is_optical', 'name'}
        for k in ('image', 'valid_mask', 'keypoints', 'is_optical'):
            want = g['single_%d_%s' % (i, k)]
            got = s[k].numpy()
            assert got.dtype == want.dtype and np.array_equal(got, want), (i, k)
