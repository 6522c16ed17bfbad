"""CPU: the homographic-adaptation oracle (oracle/ha_oracle.py) and the product's host-side homography sampler against
the golden vectors the imported reference produced (tests/golden/make_golden_ha.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ha_oracle as HA


@pytest.fixture(scope='module')
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, 'homographic_adaptation.npz'))


def test_sample_homography_matches_reference_vectors(golden):
    """product host logic (no GPU involved): same np.random draws, same matrices as the reference."""
    from multipoint_amd.utils.homographies import sample_homography
    shape = golden['sample_shape']
    for i, kw in enumerate(json.loads(str(golden['sample_cases']))):
        for seed, want in enumerate(golden['sample_%d' % i]):
            np.random.seed(seed)
            got = sample_homography(np.array(shape), **kw)
            assert np.allclose(got, want, rtol=1e-9, atol=1e-11), (i, seed)


def test_sample_homography_maps_corners_onto_patch():
    from multipoint_amd.utils.homographies import sample_homography, get_perspective_transform, warp_keypoints
    np.random.seed(4)
    h = sample_homography(np.array([240, 320]), allow_artifacts=False, max_angle=0.3)
    corners_yx = np.array([[0, 0], [240, 0], [240, 320], [0, 320]])
    w = warp_keypoints(corners_yx, h, float)
    assert w.min() >= -1e-6 and (w[:, 0] <= 240 + 1e-6).all() and (w[:, 1] <= 320 + 1e-6).all()
    src = np.array([[0., 0.], [0., 5.], [7., 5.], [7., 0.]])
    t = get_perspective_transform(src, src * 2 + 1)
    assert np.allclose(t, [[2, 0, 1], [0, 2, 1], [0, 0, 1]], atol=1e-12)
    with pytest.raises(ValueError):
        get_perspective_transform(src[:3], src[:3])


def test_oracle_reproduces_reference_driver(oracle, golden):
    cases = json.loads(str(golden['ha_cases']))
    homs_cfg = json.loads(str(golden['ha_homographies_cfg']))
    for name, (pair, img_seed, _, hc) in cases.items():
        cfg = json.loads(str(golden['model_cfg_pair'])) if pair else dict(oracle.SHIPPED_MODEL_CONFIG)
        sd = oracle.make_weights(int(golden['weight_seed']), cfg)
        img = oracle.make_images(img_seed, 4 if pair else 2, 64, 64)
        hc = dict(hc, homographies=homs_cfg)
        homs = golden['ha_%s_homographies' % name]
        if pair:
            flags = [torch.ones(2, 1, dtype=torch.bool), torch.zeros(2, 1, dtype=torch.bool)]
            out, _ = HA.homographic_adaptation(
                [img[:2], img[2:]], lambda i, x: oracle.forward(sd, x, cfg, is_optical=flags[i])['prob'], hc, homs,
                aggregation=hc['aggregation'])
        else:
            out, _ = HA.homographic_adaptation([img], lambda i, x: oracle.forward(sd, x, cfg)['prob'], hc, homs)
        want = torch.from_numpy(golden['ha_%s_out' % name])
        assert out.shape == want.shape
        assert (out - want).abs().max().item() <= 1e-6, name


def test_cv2_stand_ins():
    ones = np.ones((6, 8))
    eye = np.eye(3)
    assert np.array_equal(HA.cv2_warp_perspective_nearest(ones, eye, (8, 6)), ones)
    shift = np.array([[1., 0, 2], [0, 1, 1], [0, 0, 1]])            # dst(x, y) = src(x - 2, y - 1)
    w = HA.cv2_warp_perspective_nearest(np.arange(48.).reshape(6, 8), shift, (8, 6))
    assert w[1, 2] == 0 and w[5, 7] == 37 and (w[0] == 0).all() and (w[:, :2] == 0).all()
    m = np.ones((7, 9)); m[3, 4] = 0
    e = HA.cv2_erode(m, np.ones((3, 3), np.float32))
    assert e.sum() == 7 * 9 - 9 and e[0, 0] == 1                     # the image border does not erode
    assert HA.compute_valid_mask((7, 9), eye, 1, mask_border=True).sum() == 5 * 7
    assert HA.compute_valid_mask((7, 9), eye, 1, mask_border=False).sum() == 7 * 9
    assert HA.compute_valid_mask((7, 9), eye, 0, mask_border=True).sum() == 7 * 9


def test_kornia_stand_in_is_a_pixel_space_warp():
    """warp_perspective(src, M) samples src at M^-1 (x, y): integer translations and flips are exact."""
    torch.manual_seed(0)
    src = torch.rand(2, 1, 12, 16)
    M = torch.tensor([[1., 0, 3], [0, 1, -2], [0, 0, 1]])[None].repeat(2, 1, 1)
    out = HA.warp_perspective(src, M, (12, 16))
    assert torch.allclose(out[:, :, :10, 3:], src[:, :, 2:, :13], atol=1e-5)
    assert out[:, :, 10:].abs().max() <= 1e-5 and out[:, :, :, :3].abs().max() <= 1e-5
    flip = torch.tensor([[-1., 0, 15], [0, 1, 0], [0, 0, 1]])[None].repeat(2, 1, 1)
    assert torch.allclose(HA.warp_perspective(src, flip, (12, 16)), src.flip(-1), atol=1e-5)


def test_host_side_checks_without_gpu():
    import multipoint_amd.utils as U
    with pytest.raises(ValueError):
        U.homographic_adaptation({'image': torch.zeros(1, 1, 8, 8)}, None, {'num': 0})
    with pytest.raises(ValueError):
        U.homographic_adaptation({'image': torch.zeros(1, 1, 8, 8)}, None, {'filter_size': 4})
    d = {'optical': {'image': torch.zeros(1, 1, 8, 8)}, 'thermal': {'image': torch.zeros(1, 1, 8, 8)}}
    with pytest.raises(ValueError):
        U.homographic_adaptation_multispectral(d, None, {'aggregation': 'max'})
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):                            # no CPU fallback
            U.homographic_adaptation({'image': torch.zeros(1, 1, 8, 8)}, None, {'num': 1})
    assert U.homography_adaptation_default_config['num'] == 100     # the defaults are not written into
    k = U.get_gaussian_filter(5)
    assert k.shape == (1, 1, 5, 5) and abs(k.sum().item() - 1) < 1e-6
    assert torch.allclose(k, HA.gaussian_weights(5), atol=1e-8)
    pts = np.array([[0, 0], [5, 7], [-1, 3], [4, 9]])
    assert U.filter_points(pts, (5, 9)).tolist() == [[0, 0]]


# ------------------------------------------------------------------------------------------------------------------
# cv2.warpPerspective(INTER_LINEAR) restatement used by the dataset's homographic augmentation (parity unpinned: cv2
# is absent; these are the defining properties of the published algorithm)
# ------------------------------------------------------------------------------------------------------------------
def test_cv2_linear_warp_properties():
    rng = np.random.default_rng(0)
    img = rng.random((48, 80), dtype=np.float32)
    for border in ('reflect101', 'constant'):
        assert np.array_equal(HA.cv2_warp_perspective_linear(img, np.eye(3), (80, 48), border), img)
    T = np.array([[1, 0, 3.], [0, 1, -2.], [0, 0, 1]])
    o = HA.cv2_warp_perspective_linear(img, T, (80, 48), 'constant')
    assert np.array_equal(o[:46, 3:], img[2:, :77]) and (o[:, :3] == 0).all() and (o[46:] == 0).all()
    o = HA.cv2_warp_perspective_linear(img, T, (80, 48), 'reflect101')
    assert np.array_equal(o[:46, 3:], img[2:, :77])
    assert np.array_equal(o[0, :3], img[2, [3, 2, 1]])                 # gfedcb|abcdefgh|gfedcba
    assert np.array_equal(o[46:, 3:], img[[46, 45], :77])
    # half-pixel shift: weights 16/32 exactly -> mean of the two neighbours
    o = HA.cv2_warp_perspective_linear(img, np.array([[1, 0, .5], [0, 1, 0.], [0, 0, 1]]), (80, 48), 'constant')
    assert np.array_equal(o[:, 1:], (img[:, 1:] * np.float32(.5) + img[:, :-1] * np.float32(.5)).astype(np.float32))
    # coordinates are quantised to 1/32 px: shifts of 1/64 px less than a grid point round to it (half to even)
    a = HA.cv2_warp_perspective_linear(img, np.array([[1, 0, 10 / 32], [0, 1, 0.], [0, 0, 1]]), (80, 48))
    b = HA.cv2_warp_perspective_linear(img, np.array([[1, 0, 10 / 32 + 1 / 80], [0, 1, 0.], [0, 0, 1]]), (80, 48))
    assert np.array_equal(a, b)
    # a constant image stays constant under reflection for any homography (weights sum to 1 up to fp32 rounding)
    np.random.seed(3)
    from multipoint_amd.utils.homographies import sample_homography
    Hm = sample_homography((48, 80), perspective_amplitude_x=0.2, perspective_amplitude_y=0.2)
    c = HA.cv2_warp_perspective_linear(np.full((48, 80), 0.75, np.float32), Hm, (80, 48), 'reflect101')
    assert np.abs(c - 0.75).max() <= 2e-7
    # against float64 bilinear interpolation at the exact coordinates: within the 1/64 px quantisation x gradient
    ys, xs = np.mgrid[0:48, 0:80].astype(np.float64)
    smooth = (np.sin(xs / 9.0) + np.cos(ys / 7.0)).astype(np.float32)
    w = HA.cv2_warp_perspective_linear(smooth, Hm, (80, 48), 'constant')
    Mi = np.linalg.inv(Hm)
    den = Mi[2, 0] * xs + Mi[2, 1] * ys + Mi[2, 2]
    u, v = (Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]) / den, (Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]) / den
    inside = (u >= 0) & (u <= 78.9) & (v >= 0) & (v <= 46.9)
    u0, v0 = np.floor(u).astype(int).clip(0, 78), np.floor(v).astype(int).clip(0, 46)
    fu, fv = u - u0, v - v0
    s64 = smooth.astype(np.float64)
    exact = (s64[v0, u0] * (1 - fu) * (1 - fv) + s64[v0, u0 + 1] * fu * (1 - fv) +
             s64[v0 + 1, u0] * (1 - fu) * fv + s64[v0 + 1, u0 + 1] * fu * fv)
    # |gradient| <= 1/9 + 1/7 per pixel, coordinate error <= 1/64 px per axis
    assert inside.sum() > 500 and np.abs(w - exact)[inside].max() < (1 / 9 + 1 / 7) / 64 + 1e-6


def test_cv2_invert3_and_host_mirror():
    from multipoint_amd.datasets.augmentation import cv_invert3
    rng = np.random.default_rng(1)
    for _ in range(20):
        M = rng.normal(size=(3, 3)) + 2 * np.eye(3)
        assert np.allclose(HA.cv2_invert3(M), np.linalg.inv(M), rtol=1e-10, atol=1e-12)
        assert np.array_equal(HA.cv2_invert3(M), cv_invert3(M))      # product host code and oracle: same bits
    assert np.array_equal(HA.cv2_invert3(np.ones((3, 3))), np.zeros((3, 3)))
    assert np.array_equal(cv_invert3(np.ones((3, 3))), np.zeros((3, 3)))


def test_oracle_homographic_augmentation_keypoints():
    rng = np.random.default_rng(2)
    img = rng.random((40, 64), dtype=np.float32)
    kp = np.stack([rng.integers(0, 40, 30), rng.integers(0, 64, 30)], axis=1)
    T = np.array([[1, 0, 10.], [0, 1, 5.], [0, 0, 1]])
    w, pts, mask = HA.homographic_augmentation(img, kp, T)
    keep = (kp[:, 0] + 5 < 40) & (kp[:, 1] + 10 < 64)
    assert np.array_equal(pts, kp[keep] + np.array([[5, 10]]))
    assert mask[5:, 10:].all() and not mask[:5].any() and not mask[:, :10].any()
    w, pts, mask = HA.homographic_augmentation(img, np.zeros((0, 2), int), T)
    assert pts.shape == (0, 2)
