"""CPU: the oracle (oracle/mp_oracle.py) against the committed golden vectors that were produced by
importing the reference itself (tests/golden/make_golden.py).  This is what pins the oracle on the
GPU box, where /root/reference does not exist."""
import os

import numpy as np
import pytest
import torch


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_forward_64x64_full(oracle, golden_dir):
    g = _load(golden_dir, 'forward_64x64.npz')
    sd = oracle.make_weights(int(g['weight_seed']), oracle.SHIPPED_MODEL_CONFIG)
    img = oracle.make_images(int(g['image_seed']), int(g['B']), int(g['H']), int(g['W']))
    out = oracle.forward(sd, img, oracle.SHIPPED_MODEL_CONFIG)
    assert out['prob'].shape == (2, 1, 64, 64) and out['desc'].shape == (2, 64, 8, 8)
    assert np.abs(out['prob'].numpy() - g['prob']).max() <= 1e-6
    assert np.abs(out['desc'].numpy() - g['desc']).max() <= 1e-6


def test_forward_240x320_samples(oracle, golden_dir):
    g = _load(golden_dir, 'forward_240x320.npz')
    sd = oracle.make_weights(int(g['weight_seed']), oracle.SHIPPED_MODEL_CONFIG)
    img = oracle.make_images(int(g['image_seed']), 1, 240, 320)
    out = oracle.forward(sd, img, oracle.SHIPPED_MODEL_CONFIG)
    p = out['prob'].numpy().ravel(); d = out['desc'].numpy().ravel()
    assert np.abs(p[g['prob_idx']] - g['prob_val']).max() <= 1e-6
    assert np.abs(d[g['desc_idx']] - g['desc_val']).max() <= 1e-6
    assert abs(p.astype(np.float64).sum() - float(g['prob_sum'])) <= 1e-3
    assert abs(np.abs(d.astype(np.float64)).sum() - float(g['desc_abs_sum'])) <= 1e-2
    # the synthetic detector is "trained-like": a few thousand candidates, not 69 % of all pixels
    assert 500 < int((p > 0.015).sum()) < 20000


@pytest.mark.parametrize('name,upd', [('multispectral', {'multispectral': True}), ('zero_pad', {'reflection_pad': False}),
                                      ('bn_first', {'bn_first': True}), ('desc256', {'descriptor_size': 256}),
                                      ('no_final_bn', {'final_batchnorm': False}),
                                      ('single_conv', {'double_convolution': False}),
                                      ('single_conv_ms_zero_pad', {'double_convolution': False, 'multispectral': True,
                                                                   'reflection_pad': False, 'bn_first': True}),
                                      ('no_normalize', {'normalize_descriptors': False})])
def test_forward_variants(oracle, golden_dir, name, upd):
    g = _load(golden_dir, 'forward_variants.npz')
    cfg = dict(oracle.SHIPPED_MODEL_CONFIG); cfg.update(upd)
    sd = oracle.make_weights(int(g['weight_seed']), cfg)
    img = oracle.make_images(int(g['image_seed']), 3, 32, 48)
    flags = torch.tensor([[True], [False], [True]])
    out = oracle.forward(sd, img, cfg, is_optical=flags)
    assert np.abs(out['prob'].numpy() - g[name + '_prob']).max() <= 1e-6
    assert np.abs(out['desc'].numpy() - g[name + '_desc']).max() <= 2e-6 * max(1.0, np.abs(g[name + '_desc']).max())


def test_forward_logits(oracle, golden_dir):
    g = _load(golden_dir, 'forward_variants.npz')
    sd = oracle.make_weights(0, oracle.SHIPPED_MODEL_CONFIG)
    out = oracle.forward(sd, oracle.make_images(int(g['image_seed']), 3, 32, 48), oracle.SHIPPED_MODEL_CONFIG,
                         return_logits=True)
    assert out['prob'] is None
    assert np.abs(out['logits'].numpy() - g['logits']).max() <= 1e-5


def test_sampling(oracle, golden_dir):
    g = _load(golden_dir, 'sampling.npz')
    rows = oracle.interpolate_descriptors(g['keypoints'], g['desc'], int(g['H']), int(g['W']))
    assert rows.shape == g['rows'].shape
    assert np.abs(rows - g['rows']).max() <= 1e-6
    rows_t = oracle.interpolate_descriptors_torch(g['keypoints'], g['desc'], int(g['H']), int(g['W']))
    assert np.abs(rows_t - g['rows']).max() <= 1e-7
    assert np.abs(np.linalg.norm(rows, axis=1) - 1).max() < 1e-5


def test_matcher(oracle, golden_dir):
    g = _load(golden_dir, 'matcher.npz')
    q, t, d = oracle.nn_match(g['d1'], g['d2'], float(g['threshold']))
    assert np.array_equal(q, g['query']) and np.array_equal(t, g['train'])
    assert np.abs(d - g['distance']).max() <= 1e-6
    assert len(q) >= 80          # the fixture holds 100 true correspondences


def test_depth_to_space(oracle, golden_dir):
    g = _load(golden_dir, 'depth_to_space.npz')
    x = torch.from_numpy(g['x'])
    assert np.array_equal(oracle.depth_to_space(x, 8).numpy(), g['d2s'])
    assert np.array_equal(g['s2d'], g['x'])


def test_forward_magicleap(oracle, golden_dir):
    g = _load(golden_dir, 'forward_magicleap.npz')
    sd = oracle.make_weights_magicleap(int(g['weight_seed']))
    out = oracle.forward_magicleap(sd, oracle.make_images(int(g['image_seed']), 2, 32, 48))
    assert np.abs(out['logits'].numpy() - g['logits']).max() <= 1e-5
    assert np.abs(out['desc'].numpy() - g['desc']).max() <= 1e-6
    assert np.abs(out['prob'].numpy() - g['prob']).max() <= 1e-6


def test_autocast_restatement_is_fp16_noise_away_from_fp32(oracle):
    """mixed_precision restatement (PARITY UNPINNED: autocast needs CUDA): every intermediate is an fp16 value, the
    result stays within fp16 noise of the fp32 forward, descriptors are normalised in fp32."""
    import torch
    cfg = dict(oracle.SHIPPED_MODEL_CONFIG); cfg16 = dict(cfg); cfg16['mixed_precision'] = True
    sd = oracle.make_weights(0, cfg)
    img = oracle.make_images(9, 2, 64, 64)
    a, b = oracle.forward(sd, img, cfg), oracle.forward(sd, img, cfg16)
    assert 0 < (a['prob'] - b['prob']).abs().max().item() < 2e-2
    assert 0 < (a['desc'] - b['desc']).abs().max().item() < 4e-3
    assert (b['desc'].pow(2).sum(1).sqrt() - 1).abs().max().item() < 1e-5
    lg = oracle.forward(sd, img, cfg16, return_logits=True)['logits']
    assert torch.equal(lg, lg.half().float())


def test_detector_metrics_golden(oracle, golden_dir):
    """oracle.compute_tp_fp_dist / detector_precision_recall vs the outputs of the reference's compute_tp_fp_dist
    (multipoint/utils/evaluation.py:56-97) and the compute_detector_metrics tail (tests/golden/make_golden_eval.py)."""
    g = np.load(os.path.join(golden_dir, 'detector_metrics.npz'))
    tp, fp, prob, n_gt, dist = [], [], [], 0, []
    for i in range(4):
        r = oracle.compute_tp_fp_dist(g['prob_%d' % i], g['keypoints_%d' % i])
        assert np.array_equal(r[0], g['tp_%d' % i]) and np.array_equal(r[2], g['sorted_prob_%d' % i])
        assert r[3] == int(g['n_gt_%d' % i]) and np.array_equal(r[4], g['dist_%d' % i])
        tp.append(r[0]); fp.append(r[1]); prob.append(r[2].astype(np.float64)); n_gt += r[3]; dist.append(r[4])
    assert 0 < np.concatenate(tp).sum() < len(np.concatenate(tp))
    p, r, pr = oracle.detector_precision_recall(np.concatenate(tp), np.concatenate(fp), np.concatenate(prob), n_gt)
    assert np.array_equal(p, g['precision']) and np.array_equal(r, g['recall']) and np.array_equal(pr, g['all_prob'])
    assert np.array_equal(np.concatenate(dist).astype(np.float64), g['all_dist'])
    assert float(np.sum(p[1:] * (r[1:] - r[:-1]))) == float(g['mAP'])


def test_dataset_augmentation_golden(golden_dir):
    """The oracle's homographic augmentation + the product's host sampler replayed with the reference's draw order
    vs the samples the reference's ImagePairDataset emitted (tests/golden/make_golden_eval.py)."""
    import random
    from oracle import ha_oracle as HA
    from multipoint_amd.utils.homographies import sample_homography
    HCFG = HA.PREDICTION_AUGMENTATION
    g = np.load(os.path.join(golden_dir, 'dataset_augmentation.npz'))
    sides = set()
    for i in range(3):
        random.seed(302 + i); np.random.seed(400 + i)
        warp_optical = bool(random.randint(0, 1))
        side, other = ('optical', 'thermal') if warp_optical else ('thermal', 'optical')
        sides.add(side)
        hom = sample_homography((48, 64), **HCFG['params'])
        img, pts, mask = HA.homographic_augmentation(g['in_%s_%d' % (side, i)], g['in_keypoints_%d' % i], hom)
        assert np.array_equal(g['pair_%d_%s_homography' % (i, side)], hom.astype(np.float32))
        assert np.array_equal(g['pair_%d_%s_homography' % (i, other)], np.eye(3, dtype=np.float32))
        assert np.array_equal(g['pair_%d_%s_image' % (i, side)][0], img)
        assert np.array_equal(g['pair_%d_%s_valid_mask' % (i, side)][0], mask.astype(bool))
        km = np.zeros((48, 64), bool); km[pts[:, 0], pts[:, 1]] = True
        assert np.array_equal(g['pair_%d_%s_keypoints' % (i, side)], km)
        assert np.array_equal(g['pair_%d_%s_image' % (i, other)][0], g['in_%s_%d' % (other, i)])
        assert g['pair_%d_%s_valid_mask' % (i, other)].all()
    assert len(sides) == 2                                   # the seeds exercise both branches (:211, :221)
    for i in range(3):
        random.seed(502 + i); np.random.seed(600 + i)
        is_optical = bool(random.randint(0, 1))
        hom = sample_homography((48, 64), **HCFG['params'])
        img, pts, mask = HA.homographic_augmentation(g['in_%s_%d' % ('optical' if is_optical else 'thermal', i)],
                                                     g['in_keypoints_%d' % i], hom)
        assert bool(g['single_%d_is_optical' % i][0]) is is_optical
        assert np.array_equal(g['single_%d_image' % i][0], img) and np.array_equal(g['single_%d_valid_mask' % i][0], mask.astype(bool))
        km = np.zeros((48, 64), bool); km[pts[:, 0], pts[:, 1]] = True
        assert np.array_equal(g['single_%d_keypoints' % i], km)


# ------------------------------------------------------------------ fp16 (mixed_precision) oracle pinned to the reference
def _f16_case(oracle, z, c):
    import json
    cfg = dict(oracle.SHIPPED_MODEL_CONFIG); cfg['mixed_precision'] = True
    if c + '_cfg' in z.files:                    # cases c, d: another model config under autocast
        cfg.update(json.loads(str(z[c + '_cfg'])))
    sd = oracle.make_weights(int(z['weight_seed']), cfg)
    img = oracle.make_images(int(z[c + '_seed']), int(z[c + '_B']), int(z[c + '_H']), int(z[c + '_W']))
    return cfg, sd, img


@pytest.mark.parametrize('c', ['a', 'b', 'c', 'd'])
def test_f16_oracle_against_reference_autocast_fixture(oracle, golden_dir, c):
    """tests/golden/forward_f16.npz = the imported reference with `mixed_precision: true`, run under
    torch.autocast('cpu', float16) in place of torch.cuda.amp.autocast (make_golden_f16.py; MultiPoint.py:99-104).  Two
    correct fp16 evaluations agree only to rounding flips that compound through 12 layers, so the oracle's restatement of
    autocast's rounding points (mp_oracle._block/_head) is pinned DISTRIBUTIONALLY, in fp16 steps of the quantity the
    network rounds: most elements within a step, a tail of a few steps, no bias.  Observed: logits median 0.25 / p99.9
    2.0-2.5 / max 3 steps, descriptors 0.27 / 1.9-2.0 / 2.4, prob above threshold median rel 3.6e-3-5.9e-3, p99.9 2.3e-2."""
    from oracle import f16_stats as S
    z = np.load(os.path.join(golden_dir, 'forward_f16.npz'))
    cfg, sd, img = _f16_case(oracle, z, c)
    out = oracle.forward(sd, img, cfg)
    out['logits'] = oracle.forward(sd, img, cfg, return_logits=True)['logits']
    out = {k: v.numpy() for k, v in out.items() if v is not None}
    v, ax = S.fixture_views(z, out, c)
    ls, ds, ps = S.logits_stats(*v['logits']), S.desc_stats(*v['desc'], channel_axis=ax), S.prob_rel_stats(*v['prob'])
    assert ls['median'] <= 0.5 and ls['p999'] <= 4.0 and ls['max'] <= 6.0, ls
    assert ds['median'] <= 0.5 and ds['p999'] <= 4.0 and ds['max'] <= 6.0, ds
    # (no bias: only meaningful where the two evaluations differ at all -- the shallow channel_version 2 network agrees to a handful of
    # half-step flips, whose mean says nothing)
    for st in (ls, ds):
        assert st['mean_abs'] < 0.05 or abs(st['mean_signed']) <= 0.05 * st['mean_abs'], st
    assert ps['median_abs_rel'] <= 1e-2 and ps['p999_abs_rel'] <= 5e-2, ps
    assert S.unbiased(ps), ps
    if c == 'b':
        n = int((out['prob'] > 0.015).sum())
        assert abs(n - int(z['b_n_above_thr'])) <= 0.01 * int(z['b_n_above_thr'])


def test_f16_fixture_records_autocast_rounding_points(golden_dir):
    """The dtype every leaf module of the reference returned under autocast is part of the fixture: the rounding points the
    oracle restates (Conv2d -> fp16, BatchNorm2d -> fp16, pooling / ReLU on fp16) are the reference's, not an assumption."""
    import json
    z = np.load(os.path.join(golden_dir, 'forward_f16.npz'))
    dt = json.loads(str(z['dtypes']))
    kinds = {}
    for name, (kind, dtype) in dt.items():
        kinds.setdefault(kind, set()).add(dtype)
    assert kinds['Conv2d'] == {'torch.float16'} and kinds['BatchNorm2d'] == {'torch.float16'}
    assert kinds['ReLU'] == {'torch.float16'} and kinds['MaxPool2d'] == {'torch.float16'}
    assert kinds['Softmax2d'] == {'torch.float32'}
