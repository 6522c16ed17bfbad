"""Build-container only (needs /root/reference): the oracle against the IMPORTED reference on fresh
seeds.  Skipped on the GPU box; the committed golden vectors (test_oracle_golden.py) travel instead."""
import collections
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.reference


@pytest.fixture(scope='module')
def ref():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import ref_shim
    return ref_shim.install()


def test_state_dict_layout_and_forward(oracle, ref):
    models, utils = ref
    cfg = dict(oracle.SHIPPED_MODEL_CONFIG)
    net = models.MultiPoint(dict(cfg)).eval()
    spec = oracle.state_dict_spec(cfg)
    rsd = net.state_dict()
    assert [k for k, _, _ in spec] == list(rsd.keys())
    for k, s, d in spec:
        assert tuple(rsd[k].shape) == tuple(s) and rsd[k].dtype == d
    sd = oracle.make_weights(21, cfg)
    net.load_state_dict(sd)
    img = oracle.make_images(22, 2, 120, 160)
    with torch.no_grad():
        r = net({'image': img})
    o = oracle.forward(sd, img, cfg)
    assert (r['prob'] - o['prob']).abs().max().item() <= 1e-7
    assert (r['desc'] - o['desc']).abs().max().item() <= 1e-7


def test_multispectral_default_config(oracle, ref):
    models, utils = ref
    cfg = {'multispectral': True, 'descriptor_size': 128}
    net = models.MultiPoint(dict(cfg)).eval()
    spec = oracle.state_dict_spec(cfg)
    assert [k for k, _, _ in spec] == list(net.state_dict().keys())
    sd = oracle.make_weights(5, cfg)
    net.load_state_dict(sd)
    img = oracle.make_images(6, 4, 32, 40)
    flags = torch.tensor([[True], [False], [False], [True]])
    with torch.no_grad():
        r = net({'image': img, 'is_optical': flags})
    o = oracle.forward(sd, img, cfg, is_optical=flags)
    assert (r['prob'] - o['prob']).abs().max().item() <= 1e-7
    assert (r['desc'] - o['desc']).abs().max().item() <= 1e-7


def test_interpolate_and_matcher(oracle, ref):
    models, utils = ref
    rng = np.random.default_rng(1)
    desc = rng.standard_normal((64, 60, 80)).astype(np.float32)
    kp = np.stack([rng.integers(0, 480, 500), rng.integers(0, 640, 500)], 1).astype(np.int64)
    kp_t = torch.from_numpy(kp.copy())
    r = utils.interpolate_descriptors(kp_t, torch.from_numpy(desc), 480, 640).numpy()
    assert np.array_equal(kp_t.numpy(), kp)                      # reference does not mutate its input
    assert np.abs(oracle.interpolate_descriptors(kp, desc, 480, 640) - r).max() <= 1e-6
    d1 = r[:250]; d2 = r[200:]
    m = utils.NNMatcher(0.7).match(d1, d2)
    q, t, d = oracle.nn_match(d1, d2, 0.7)
    assert [x.queryIdx for x in m] == list(q) and [x.trainIdx for x in m] == list(t)
    assert np.allclose([x.distance for x in m], d, atol=1e-7)
    with pytest.raises(ValueError):
        utils.NNMatcher(-1.0)


def test_product_host_helpers_match_reference(ref):
    """dict_update / fix_model_weigth_keys / depth_to_space of the product package vs the reference."""
    models, utils = ref
    import multipoint_amd.utils as U
    a = {'a': 1, 'b': {'c': 2, 'd': 3}}; b = {'b': {'c': 5}, 'e': 6}
    import copy
    assert U.dict_update(copy.deepcopy(a), b) == utils.dict_update(copy.deepcopy(a), b)
    w = collections.OrderedDict([('module__encoder.1.weight', 1), ('x__y__detector.4.bias', 2), ('plain', 3)])
    assert list(U.fix_model_weigth_keys(w).items()) == list(utils.fix_model_weigth_keys(w).items())
    x = torch.randn(2, 64, 3, 4)
    assert torch.equal(U.depth_to_space(x, 8), utils.depth_to_space(x, 8))
    assert torch.equal(U.space_to_depth(U.depth_to_space(x, 8), 8), x)
    import multipoint_amd.models as M
    assert M.MultiPoint.default_config == models.MultiPoint.default_config
    for cfg in ({'multispectral': False, 'descriptor_size': 64}, {'bn_first': True}, {'final_batchnorm': False}):
        ours = M.MultiPoint(dict(cfg)).state_dict_spec()
        theirs = models.MultiPoint(dict(cfg)).state_dict()
        assert [k for k, _, _ in ours] == list(theirs.keys())
        assert all(tuple(theirs[k].shape) == tuple(s) for k, s, _ in ours)


def test_magicleap_vs_reference(oracle, ref):
    models, utils = ref
    net = models.SuperPointMagicLeap().eval()
    sd = oracle.make_weights_magicleap(8)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd)
    img = oracle.make_images(9, 2, 64, 96)
    with torch.no_grad():
        r = net({'image': img})
    o = oracle.forward_magicleap(sd, img)
    for k in ('logits', 'desc', 'prob'):
        assert (r[k] - o[k]).abs().max().item() <= 1e-6
    import multipoint_amd.models as M
    assert [k for k, _, _ in M.SuperPointMagicLeap().state_dict_spec()] == list(sd.keys())


def test_descriptor_metrics_against_reference_driver(oracle, ref, monkeypatch):
    """The REFERENCE's own compute_descriptor_metrics (multipoint/utils/evaluation.py:209-439) run here on CPU, with
    the absent third-party calls (torchvision nms, cv2 BFMatcher / perspectiveTransform) served by the oracle's
    restatements -- so this pins the bookkeeping (ground-truth homography, correctness matrices, n_gt, true positives,
    M-score, precision/recall, NN-mAP) of the oracle + the product's host code, not the third-party arithmetic."""
    import sys
    import types
    models, utils = ref
    import multipoint.utils.evaluation as ev
    import multipoint.utils.utils as ru
    import multipoint.utils.matching as rm
    import multipoint.utils.homographies as rh
    cv2 = sys.modules['cv2']

    def nms(boxes, scores, iou):
        keep = oracle.nms_greedy(boxes.numpy().astype(np.float32), scores.numpy().astype(np.float32), float(iou))
        return torch.as_tensor(np.asarray(keep), dtype=torch.int64)

    def batched_nms(boxes, scores, idxs, iou):
        keep = []
        for i in torch.unique(idxs):
            sel = torch.nonzero(idxs == i)[:, 0]
            keep.append(sel[nms(boxes[sel], scores[sel], iou)])
        keep = torch.cat(keep) if keep else torch.zeros(0, dtype=torch.int64)
        return keep[torch.argsort(scores[keep], descending=True, stable=True)]

    class BFMatcher:
        def __init__(self, norm, crossCheck=False):
            assert crossCheck
        def match(self, d1, d2):
            q, t, d = oracle.bf_match_crosscheck(d1, d2)
            return [cv2.DMatch(int(a), int(b), float(c)) for a, b, c in zip(q, t, d)]

    def perspectiveTransform(pts, h):
        out = oracle.warp_keypoints(pts[0][:, ::-1], h)          # oracle takes / returns (y, x)
        return out[None, :, ::-1]

    class KeyPoint:
        def __init__(self, x, y, size):
            self.pt = (x, y)
    monkeypatch.setattr(ru, 'nms', nms, raising=False); monkeypatch.setattr(ru, 'batched_nms', batched_nms, raising=False)
    for name, val in (('BFMatcher', BFMatcher), ('perspectiveTransform', perspectiveTransform), ('KeyPoint', KeyPoint),
                      ('findHomography', lambda *a, **k: (None, None)), ('RANSAC', 8), ('NORM_L2', 4)):
        monkeypatch.setattr(cv2, name, val, raising=False)

    cfg = dict(oracle.SHIPPED_MODEL_CONFIG)
    sd = oracle.make_weights(31, cfg)
    net = models.MultiPoint(dict(cfg)).eval(); net.load_state_dict(sd)
    H, W, B = 96, 128, 3
    rng = np.random.default_rng(5)
    opt = oracle.make_images(41, B, H, W)
    th = (opt + 0.02 * torch.from_numpy(rng.standard_normal(opt.shape).astype(np.float32))).clamp(0, 1)
    ho = torch.eye(3).repeat(B, 1, 1); ht = torch.eye(3).repeat(B, 1, 1)
    ht[:, 0, 2] = torch.tensor([0.0, 1.0, -2.0]); ht[1, 0, 0] = 1.01
    ones = torch.ones((B, 1, H, W), dtype=torch.bool)
    batch = {'optical': {'image': opt, 'valid_mask': ones, 'homography': ho, 'is_optical': torch.ones(B, 1, dtype=torch.bool)},
             'thermal': {'image': th, 'valid_mask': ones, 'homography': ht, 'is_optical': torch.zeros(B, 1, dtype=torch.bool)}}
    pred = {'nms': 4, 'detection_threshold': 0.015, 'topk': 150, 'cpu_nms': True, 'reprojection_threshold': 3,
            'matching': {'method': 'bfmatcher', 'method_kwargs': {'crossCheck': True}, 'knn_matches': False}}
    with torch.no_grad():
        r = ev.compute_descriptor_metrics(net, [batch], torch.device('cpu'), pred, 4.0, 4.0)

    # the same through the oracle (per pair) + the product's host bookkeeping
    from multipoint_amd.utils.evaluation import summarize_descriptor_metrics
    pairs = oracle.process_pairs(sd, cfg, opt, th, nms=4, detection_threshold=0.015, topk=150)
    tp_o, tp_t, dist, ms_o, ms_t, ngo, ngt = [], [], [], [], [], 0, 0
    for p, rec in enumerate(pairs):
        m = oracle.descriptor_metrics_pair(rec['kp_optical'], rec['kp_thermal'], rec['match_query'], rec['match_train'],
                                           ho[p], ht[p], 4.0, H, W)
        tp_o.append(m['tp_optical']); tp_t.append(m['tp_thermal']); dist.append(rec['match_dist'])
        ngo += m['n_gt_optical']; ngt += m['n_gt_thermal']
        ms_o.append(m['tp_optical'].sum() / m['N_optical'] if m['N_optical'] else 0.0)
        ms_t.append(m['tp_thermal'].sum() / m['N_thermal'] if m['N_thermal'] else 0.0)
    d = np.concatenate(dist)
    mine = summarize_descriptor_metrics(np.concatenate(tp_o), d, np.concatenate(tp_t), d, ngo, ngt, np.array(ms_o), np.array(ms_t))
    assert len(r['tp_optical']) == len(mine['tp_optical']) > 50
    assert r['tp_optical'].sum() == mine['tp_optical'].sum() and r['tp_thermal'].sum() == mine['tp_thermal'].sum()
    assert 0 < r['tp_optical'].sum() < len(r['tp_optical'])               # the case exercises true AND false positives
    # (the driver's BFMatcher stub returns ||a-b||, process_pairs the NNMatcher form sqrt(2-2ab): equal to ~1e-5 near 0)
    assert np.allclose(np.sort(r['distance_optical']), np.sort(mine['distance_optical']), atol=1e-4)
    assert np.allclose(r['m_score_optical'], mine['m_score_optical']) and np.allclose(r['m_score_thermal'], mine['m_score_thermal'])
    assert abs(r['m_score'] - mine['m_score']) < 1e-12
    assert abs(r['nn_map_optical'] - mine['nn_map_optical']) < 1e-6 and abs(r['nn_map_thermal'] - mine['nn_map_thermal']) < 1e-6
    assert np.allclose(r['recall_optical'][-2], mine['recall_optical'][-2])


def test_repeatability_against_reference_driver(oracle, ref, monkeypatch):
    """The reference's compute_repeatability_multispectral (evaluation.py:105-200) run here with oracle stubs for the
    absent torchvision / cv2 calls, against oracle.repeatability_pair on the oracle's own keypoints."""
    import sys
    models, utils = ref
    import multipoint.utils.evaluation as ev
    import multipoint.utils.utils as ru
    cv2 = sys.modules['cv2']

    def nms(boxes, scores, iou):
        keep = oracle.nms_greedy(boxes.numpy().astype(np.float32), scores.numpy().astype(np.float32), float(iou))
        return torch.as_tensor(np.asarray(keep), dtype=torch.int64)

    def batched_nms(boxes, scores, idxs, iou):
        keep = []
        for i in torch.unique(idxs):
            sel = torch.nonzero(idxs == i)[:, 0]
            keep.append(sel[nms(boxes[sel], scores[sel], iou)])
        keep = torch.cat(keep) if keep else torch.zeros(0, dtype=torch.int64)
        return keep[torch.argsort(scores[keep], descending=True, stable=True)]
    monkeypatch.setattr(ru, 'nms', nms, raising=False); monkeypatch.setattr(ru, 'batched_nms', batched_nms, raising=False)
    monkeypatch.setattr(cv2, 'perspectiveTransform',
                        lambda pts, h: oracle.warp_keypoints(pts[0][:, ::-1], h)[None, :, ::-1], raising=False)
    cfg = dict(oracle.SHIPPED_MODEL_CONFIG)
    sd = oracle.make_weights(33, cfg)
    net = models.MultiPoint(dict(cfg)).eval(); net.load_state_dict(sd)
    H, W, B = 96, 128, 3
    rng = np.random.default_rng(6)
    opt = oracle.make_images(43, B, H, W)
    th = (opt + 0.02 * torch.from_numpy(rng.standard_normal(opt.shape).astype(np.float32))).clamp(0, 1)
    ho = torch.eye(3).repeat(B, 1, 1); ht = torch.eye(3).repeat(B, 1, 1)
    ht[:, 0, 2] = torch.tensor([0.0, 2.0, -3.0]); ho[2, 1, 2] = 1.5; ht[1, 0, 0] = 1.02
    ones = torch.ones((B, 1, H, W), dtype=torch.bool)
    batch = {'optical': {'image': opt, 'valid_mask': ones, 'homography': ho}, 'thermal': {'image': th, 'valid_mask': ones, 'homography': ht}}
    config = {'prediction': {'nms': 4, 'detection_threshold': 0.015, 'topk': 120, 'cpu_nms': True}}
    with torch.no_grad():
        mean, rep, n_o, n_t = ev.compute_repeatability_multispectral(net, [batch], torch.device('cpu'), config, distance_thresh=3)
    mine = []
    for p in range(B):
        kps = []
        for img in (opt, th):
            prob = oracle.forward(sd, img[p:p + 1], cfg)['prob']
            nmsp = oracle.box_nms(prob.numpy(), 4, 0.015, keep_top_k=120)
            kps.append(oracle.keypoints_from_map(nmsp[0, 0], 0.015))
        assert len(kps[0]) == n_o[p] and len(kps[1]) == n_t[p]
        c1, c2, nt, no = oracle.repeatability_pair(kps[0], kps[1], ho[p], ht[p], H, W, 3)
        mine.append((c1 + c2) / (nt + no))
    assert np.allclose(rep, mine, atol=1e-12) and 0.2 < mean < 1.0


@pytest.mark.parametrize('upd', [{'channel_version': 1}, {'channel_version': 2, 'descriptor_size': 128}])
def test_channel_versions_against_reference(oracle, ref, upd):
    models, utils = ref
    cfg = dict(oracle.SHIPPED_MODEL_CONFIG); cfg.update(upd)
    net = models.MultiPoint(dict(cfg)).eval()
    spec = oracle.state_dict_spec(cfg)
    assert [k for k, _, _ in spec] == list(net.state_dict().keys())
    sd = oracle.make_weights(51, cfg)
    net.load_state_dict(sd)
    img = oracle.make_images(52, 2, 64, 96)
    with torch.no_grad():
        r = net({'image': img})
    o = oracle.forward(sd, img, cfg)
    assert (r['prob'] - o['prob']).abs().max().item() <= 1e-7 and (r['desc'] - o['desc']).abs().max().item() <= 1e-7
    from multipoint_amd.models import MultiPoint
    assert [(k, tuple(s)) for k, s, _ in MultiPoint(dict(cfg)).state_dict_spec()] == [(k, tuple(s)) for k, s, _ in spec]


# ---- homographic adaptation (reference multipoint/utils/homographies.py) ---------------------------------------
@pytest.fixture(scope='module')
def ref_homographies(ref):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import ref_shim
    return ref_shim.install_homographies()


def test_product_sample_homography_matches_reference(ref_homographies):
    """host logic of the product (no GPU): draw for draw the reference's sampler, on fresh seeds"""
    from multipoint_amd.utils.homographies import sample_homography, homography_adaptation_default_config
    RH = ref_homographies
    cases = [{}, {'allow_artifacts': False, 'max_angle': 0.4}, {'perspective': False, 'n_scales': 3},
             {'rotation': False, 'translation': False, 'scaling_amplitude': 0.5},
             dict(homography_adaptation_default_config['homographies'])]
    for seed in range(100, 112):
        for kw in cases:
            for shape in ((240, 320), (64, 64)):
                np.random.seed(seed); want = RH.sample_homography(np.array(shape), **kw)
                state = np.random.get_state()[1][:4].copy()
                np.random.seed(seed); got = sample_homography(np.array(shape), **kw)
                assert np.allclose(got, want, rtol=1e-9, atol=1e-11), (seed, kw)
                assert np.array_equal(np.random.get_state()[1][:4], state)       # same number of draws consumed


def test_oracle_homographic_adaptation_matches_reference_driver(oracle, ref, ref_homographies):
    import copy
    from oracle import ha_oracle as HA
    models, utils = ref
    RH = ref_homographies
    pristine = copy.deepcopy(RH.homography_adaptation_default_config)
    for pair, agg, hc in ((False, None, {'num': 3, 'erosion_radius': 2, 'mask_border': True, 'min_count': 2, 'filter_size': 0}),
                          (False, None, {'num': 3, 'erosion_radius': 4, 'mask_border': False, 'min_count': 0, 'filter_size': 5}),
                          (True, 'prod', {'num': 3, 'aggregation': 'prod', 'erosion_radius': 3, 'min_count': 3}),
                          (True, 'sum', {'num': 2, 'aggregation': 'sum', 'erosion_radius': 0, 'filter_size': 3})):
        cfg = {'multispectral': True, 'descriptor_size': 64} if pair else dict(oracle.SHIPPED_MODEL_CONFIG)
        sd = oracle.make_weights(17, cfg)
        net = models.MultiPoint(dict(cfg)).eval()
        net.load_state_dict(sd)
        img = oracle.make_images(41, 4 if pair else 2, 48, 56)
        RH.homography_adaptation_default_config.clear()          # the reference's dict_update writes into its default
        RH.homography_adaptation_default_config.update(copy.deepcopy(pristine))
        np.random.seed(23)
        with torch.no_grad():
            if pair:
                flags = [torch.ones(2, 1, dtype=torch.bool), torch.zeros(2, 1, dtype=torch.bool)]
                data = {'optical': {'image': img[:2], 'is_optical': flags[0]},
                        'thermal': {'image': img[2:], 'is_optical': flags[1]}}
                want = RH.homographic_adaptation_multispectral(data, net, copy.deepcopy(hc))
                fwd = lambda i, x: oracle.forward(sd, x, cfg, is_optical=flags[i])['prob']
                streams = [img[:2], img[2:]]
            else:
                want = RH.homographic_adaptation({'image': img}, net, copy.deepcopy(hc))
                fwd = lambda i, x: oracle.forward(sd, x, cfg)['prob']
                streams = [img]
        full = HA.full_config(hc)
        np.random.seed(23)
        homs = [RH.sample_homography(np.array([48, 56]), **full['homographies']) for _ in range(hc['num'] - 1)]
        got, _ = HA.homographic_adaptation(streams, fwd, hc, homs, aggregation=agg)
        assert (got - want).abs().max().item() <= 1e-6, hc
    RH.homography_adaptation_default_config.clear()
    RH.homography_adaptation_default_config.update(pristine)


def test_detector_metrics_against_reference(oracle, ref, monkeypatch):
    """The reference's compute_tp_fp_dist (evaluation.py:56-97, pure torch/numpy: runs here unmodified) and its
    compute_detector_metrics driver (:10-54, torchvision nms served by the oracle) against the oracle's restatement:
    pins the true-positive rule, n_gt, the distance list and the precision/recall bookkeeping."""
    models, utils = ref
    import multipoint.utils.evaluation as ev
    import multipoint.utils.utils as ru
    rng = np.random.default_rng(3)
    for case in range(6):
        H, W = 40, 56
        prob = np.zeros((H, W), np.float32)
        n = [60, 200, 5, 0, 400, 1][case]
        idx = rng.choice(H * W, n, replace=False)
        prob.flat[idx] = rng.permutation(n).astype(np.float32) / max(n, 1) * 0.9 + 0.01        # distinct scores
        km = np.zeros((H, W), bool)
        g = [40, 150, 0, 10, 300, 1][case]
        gi = np.concatenate([rng.choice(idx, min(g // 2, n), replace=False), rng.choice(H * W, g - min(g // 2, n), replace=False)]) if g else np.zeros(0, int)
        km.flat[gi.astype(int)] = True
        if case == 5:
            km[:] = False; km.flat[idx[0]] = True
        if n == 0:
            continue                                     # the reference's indexing fails on an empty prediction list
        r = ev.compute_tp_fp_dist(torch.from_numpy(prob), torch.from_numpy(km))
        o = oracle.compute_tp_fp_dist(prob, km)
        assert np.array_equal(r[0], o[0]) and np.array_equal(r[1], o[1]), case
        assert np.array_equal(r[2], o[2]) and r[3] == o[3]
        assert np.array_equal(np.asarray(r[4]), o[4]), case

    # the driver on two batches
    def nms(boxes, scores, iou):
        keep = oracle.nms_greedy(boxes.numpy().astype(np.float32), scores.numpy().astype(np.float32), float(iou))
        return torch.as_tensor(np.asarray(keep), dtype=torch.int64)

    def batched_nms(boxes, scores, idxs, iou):
        keep = []
        for i in torch.unique(idxs):
            sel = torch.nonzero(idxs == i)[:, 0]
            keep.append(sel[nms(boxes[sel], scores[sel], iou)])
        keep = torch.cat(keep) if keep else torch.zeros(0, dtype=torch.int64)
        return keep[torch.argsort(scores[keep], descending=True, stable=True)]
    monkeypatch.setattr(ru, 'nms', nms, raising=False); monkeypatch.setattr(ru, 'batched_nms', batched_nms, raising=False)
    cfg = dict(oracle.SHIPPED_MODEL_CONFIG)
    sd = oracle.make_weights(31, cfg)
    net = models.MultiPoint(dict(cfg)).eval(); net.load_state_dict(sd)
    H, W = 64, 96
    batches, mine = [], []
    for b in range(2):
        img = oracle.make_images(50 + b, 2, H, W)
        vm = torch.ones((2, 1, H, W), dtype=torch.bool); vm[:, :, :6] = False
        with torch.no_grad():
            pr = oracle.forward(sd, img, cfg)['prob'] * vm
        pn = oracle.box_nms(pr.numpy(), 4, 0.015)
        km = torch.zeros((2, H, W), dtype=torch.bool)
        for i in range(2):
            kept = np.argwhere(pn[i, 0] > 0.015)
            sel = kept[rng.choice(len(kept), len(kept) // 2, replace=False)]
            sel = np.clip(sel + rng.integers(-2, 3, sel.shape), 0, [H - 1, W - 1])       # jittered labels
            km[i, sel[:, 0], sel[:, 1]] = True
            mine.append(oracle.compute_tp_fp_dist(pn[i, 0], km[i].numpy()))
        batches.append({'image': img, 'valid_mask': vm, 'keypoints': km, 'is_optical': torch.ones(2, 1, dtype=torch.bool)})
    with torch.no_grad():
        precision, recall, prob, dist = ev.compute_detector_metrics(net, batches, torch.device('cpu'),
                                                                    {'nms': 4, 'detection_threshold': 0.015})
    tp = np.concatenate([m[0] for m in mine]); fp = np.concatenate([m[1] for m in mine])
    pp = np.concatenate([m[2] for m in mine]).astype(np.float64)
    p2, r2, prob2 = oracle.detector_precision_recall(tp, fp, pp, sum(m[3] for m in mine))
    assert 0 < tp.sum() < len(tp)
    assert np.array_equal(prob, prob2) and np.allclose(precision, p2, atol=0, rtol=0) and np.allclose(recall, r2, atol=0, rtol=0)
    assert np.array_equal(dist, np.concatenate([m[4] for m in mine]).astype(np.float64))
    assert abs(ev.compute_mAP(precision, recall) - ev.compute_mAP(p2, r2)) == 0


def test_threshold_matcher_against_reference(oracle, ref):
    """ThresholdMatcher (matching.py:74-99) is in-repo numpy: the oracle restatement is pinned exactly."""
    models, utils = ref
    rng = np.random.default_rng(8)
    d1 = rng.standard_normal((120, 64)).astype(np.float32); d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    d2 = np.concatenate([d1[:60] + 0.05 * rng.standard_normal((60, 64)).astype(np.float32),
                         rng.standard_normal((50, 64)).astype(np.float32)]); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    for thr in (0.4, 0.7, 1.3):
        m = utils.get_matches(d1, d2, 'thresholdmatcher', threshold=thr)
        q, t, d = oracle.threshold_match(d1, d2, thr)
        assert [x.queryIdx for x in m] == list(q) and [x.trainIdx for x in m] == list(t) and len(q) > 0
        assert np.array_equal(np.array([x.distance for x in m], np.float32), d)
    with pytest.raises(ValueError):
        utils.ThresholdMatcher(-0.5)
