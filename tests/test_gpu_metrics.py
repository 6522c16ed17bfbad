"""GPU (MI355X): mp_pair_metrics (the per-sample arithmetic of utils.compute_descriptor_metrics, reference
multipoint/utils/evaluation.py:259,287-328) against the oracle's restatement -- integer results, bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _random_homography(rng, H, W, strength):
    a = np.eye(3)
    a[:2, :2] += rng.normal(0, 0.05 * strength, (2, 2))
    a[:2, 2] += rng.normal(0, 6.0 * strength, 2)
    a[2, :2] += rng.normal(0, 2e-4 * strength, 2)
    return a.astype(np.float32)


def _make_results(rng, P, K, H, W, hs, fill):
    """Synthetic device-resident lists: thermal keypoints = optical ones pushed through the ground-truth homography
    plus noise (so that a realistic share of the correctness tests is near the threshold), random mutual matches."""
    from multipoint_amd.pipeline import PairResults
    kp = np.zeros((2 * P, K, 2), np.int32); cnt = np.zeros(2 * P, np.int32); midx = -np.ones((P, K), np.int32)
    for p in range(P):
        no = int(rng.integers(0, K + 1)) if fill == 'ragged' else K
        o = np.stack([rng.integers(0, H, no), rng.integers(0, W, no)], 1)
        gt = hs[p][1].astype(np.float64) @ np.linalg.inv(hs[p][0].astype(np.float64))
        xy1 = np.concatenate([o[:, ::-1], np.ones((no, 1))], 1) @ gt.T
        t = (xy1[:, :2] / xy1[:, 2:3])[:, ::-1] + rng.normal(0, 2.5, (no, 2))
        t = np.round(t).astype(np.int64)
        keep = rng.random(no) < 0.9
        t = t[keep]
        nt = min(len(t), K)
        t = np.clip(t[:nt], [-3, -3], [H + 2, W + 2])          # a few just outside the image
        kp[2 * p, :no] = o; kp[2 * p + 1, :nt] = t; cnt[2 * p] = no; cnt[2 * p + 1] = nt
        # mutual matches: a random partial injection optical -> thermal
        n_m = min(no, nt)
        if n_m:
            qs = rng.permutation(no)[:n_m]; ts = rng.permutation(nt)[:n_m]
            sel = rng.random(n_m) < 0.6
            midx[p, qs[sel]] = ts[sel]
    dev = 'cuda'
    res = PairResults(torch.from_numpy(kp).to(dev), None, torch.from_numpy(cnt).to(dev), None,
                      torch.from_numpy(midx).to(dev), torch.zeros((P, K), device=dev), None, H, W)
    return res, kp, cnt, midx


@pytest.mark.parametrize('P,K,H,W,strength,fill', [(3, 300, 240, 320, 1.0, 'ragged'), (2, 1000, 480, 640, 0.3, 'full'),
                                                   (2, 64, 64, 64, 0.0, 'ragged'), (1, 2000, 1024, 1280, 1.0, 'full')])
def test_pair_metrics_bit_exact(oracle, P, K, H, W, strength, fill):
    import multipoint_amd.utils as U
    rng = np.random.default_rng(K + P)
    hs = [(_random_homography(rng, H, W, strength), _random_homography(rng, H, W, strength)) for _ in range(P)]
    res, kp, cnt, midx = _make_results(rng, P, K, H, W, hs, fill)
    hom = U.ground_truth_homographies(np.stack([h[0] for h in hs]), np.stack([h[1] for h in hs]))
    thr = 4.0
    metrics, tp = U.pair_metrics(res, hom, thr)
    metrics = metrics.cpu().numpy(); tp = tp.cpu().numpy()
    for p in range(P):
        no, nt = cnt[2 * p], cnt[2 * p + 1]
        q = np.nonzero(midx[p, :no] >= 0)[0]; t = midx[p, q]
        ref = oracle.descriptor_metrics_pair(kp[2 * p, :no], kp[2 * p + 1, :nt], q, t, hs[p][0], hs[p][1], thr, H, W)
        assert metrics[p, 0] == ref['n_gt_optical'] and metrics[p, 1] == ref['n_gt_thermal']
        assert metrics[p, 2] == int(ref['tp_optical'].sum()) and metrics[p, 3] == int(ref['tp_thermal'].sum())
        assert metrics[p, 4] == ref['N_optical'] and metrics[p, 5] == ref['N_thermal']
        assert metrics[p, 6] == len(q)
        assert np.array_equal(tp[2 * p, q].astype(bool), ref['tp_optical'])
        assert np.array_equal(tp[2 * p + 1, t].astype(bool), ref['tp_thermal'])
        un = np.setdiff1d(np.arange(K), q); assert not tp[2 * p, un].any()


def test_compute_descriptor_metrics_end_to_end(oracle):
    """The reference driver's signature and result keys (evaluation.py:209-439) on a synthetic loader whose thermal
    image is the optical one: every mutual match of identical images is a true positive."""
    import multipoint_amd.models as M
    import multipoint_amd.utils as U
    cfg = oracle.SHIPPED_MODEL_CONFIG
    net = M.MultiPoint(dict(cfg)); net.load_state_dict(oracle.make_weights(0, cfg)); net.to('cuda'); net.eval()
    img = oracle.make_images(5, 2, 120, 160)
    batch = {'optical': {'image': img, 'valid_mask': torch.ones_like(img, dtype=torch.bool)},
             'thermal': {'image': img.clone(), 'valid_mask': torch.ones_like(img, dtype=torch.bool)}}
    pred = {'nms': 4, 'detection_threshold': 0.015, 'topk': 300, 'cpu_nms': False,
            'matching': {'method': 'bfmatcher', 'method_kwargs': {'crossCheck': True}, 'knn_matches': False}}
    out = U.compute_descriptor_metrics(net, [batch], torch.device('cuda'), pred, 4.0, 4.0)
    for k in ('tp_optical', 'fp_optical', 'distance_optical', 'recall_optical', 'precision_optical', 'nn_map_optical',
              'nn_map_thermal', 'nn_map', 'm_score_optical', 'm_score', 'pts_dist', 'h_correctness'):
        assert k in out
    assert len(out['tp_optical']) > 100 and out['tp_optical'].all() and out['tp_thermal'].all()
    assert abs(out['nn_map'] - 1.0) < 1e-9 and abs(out['m_score'] - 1.0) < 1e-9
    # identical images: the estimated homography is the identity to sub-pixel accuracy
    assert out['pts_dist'].shape == (2,) and out['pts_dist'].max() < 0.5 and out['h_correctness'] == 1.0


@pytest.mark.parametrize('P,K,H,W,strength', [(3, 300, 240, 320, 1.0), (2, 1000, 480, 640, 0.3), (2, 64, 64, 64, 0.0)])
def test_repeatability_counts_bit_exact(oracle, P, K, H, W, strength):
    import multipoint_amd.utils as U
    rng = np.random.default_rng(7 * K + P)
    hs = [(_random_homography(rng, H, W, strength), _random_homography(rng, H, W, strength)) for _ in range(P)]
    res, kp, cnt, _ = _make_results(rng, P, K, H, W, hs, 'ragged')
    kp = np.clip(kp, 0, [H - 1, W - 1]).astype(np.int32)                   # detector keypoints are always inside the image
    for thr in (3, 1.5):
        c = U.repeatability_counts(torch.from_numpy(kp).cuda(), res.kp_count, np.stack([h[0] for h in hs]),
                                   np.stack([h[1] for h in hs]), H, W, thr).cpu().numpy()
        for p in range(P):
            ref = oracle.repeatability_pair(kp[2 * p, :cnt[2 * p]], kp[2 * p + 1, :cnt[2 * p + 1]], hs[p][0], hs[p][1], H, W, thr)
            assert tuple(int(v) for v in c[p]) == ref


def test_compute_repeatability_end_to_end(oracle):
    import multipoint_amd.models as M
    import multipoint_amd.utils as U
    cfg = oracle.SHIPPED_MODEL_CONFIG
    net = M.MultiPoint(dict(cfg)); net.load_state_dict(oracle.make_weights(0, cfg)); net.to('cuda'); net.eval()
    img = oracle.make_images(5, 2, 120, 160)
    ones = torch.ones_like(img, dtype=torch.bool)
    batch = {'optical': {'image': img, 'valid_mask': ones}, 'thermal': {'image': img.clone(), 'valid_mask': ones}}
    config = {'prediction': {'nms': 4, 'detection_threshold': 0.015, 'topk': 300, 'cpu_nms': False}}
    mean, rep, n_o, n_t = U.compute_repeatability_multispectral(net, [batch], torch.device('cuda'), config, distance_thresh=3)
    assert mean == 1.0 and len(rep) == 2 and n_o == n_t and all(n > 50 for n in n_o)


def _planted(rng, P, K, H, W, outlier_frac):
    """Matched keypoint lists with a planted homography per pair: thermal = H(optical) + sub-pixel noise, rounded,
    a fraction of the matches replaced by random points."""
    from multipoint_amd.pipeline import PairResults
    kp = np.zeros((2 * P, K, 2), np.int32); cnt = np.zeros(2 * P, np.int32); midx = -np.ones((P, K), np.int32)
    planted = []
    for p in range(P):
        n = K if p % 2 == 0 else int(K * 0.6)
        hm = np.eye(3); hm[:2, :2] += rng.normal(0, 0.03, (2, 2)); hm[:2, 2] += rng.normal(0, 8, 2); hm[2, :2] += rng.normal(0, 5e-5, 2)
        o_xy = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.float64)
        xy1 = np.concatenate([o_xy, np.ones((n, 1))], 1) @ hm.T
        t_xy = np.round(xy1[:, :2] / xy1[:, 2:3] + rng.normal(0, 0.3, (n, 2)))
        bad = rng.random(n) < outlier_frac
        t_xy[bad] = np.stack([rng.integers(0, W, bad.sum()), rng.integers(0, H, bad.sum())], 1)
        perm = rng.permutation(n)                                     # thermal list in a different order
        kp[2 * p, :n] = o_xy[:, ::-1]; kp[2 * p + 1, perm] = t_xy[:, ::-1]
        cnt[2 * p] = cnt[2 * p + 1] = n
        matched = rng.random(n) < 0.9
        midx[p, :n][matched] = perm[matched]
        planted.append((hm, bad))
    res = PairResults(torch.from_numpy(kp).cuda(), None, torch.from_numpy(cnt).cuda(), None, torch.from_numpy(midx).cuda(),
                      torch.zeros((P, K), device='cuda'), None, H, W)
    return res, kp, cnt, midx, planted


@pytest.mark.parametrize('outlier_frac,K', [(0.0, 200), (0.3, 500), (0.6, 1000)])
def test_find_homography_matches_oracle_and_recovers_planted(oracle, outlier_frac, K):
    """mp_find_homography against the oracle's CPU restatement of the same algorithm (same winner / inlier set, H to
    rounding) and against the planted model (the property cv2.findHomography is used for)."""
    import multipoint_amd.utils as U
    rng = np.random.default_rng(int(outlier_frac * 10) + K)
    P, H, W, T, thr = 3, 480, 640, 512, 3.0
    res, kp, cnt, midx, planted = _planted(rng, P, K, H, W, outlier_frac)
    Hm, mask, nin = U.find_homography(res, thr, max_iters=T, seed=7)
    Hm = Hm.cpu().numpy(); mask = mask.cpu().numpy().astype(bool); nin = nin.cpu().numpy()
    corners = np.array([[0, 0, 1], [W, 0, 1], [0, H, 1], [W, H, 1]], dtype=np.float64)
    for p in range(P):
        n = cnt[2 * p]
        q = np.nonzero(midx[p, :n] >= 0)[0]; t = midx[p, q]
        a = kp[2 * p, q][:, ::-1]; b = kp[2 * p + 1, t][:, ::-1]
        Ho, mo = oracle.ransac_homography(a, b, thr, T, 7, p)
        assert Ho is not None and nin[p] == mo.sum() == mask[p].sum()
        assert np.array_equal(mask[p, q], mo) and not mask[p, np.setdiff1d(np.arange(K), q)].any()
        assert np.abs(Hm[p] - Ho).max() <= 1e-6 * max(1.0, np.abs(Ho).max())
        # planted model: every clean match is an inlier region-wise; corners land within a pixel or two
        hm, bad = planted[p]
        ce = corners @ Hm[p].T; cg = corners @ hm.T
        err = np.linalg.norm(ce[:, :2] / ce[:, 2:3] - cg[:, :2] / cg[:, 2:3], axis=1)
        assert err.max() < 2.0
        assert mask[p, q][~bad[q]].mean() > 0.95 and (outlier_frac == 0 or mask[p, q][bad[q]].mean() < 0.1)


def test_find_homography_degenerate_inputs():
    import multipoint_amd.utils as U
    from multipoint_amd.pipeline import PairResults
    K = 64
    kp = torch.zeros((4, K, 2), dtype=torch.int32, device='cuda')
    cnt = torch.tensor([3, 3, 10, 10], dtype=torch.int32, device='cuda')
    midx = -torch.ones((2, K), dtype=torch.int32, device='cuda')
    midx[0, :3] = torch.arange(3, dtype=torch.int32)                 # fewer than 4 matches
    kp[2, :10, 1] = torch.arange(10, dtype=torch.int32); kp[3, :10, 1] = torch.arange(10, dtype=torch.int32)   # collinear
    midx[1, :10] = torch.arange(10, dtype=torch.int32)
    res = PairResults(kp, None, cnt, None, midx, torch.zeros((2, K), device='cuda'), None, 64, 64)
    Hm, mask, nin = U.find_homography(res, 3.0, max_iters=256)
    assert nin.tolist() == [0, 0] and not mask.any() and (Hm == 0).all()
    with pytest.raises(ValueError):
        U.find_homography(res, -1.0)


# ----------------------------------------------------------------------------------------------------------------------
# single-image detector metrics (mp_detector_metrics; multipoint/utils/evaluation.py:10-97)
# ----------------------------------------------------------------------------------------------------------------------
def _check_records(got, want):
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(got[2], want[2]) and got[3] == want[3]
    assert got[4].dtype == np.float32 and np.array_equal(got[4], want[4])


def test_detector_metrics_golden(golden_dir):
    """GPU vs the outputs of the reference's compute_tp_fp_dist (tests/golden/detector_metrics.npz)."""
    import multipoint_amd.utils as U
    g = np.load(os.path.join(golden_dir, 'detector_metrics.npz'))
    prob = torch.from_numpy(np.stack([g['prob_%d' % i] for i in range(4)])).to(DEV)
    km = torch.from_numpy(np.stack([g['keypoints_%d' % i] for i in range(4)])).to(DEV)
    recs = U.detector_records(prob, km)
    for i, r in enumerate(recs):
        _check_records(r, (g['tp_%d' % i], ~g['tp_%d' % i], g['sorted_prob_%d' % i], int(g['n_gt_%d' % i]), g['dist_%d' % i]))
    one = U.compute_tp_fp_dist(prob[1].cpu(), km[1].cpu())
    _check_records(one, recs[1])
    lst = U.compute_tp_fp_dist(prob[1], torch.nonzero(km[1]))                 # (N,2) label list instead of a map
    _check_records(lst, recs[1])


@pytest.mark.parametrize('H,W,density', [(37, 53, 0.05), (120, 160, 0.3), (64, 64, 1.0)])
def test_detector_metrics_vs_oracle(oracle, H, W, density):
    """random maps incl. tied scores, dense predictions (nms = 0 case), clustered labels, empty images."""
    import multipoint_amd.utils as U
    rng = np.random.default_rng(H * W)
    B = 4
    prob = (rng.random((B, H, W)) * (rng.random((B, H, W)) < density)).astype(np.float32)
    prob[1] = np.round(prob[1] * 8) / 8                                       # many exact ties
    prob[3] = 0                                                               # no prediction at all
    km = rng.random((B, H, W)) < 0.02
    km[0, 10:14, 10:14] = True                                                # clustered labels
    km[2] = False                                                             # no label at all
    for thr in (2.0, 1.0, 1.5, 0.0, 2.9):
        recs = U.detector_records(torch.from_numpy(prob).to(DEV), torch.from_numpy(km).to(DEV), 1e-4, thr)
        for b in range(B):
            _check_records(recs[b], oracle.compute_tp_fp_dist(prob[b], km[b], 1e-4, thr))
    assert recs[3][0].size == 0 and recs[2][0].sum() == 0
    with pytest.raises(ValueError):
        U.detector_records(torch.from_numpy(prob).to(DEV), torch.from_numpy(km).to(DEV), 1e-4, 3.0)
    with pytest.raises(ValueError):
        U.detector_records(torch.from_numpy(prob).to(DEV), torch.from_numpy(km[:, :-1]).to(DEV))


def test_detector_metrics_full_size_properties():
    """BASELINE size (64 maps 480x640), dense predictions: every label is claimed by at most one true positive, a
    true positive exists exactly for the labels with a prediction in range, ranks are non-increasing."""
    import multipoint_amd.utils as U
    g = torch.Generator(device='cpu').manual_seed(5)
    B, H, W = 64, 480, 640
    prob = torch.rand((B, H, W), generator=g) * (torch.rand((B, H, W), generator=g) < 0.02)
    km = torch.rand((B, H, W), generator=g) < 0.004
    recs = U.detector_records(prob.to(DEV), km.to(DEV))
    tot_tp = 0
    for b in (0, 17, 63):
        tp, fp, p, n_gt, dist = recs[b]
        assert n_gt == int(km[b].sum()) and len(p) == int((prob[b] > 1e-4).sum())
        assert np.all(np.diff(p) <= 0) and np.array_equal(fp, ~tp)
        # labels with at least one prediction within 2 px whose FIRST label is that label
        assert tp.sum() <= n_gt and set(np.unique(dist)) <= {np.float32(0), np.float32(1), np.float32(2 ** .5), np.float32(2)}
        tot_tp += tp.sum()
    assert tot_tp > 0


def test_compute_detector_metrics_driver(oracle):
    """The driver (forward + valid mask + NMS on the GPU, mp_detector_metrics, numpy tail) vs the oracle pipeline."""
    import multipoint_amd.models as models
    import multipoint_amd.utils as U
    cfg = dict(oracle.SHIPPED_MODEL_CONFIG)
    sd = oracle.make_weights(31, cfg)
    net = models.MultiPoint(cfg); net.load_state_dict(sd); net.to(DEV); net.eval()
    rng = np.random.default_rng(9)
    H, W = 64, 96
    batches, mine = [], []
    for b in range(2):
        img = oracle.make_images(50 + b, 2, H, W)
        vm = torch.ones((2, 1, H, W), dtype=torch.bool); vm[:, :, :6] = False
        pn = U.box_nms(net({'image': img.to(DEV)})['prob'], 4, 0.015, valid_mask=vm.to(DEV)).cpu().numpy()
        km = torch.zeros((2, H, W), dtype=torch.bool)
        for i in range(2):
            kept = np.argwhere(pn[i, 0] > 0.015)
            sel = kept[rng.choice(len(kept), len(kept) // 2, replace=False)]
            sel = np.clip(sel + rng.integers(-2, 3, sel.shape), 0, [H - 1, W - 1])
            km[i, sel[:, 0], sel[:, 1]] = True
            mine.append(oracle.compute_tp_fp_dist(pn[i, 0], km[i].numpy()))
        batches.append({'image': img, 'valid_mask': vm, 'keypoints': km, 'is_optical': torch.ones(2, 1, dtype=torch.bool)})
    precision, recall, prob, dist = U.compute_detector_metrics(net, batches, DEV, {'nms': 4, 'detection_threshold': 0.015})
    tp = np.concatenate([m[0] for m in mine]); fp = np.concatenate([m[1] for m in mine])
    p2, r2, prob2 = oracle.detector_precision_recall(tp, fp, np.concatenate([m[2] for m in mine]).astype(np.float64),
                                                     sum(m[3] for m in mine))
    assert 0 < tp.sum() < len(tp)
    assert np.array_equal(prob, prob2) and np.array_equal(precision, p2) and np.array_equal(recall, r2)
    assert np.array_equal(dist, np.concatenate([m[4] for m in mine]))
    assert 0.0 < U.compute_mAP(precision, recall) <= 1.0
    # nms = 0: the raw map times the valid mask
    pr, rc, _, _ = U.compute_detector_metrics(net, batches[:1], DEV, {'nms': 0, 'detection_threshold': 0.015})
    assert len(pr) == len(rc) and pr[0] >= pr[-1]


def test_descriptor_metrics_matcher_config(oracle):
    """evaluation.py:273-282 hard-codes BFMatcher(crossCheck) for the NN-mAP / M-score matches; config['matching'] only
    selects the matches the homography is estimated from (:332-336).  So every matcher config gives the same metrics,
    and the non-default ones (nnmatcher, ratio test, thresholdmatcher) run through get_matches per pair."""
    import random
    import multipoint_amd.models as models
    import multipoint_amd.utils as U
    from multipoint_amd.datasets import SyntheticPairs
    from oracle import ha_oracle as HA
    cfg = dict(oracle.SHIPPED_MODEL_CONFIG)
    net = models.MultiPoint(cfg); net.load_state_dict(oracle.make_weights(0, cfg)); net.to(DEV); net.eval()
    hc = dict(HA.PREDICTION_AUGMENTATION)
    hc['params'] = dict(hc['params'], perspective_amplitude_x=0.02, perspective_amplitude_y=0.02, max_angle=0.05,
                        scaling_amplitude=0.02)                        # mild warps: the matches support a homography
    results = {}
    for name, matching in (('default', {'method': 'bfmatcher', 'method_kwargs': {'crossCheck': True}, 'knn_matches': False}),
                           ('nn', {'method': 'nnmatcher', 'method_kwargs': {'threshold': 0.9}, 'knn_matches': False}),
                           ('ratio', {'method': 'bfmatcher', 'method_kwargs': {}, 'knn_matches': True}),
                           ('thr', {'method': 'thresholdmatcher', 'method_kwargs': {'threshold': 0.5}, 'knn_matches': False})):
        ds = SyntheticPairs({'num_samples': 4, 'height': 120, 'width': 160, 'augmentation': {'homographic': hc}})

        class SamePair(torch.utils.data.Dataset):                     # thermal := optical content, so matches exist
            def __len__(self): return len(ds)
            def __getitem__(self, i):
                random.seed(10 + i); np.random.seed(20 + i)
                s = ds[i]
                base = torch.from_numpy(SyntheticPairs.make_pair(0, i, 120, 160)[0])
                if torch.equal(s['optical']['homography'], torch.eye(3)):
                    s['optical']['image'] = base
                else:
                    s['thermal']['image'] = base
                return s
        loader = torch.utils.data.DataLoader(SamePair(), batch_size=2, shuffle=False, num_workers=0)
        pred = {'nms': 4, 'detection_threshold': 0.015, 'topk': 300, 'cpu_nms': True, 'reprojection_threshold': 3,
                'matching': matching}
        results[name] = U.compute_descriptor_metrics(net, loader, DEV, pred, 4, 3)
    ref = results['default']
    for name, r in results.items():
        assert r['nn_map'] == ref['nn_map'] and r['m_score'] == ref['m_score'], name
        assert np.array_equal(r['tp_optical'], ref['tp_optical'])
        assert len(r['pts_dist']) == 4 and r['h_correctness'] is not None
    assert (np.asarray(ref['pts_dist']) < 999).any()                   # at least one homography was estimated


@pytest.mark.parametrize('outlier_frac,K', [(0.1, 300), (0.5, 800), (0.75, 1000)])
def test_find_homography_against_opencv_semantics(outlier_frac, K):
    """mp_find_homography against the INDEPENDENT restatement of cv2.findHomography(..., cv2.RANSAC, thr) (oracle/
    cv_homography.py: adaptive iteration bound at confidence 0.995, refit on the consensus set, Levenberg-Marquardt polish)
    -- the product's kernel is a different design (all hypotheses in parallel, DLT refit, no LM), so the comparison is the
    one that matters to the reference's metrics (evaluation.py:330-356): corner error against the planted model, agreement
    of the h_correctness decision (mean corner distance < 3 px), and overlap of the inlier sets."""
    import multipoint_amd.utils as U
    from oracle import cv_homography as CV
    rng = np.random.default_rng(int(outlier_frac * 100) + K)
    P, H, W, thr = 8, 480, 640, 3.0
    res, kp, cnt, midx, planted = _planted(rng, P, K, H, W, outlier_frac)
    Hm, mask, nin = U.find_homography(res, thr, max_iters=2000, seed=3)
    Hm = Hm.cpu().numpy(); mask = mask.cpu().numpy().astype(bool)
    corners = np.array([[0, 0, 1], [W, 0, 1], [0, H, 1], [W, H, 1]], dtype=np.float64)

    def corner_err(Ha, Hb):
        a = corners @ Ha.T; b = corners @ Hb.T
        return np.linalg.norm(a[:, :2] / a[:, 2:3] - b[:, :2] / b[:, 2:3], axis=1).mean()
    worst_delta, ious, agree = 0.0, [], 0
    for p in range(P):
        n = cnt[2 * p]
        q = np.nonzero(midx[p, :n] >= 0)[0]; t = midx[p, q]
        a = kp[2 * p, q][:, ::-1].astype(np.float64); b = kp[2 * p + 1, t][:, ::-1].astype(np.float64)
        Hcv, mcv = CV.find_homography_ransac(a, b, thr, seed=p)
        assert Hcv is not None and Hm[p].any()
        hm, bad = planted[p]
        e_gpu, e_cv = corner_err(Hm[p], hm), corner_err(Hcv, hm)
        worst_delta = max(worst_delta, abs(e_gpu - e_cv))
        agree += (e_gpu < 3.0) == (e_cv < 3.0)
        mg = mask[p, q]; mc = mcv.astype(bool)
        ious.append((mg & mc).sum() / max((mg | mc).sum(), 1))
        # both recover the planted model to well under the 3 px correctness threshold
        assert e_gpu < 1.5 and e_cv < 1.5, (p, e_gpu, e_cv)
    print('\n[findHomography vs OpenCV semantics, outliers %.2f] max |corner error delta| %.3f px, min inlier IoU %.3f'
          % (outlier_frac, worst_delta, min(ious)))
    assert agree == P                       # the same h_correctness on every pair
    assert worst_delta < 0.75               # LM polish vs DLT refit: sub-pixel differences only
    # OpenCV's mask is the consensus set of its best 4-POINT sample (not re-evaluated after the refit); the kernel's is the
    # consensus set of the best of 2000 samples: at 75 % outliers they overlap by 0.84, at <= 50 % by 0.99
    assert min(ious) > 0.8
