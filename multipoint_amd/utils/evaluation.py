"""Host-side mirror of the descriptor-metric driver of the reference (multipoint/utils/evaluation.py:209-439,
`compute_descriptor_metrics`, the `-e` mode of predict_align_image_pair.py:69-73): forward, NMS/top-k, descriptor
sampling and mutual-NN matching run through PairPipeline, and the per-sample arithmetic (keypoint warping by the
ground-truth homography, the N x M correctness test, true positives of the matches, matching score) runs on the GPU
behind mp_pair_metrics.  Only the final precision/recall bookkeeping over the concatenated match lists is numpy, as
in the reference (:360-419).

The RANSAC homography estimate (:330-349, cv2.findHomography) runs on the GPU as well (mp_find_homography: same
algorithm family, not OpenCV's RNG, so estimates agree with OpenCV's only to the reprojection tolerance); the
4-corner error derived from it (:351-356) is four points of numpy per pair."""
import ctypes

import numpy as np
import torch

from .. import _lib


def div0(a, b):
    # evaluation.py:202-207
    with np.errstate(divide='ignore', invalid='ignore'):
        c = np.true_divide(a, b)
        idx = ~np.isfinite(c)
        c[idx] = np.where(a[idx] == 0, 1, 0)
    return c


def compute_mAP(precision, recall):
    # evaluation.py:99-103
    return np.sum(precision[1:] * (recall[1:] - recall[:-1]))


def ground_truth_homographies(h_optical, h_thermal):
    """evaluation.py:259,288: gt = h_t @ inv(h_o) and its inverse, fp32 like the reference's torch ops.
    h_* : (P,3,3) tensors/arrays.  Returns a (2P,9) float64 tensor: slot 2p = gt, 2p+1 = inv(gt)."""
    ho = torch.as_tensor(h_optical, dtype=torch.float32).cpu()
    ht = torch.as_tensor(h_thermal, dtype=torch.float32).cpu()
    gt = torch.matmul(ht, torch.linalg.inv(ho))
    gti = torch.linalg.inv(gt)
    return torch.stack([gt, gti], dim=1).reshape(-1, 9).to(torch.float64)


def pair_metrics(res, homography, threshold_keypoints):
    """GPU arithmetic of evaluation.py:287-328 for the pairs of a PairResults.
    homography: (2P,9) float64 from ground_truth_homographies().
    Returns (metrics [P,8] int32 device tensor, tp [2P,K] uint8 device tensor); see include/multipoint_hip.h."""
    res.wait()
    dev = res.kp_yx.device
    P = res.num_pairs
    K = res.kp_yx.shape[1]
    hom = torch.as_tensor(homography, dtype=torch.float64).reshape(2 * P, 9).to(dev).contiguous()
    metrics = torch.empty((P, 8), dtype=torch.int32, device=dev)
    tp = torch.empty((2 * P, K), dtype=torch.uint8, device=dev)
    h = _lib.get_handle(dev)
    with torch.cuda.device(dev):
        h.check(h.lib.mp_pair_metrics(h.ptr, _lib.ptr(res.kp_yx.contiguous()), _lib.ptr(res.kp_count.contiguous()),
                                      _lib.ptr(res.match_idx.contiguous()), _lib.ptr(hom), P, K, int(res.H), int(res.W),
                                      float(threshold_keypoints), _lib.ptr(metrics), _lib.ptr(tp),
                                      _lib.stream_ptr(dev)))
    return metrics, tp


def find_homography(res, reproj_threshold=3.0, max_iters=2000, seed=0):
    """Batched cv2.findHomography(optical_pts, thermal_pts, cv2.RANSAC, reproj_threshold) for the pairs of a PairResults
    (predict_align_image_pair.py:205-216).  NOT OpenCV's algorithm bit for bit: every pair evaluates `max_iters` 4-point
    hypotheses in parallel (no confidence-driven early stop) and refits the best consensus set by the normalised DLT
    (OpenCV: adaptive iteration bound at confidence 0.995, refit, then a Levenberg-Marquardt polish).  Measured against an
    independent restatement of OpenCV 4.2's published algorithm (test infrastructure,
    tests/test_gpu_metrics.py::test_find_homography_against_opencv_semantics; planted homographies, 10-75 % outliers):
    mean corner distance differs by <= 0.11 px, the h_correctness decision (< 3 px) agrees on every pair, the inlier
    masks overlap by IoU >= 0.99 up to 50 % outliers (0.84 at 75 %: OpenCV's mask is the consensus set of its best
    4-point sample).  Returns (H [P,3,3] float64 mapping optical (x,y,1) to thermal -- all zeros
    where the reference would get None --, inlier mask [P,K] uint8 per optical keypoint, n_inliers [P] int32)."""
    res.wait()
    dev = res.kp_yx.device
    P, K = res.num_pairs, res.kp_yx.shape[1]
    Hm = torch.empty((P, 3, 3), dtype=torch.float64, device=dev)
    mask = torch.empty((P, K), dtype=torch.uint8, device=dev)
    nin = torch.empty((P,), dtype=torch.int32, device=dev)
    h = _lib.get_handle(dev)
    with torch.cuda.device(dev):
        h.check(h.lib.mp_find_homography(h.ptr, _lib.ptr(res.kp_yx.contiguous()), _lib.ptr(res.kp_count.contiguous()),
                                         _lib.ptr(res.match_idx.contiguous()), P, K, float(reproj_threshold), int(max_iters),
                                         int(seed), _lib.ptr(Hm), _lib.ptr(mask), _lib.ptr(nin), _lib.stream_ptr(dev)))
    return Hm, mask, nin


MAX_RANSAC_MATCHES = 3200        # mp_find_homography keeps a pair's correspondences in LDS


def find_homography_points(optical_pts, thermal_pts, reproj_threshold=3.0, max_iters=2000, seed=0, device=None):
    """cv2.findHomography(optical_pts, thermal_pts, cv2.RANSAC, reproj_threshold) for ONE set of corresponding (x, y)
    points (integer pixel positions, as keypoints are).  Returns (H 3x3 float64 numpy or None, mask (N,) uint8)."""
    from ..pipeline import PairResults
    a = np.asarray(optical_pts).reshape(-1, 2); b = np.asarray(thermal_pts).reshape(-1, 2)
    n = len(a)
    if n < 4:
        return None, np.zeros(n, np.uint8)
    if n > MAX_RANSAC_MATCHES:
        raise ValueError('find_homography_points: at most %d correspondences per call (the pair\'s matches live in LDS); '
                         'got %d -- keep the closest ones' % (MAX_RANSAC_MATCHES, n))
    dev = _lib.require_cuda(device)
    kp = torch.zeros((2, n, 2), dtype=torch.int32)
    kp[0] = torch.from_numpy(np.ascontiguousarray(a[:, ::-1]).astype(np.int32)); kp[1] = torch.from_numpy(np.ascontiguousarray(b[:, ::-1]).astype(np.int32))
    res = PairResults(kp.to(dev), None, torch.tensor([n, n], dtype=torch.int32, device=dev), None,
                      torch.arange(n, dtype=torch.int32, device=dev).reshape(1, n), None, None, 0, 0)
    Hm, mask, nin = find_homography(res, reproj_threshold, max_iters, seed)
    if int(nin[0]) < 4:
        return None, np.zeros(n, np.uint8)
    return Hm[0].cpu().numpy(), mask[0].cpu().numpy()


def _warp_yx(pts_yx, hmat):
    """warp_keypoints(..., np.float) for a handful of points (homographies.py:331-346)."""
    p = np.asarray(pts_yx, dtype=np.float64)
    m = np.asarray(hmat, dtype=np.float64).reshape(3, 3)
    xy1 = np.concatenate([p[:, ::-1], np.ones((len(p), 1))], 1) @ m.T
    return (xy1[:, :2] / xy1[:, 2:3])[:, ::-1]


def compute_descriptor_metrics(net, dataloader, device, config, threshold_keypoints, threshold_warp=None):
    """Same signature and result keys as the reference function (evaluation.py:209)."""
    from ..pipeline import PairPipeline
    from .utils import data_to_device
    from .matching import get_matches
    # The metrics are ALWAYS computed on cv2.BFMatcher(crossCheck=True) matches, whatever the config says -- the
    # reference hard-codes that matcher for matches_optical / matches_thermal (evaluation.py:273-282) and uses
    # config['matching'] only for the matches the homography is estimated from (:332-336).
    metric_matching = {'method': 'bfmatcher', 'method_kwargs': {'crossCheck': True}, 'knn_matches': False}
    mcfg = config.get('matching', metric_matching)
    same_matcher = (mcfg.get('method', 'bfmatcher') == 'bfmatcher' and not mcfg.get('knn_matches', False)
                    and dict(mcfg.get('method_kwargs', {})) == {'crossCheck': True})
    pipe = PairPipeline(net, dict(config, matching=metric_matching))
    tp_o, tp_t, dist_o, dist_t, ms_o, ms_t, pts_dist = [], [], [], [], [], [], []
    n_gt_o = n_gt_t = 0
    for data in dataloader:
        data = data_to_device(data, device)
        opt, th = data['optical'], data['thermal']
        B = opt['image'].shape[0]
        eye = torch.eye(3, dtype=torch.float32).repeat(B, 1, 1)
        ho = opt.get('homography', eye)
        ht = th.get('homography', eye)
        res = pipe(opt['image'], th['image'], opt.get('valid_mask'), th.get('valid_mask'))
        gth = ground_truth_homographies(ho, ht)
        metrics, tp = pair_metrics(res, gth, threshold_keypoints)
        pipe.check_converged()
        if same_matcher:
            h_est, _, n_in = find_homography(res, config.get('reprojection_threshold', 3))
            h_est = h_est.cpu().numpy(); n_in = n_in.cpu().numpy()
        else:
            # another matcher for the homography estimate: per pair through get_matches (GPU), like the reference
            h_est = np.zeros((B, 3, 3)); n_in = np.zeros(B, dtype=np.int64)
            kp_all = res.kp_yx.cpu().numpy(); cnt_all = res.kp_count.cpu().numpy()
            Kc = kp_all.shape[1]
            for p in range(B):
                no, nt = min(int(cnt_all[2 * p]), Kc), min(int(cnt_all[2 * p + 1]), Kc)
                if no == 0 or nt == 0:
                    continue
                matches = get_matches(res.desc[2 * p, :no], res.desc[2 * p + 1, :nt], mcfg['method'],
                                      mcfg.get('knn_matches', False), **mcfg.get('method_kwargs', {}))
                if len(matches) < 4:
                    continue
                if len(matches) > MAX_RANSAC_MATCHES:           # one-to-many matchers (thresholdmatcher): closest first
                    matches = sorted(matches, key=lambda mm: mm.distance)[:MAX_RANSAC_MATCHES]
                opts = np.array([kp_all[2 * p, mm.queryIdx][::-1] for mm in matches])
                tpts = np.array([kp_all[2 * p + 1, mm.trainIdx][::-1] for mm in matches])
                hp, mask = find_homography_points(opts, tpts, config.get('reprojection_threshold', 3), device=device)
                if hp is not None:
                    h_est[p] = hp; n_in[p] = int(mask.sum())
        H_o, W_o = opt['image'].shape[2:]
        m = metrics.cpu().numpy(); tp = tp.cpu().numpy()
        midx = res.match_idx.cpu().numpy(); mdist = res.match_dist.cpu().numpy()
        cnt = res.kp_count.cpu().numpy()
        K = midx.shape[1]
        for p in range(B):
            no = min(int(cnt[2 * p]), K)
            q = np.nonzero(midx[p, :no] >= 0)[0]
            # matches_optical (query = optical) and matches_thermal (query = thermal) are the same mutual pairs
            # (evaluation.py:273-282); their order does not matter, everything is re-sorted by distance below
            tp_o.append(tp[2 * p, q].astype(bool)); dist_o.append(mdist[p, q])
            tp_t.append(tp[2 * p + 1, midx[p, q]].astype(bool)); dist_t.append(mdist[p, q])
            n_gt_o += int(m[p, 0]); n_gt_t += int(m[p, 1])
            ms_o.append(float(m[p, 2]) / m[p, 4] if m[p, 4] > 0 else 0.0)
            ms_t.append(float(m[p, 3]) / m[p, 5] if m[p, 5] > 0 else 0.0)
            # homography correctness (:351-356; the reference's corner list, including its (H_o, H_o) last point)
            if n_in[p] >= 4:
                pts = np.array([[0, 0], [H_o, 0], [0, W_o], [H_o, H_o]])
                gt = gth[2 * p].numpy().reshape(3, 3)
                pts_dist.append(np.linalg.norm(_warp_yx(pts, h_est[p]) - _warp_yx(pts, gt), axis=1).sum() / 4)
            else:
                pts_dist.append(999.0)
    out = summarize_descriptor_metrics(np.concatenate(tp_o) if tp_o else np.zeros(0, bool),
                                        np.concatenate(dist_o) if dist_o else np.zeros(0, np.float32),
                                        np.concatenate(tp_t) if tp_t else np.zeros(0, bool),
                                        np.concatenate(dist_t) if dist_t else np.zeros(0, np.float32),
                                        n_gt_o, n_gt_t, np.array(ms_o), np.array(ms_t))
    pts_dist = np.array(pts_dist)
    out['pts_dist'] = pts_dist
    out['average_h_error'] = pts_dist.mean() if len(pts_dist) else None
    out['h_correctness'] = (pts_dist < threshold_warp).sum() / len(pts_dist) if len(pts_dist) and threshold_warp is not None else None
    return out


def summarize_descriptor_metrics(tp_optical, distance_optical, tp_thermal, distance_thermal, n_gt_optical,
                                 n_gt_thermal, m_score_optical, m_score_thermal):
    """evaluation.py:360-439 (precision / recall / NN-mAP / M-score bookkeeping), verbatim arithmetic."""
    sort_o = np.argsort(distance_optical, kind='stable')
    tp_optical = tp_optical[sort_o]; fp_optical = np.logical_not(tp_optical); distance_optical = distance_optical[sort_o]
    sort_t = np.argsort(distance_thermal, kind='stable')
    tp_thermal = tp_thermal[sort_t]; fp_thermal = np.logical_not(tp_thermal); distance_thermal = distance_thermal[sort_t]
    tpo, tpt = np.cumsum(tp_optical), np.cumsum(tp_thermal)
    fpo, fpt = np.cumsum(fp_optical), np.cumsum(fp_thermal)
    recall_optical = div0(tpo, n_gt_optical); recall_thermal = div0(tpt, n_gt_thermal)
    precision_optical = div0(tpo, tpo + fpo); precision_thermal = div0(tpt, tpt + fpt)
    recall_optical = np.concatenate([[0], recall_optical, [1]])
    precision_optical = np.concatenate([[0], precision_optical, [0]])
    precision_optical = np.maximum.accumulate(precision_optical[::-1])[::-1]
    recall_thermal = np.concatenate([[0], recall_thermal, [1]])
    precision_thermal = np.concatenate([[0], precision_thermal, [0]])
    precision_thermal = np.maximum.accumulate(precision_thermal[::-1])[::-1]
    nn_map_optical = compute_mAP(precision_optical, recall_optical)
    nn_map_thermal = compute_mAP(precision_thermal, recall_thermal)
    m_score = (m_score_optical.mean() + m_score_thermal.mean()) * 0.5 if len(m_score_optical) else 0.0
    return {
        'tp_optical': tp_optical, 'tp_thermal': tp_thermal, 'fp_optical': fp_optical, 'fp_thermal': fp_thermal,
        'distance_optical': distance_optical, 'distance_thermal': distance_thermal,
        'recall_optical': recall_optical, 'recall_thermal': recall_thermal,
        'precision_optical': precision_optical, 'precision_thermal': precision_thermal,
        'nn_map_optical': nn_map_optical, 'nn_map_thermal': nn_map_thermal,
        'nn_map': (nn_map_optical + nn_map_thermal) * 0.5,
        'm_score_optical': m_score_optical, 'm_score_thermal': m_score_thermal, 'm_score': m_score,
        'pts_dist': None, 'average_h_error': None, 'h_correctness': None,
    }


def repeatability_counts(kp_yx, kp_count, h_optical, h_thermal, H, W, distance_thresh):
    """GPU arithmetic of evaluation.py:156-199 on interleaved keypoint lists (slot 2p optical, 2p+1 thermal).
    Returns a [P,4] int32 device tensor: count1, count2, N_thermal, N_optical."""
    dev = kp_yx.device
    P = kp_yx.shape[0] // 2
    K = kp_yx.shape[1]
    ho = torch.as_tensor(h_optical, dtype=torch.float32).cpu().reshape(P, 3, 3)
    ht = torch.as_tensor(h_thermal, dtype=torch.float32).cpu().reshape(P, 3, 3)
    hoi, hti = torch.linalg.inv(ho), torch.linalg.inv(ht)          # fp32 like h.squeeze().inverse() (:168,173)
    hom = torch.stack([torch.stack([hoi, ht], 1), torch.stack([hti, ho], 1)], 1)       # [P][slot][warp][3][3]
    hom = hom.reshape(2 * P, 18).to(torch.float64).to(dev).contiguous()
    counts = torch.empty((P, 4), dtype=torch.int32, device=dev)
    h = _lib.get_handle(dev)
    with torch.cuda.device(dev):
        h.check(h.lib.mp_repeatability(h.ptr, _lib.ptr(kp_yx.contiguous()), _lib.ptr(kp_count.contiguous()), _lib.ptr(hom),
                                       P, K, int(H), int(W), float(distance_thresh), _lib.ptr(counts),
                                       _lib.stream_ptr(dev)))
    return counts


def compute_repeatability_multispectral(net, dataloader, device, config, distance_thresh=3, verbose=False):
    """Same signature and return value as the reference (evaluation.py:105-200):
    (mean repeatability, per-sample list, n_kp_optical, n_kp_thermal).  `config` is the whole yaml dict."""
    from .utils import box_nms_tie_robust, data_to_device, extract_keypoints
    pred = config['prediction']
    thr = pred['detection_threshold']
    cap = pred['topk'] if pred.get('topk', 0) > 0 else 4096
    repeatability, n_kp_optical, n_kp_thermal = [], [], []
    for data in dataloader:
        B = data['optical']['image'].shape[0]
        eye = torch.eye(3).repeat(B, 1, 1)
        ho = data['optical'].get('homography', eye)
        ht = data['thermal'].get('homography', eye)
        data = data_to_device(data, device)
        img = torch.stack([data['optical']['image'], data['thermal']['image']], 1).flatten(0, 1)     # interleaved
        mask = torch.stack([data['optical']['valid_mask'], data['thermal']['valid_mask']], 1).flatten(0, 1)
        flags = (torch.arange(2 * B) % 2 == 0).reshape(-1, 1)
        fwd_in = {'image': img, 'is_optical': flags}
        fwd = net(fwd_in)
        prob = fwd['prob']
        if pred['nms'] > 0:      # (top-k tie guard: flagged images are re-evaluated with the tie-exact algorithm)
            prob = box_nms_tie_robust(net, fwd_in, fwd, pred['nms'], thr, keep_top_k=pred['topk'], on_cpu=pred.get('cpu_nms', False))
        # keypoints: nonzero((prob > thr) * mask)  (:156-157) -- the mask is applied AFTER the NMS here
        kp, _, cnt = extract_keypoints(prob, thr, cap, valid_mask=mask)
        H, W = prob.shape[2:]
        c = repeatability_counts(kp, cnt, ho, ht, H, W, distance_thresh).cpu().numpy()
        cnt = cnt.cpu().numpy()
        if (cnt > cap).any():
            raise RuntimeError('more than %d keypoints in an image: set prediction.topk' % cap)
        for p in range(B):
            n_kp_optical.append(int(cnt[2 * p])); n_kp_thermal.append(int(cnt[2 * p + 1]))
            if c[p, 2] + c[p, 3] > 0:
                repeatability.append((c[p, 0] + c[p, 1]) / (c[p, 2] + c[p, 3]))
                if verbose:
                    print('repeatability: %f' % repeatability[-1])
    return np.mean(repeatability), repeatability, n_kp_optical, n_kp_thermal


# ----------------------------------------------------------------------------------------------------------------------
# single-image detector metrics (evaluation.py:10-103; predict_keypoints.py:88-104)
# ----------------------------------------------------------------------------------------------------------------------
_WINDOW_DIST = np.sqrt(np.add.outer((np.arange(5) - 2) ** 2, (np.arange(5) - 2) ** 2).astype(np.float32)).reshape(-1)


def detector_records(prob, keypoint_map, zero_threshold=1e-4, distance_thresh=2.0):
    """GPU arithmetic of compute_tp_fp_dist for a batch: prob (B,H,W) / (B,1,H,W) fp32, keypoint_map (B,H,W) bool.
    Returns per image the tuple of the reference (tp, fp, prob, n_gt, dist) with the predictions ranked by
    (prob descending, row-major index ascending)."""
    if prob.dim() == 4:
        prob = prob[:, 0]
    if prob.dim() != 3 or tuple(keypoint_map.shape) != tuple(prob.shape):
        raise ValueError('detector_records: prob and keypoint_map must both be (B,H,W); got {} and {}'.format(
            tuple(prob.shape), tuple(keypoint_map.shape)))
    dev = _lib.require_cuda(prob.device if prob.device.type == 'cuda' else None)
    p = prob.to(dev, torch.float32).contiguous()
    g = keypoint_map.to(dev).ne(0).to(torch.uint8).contiguous()
    B, H, W = p.shape
    work = torch.empty((B, H * W), dtype=torch.int64, device=dev)
    rec_index = torch.empty((B, H * W), dtype=torch.int32, device=dev)
    rec_prob = torch.empty((B, H * W), dtype=torch.float32, device=dev)
    rec_bits = torch.empty((B, H * W), dtype=torch.int32, device=dev)
    rec_count = torch.empty((B,), dtype=torch.int32, device=dev)
    n_gt = torch.empty((B,), dtype=torch.int32, device=dev)
    h = _lib.get_handle(dev)
    with torch.cuda.device(dev):
        h.check(h.lib.mp_detector_metrics(h.ptr, _lib.ptr(p), _lib.ptr(g), B, H, W, float(zero_threshold),
                                          float(distance_thresh), _lib.ptr(work), _lib.ptr(rec_index),
                                          _lib.ptr(rec_prob), _lib.ptr(rec_bits), _lib.ptr(rec_count), _lib.ptr(n_gt),
                                          _lib.stream_ptr(dev)))
    cnt = rec_count.cpu().numpy()
    ngt = n_gt.cpu().numpy()
    nmax = int(cnt.max()) if B else 0
    idx = rec_index[:, :nmax].cpu().numpy()
    prb = rec_prob[:, :nmax].cpu().numpy()
    bits = rec_bits[:, :nmax].cpu().numpy().view(np.uint32)
    out = []
    for b in range(B):
        n = int(cnt[b])
        order = np.lexsort((idx[b, :n], -prb[b, :n].astype(np.float64)))        # prob desc, then index asc
        bb = bits[b, :n][order]
        tp = (bb >> np.uint32(31)).astype(bool)
        m = ((bb[:, None] >> np.arange(25, dtype=np.uint32)[None]) & np.uint32(1)).astype(bool)
        dist = np.broadcast_to(_WINDOW_DIST[None], m.shape)[m]                   # (prediction, window row-major) order
        out.append((tp, np.logical_not(tp), prb[b, :n][order], int(ngt[b]), dist.astype(np.float32)))
    return out


def compute_tp_fp_dist(prob, keypoints, zero_threshold=1e-4, distance_thresh=2.0):
    """evaluation.py:56-97 for one (H,W) heat map; `keypoints` is the (H,W) label map or an (N,2) (y,x) list."""
    prob = torch.as_tensor(prob)
    keypoints = torch.as_tensor(keypoints)
    if prob.shape != keypoints.shape:
        kk = keypoints.to(torch.int64).cpu()
        km = torch.zeros(tuple(prob.shape), dtype=torch.bool)
        km[kk[:, 0], kk[:, 1]] = True
        keypoints = km
    return detector_records(prob[None], keypoints[None], zero_threshold, distance_thresh)[0]


def compute_detector_metrics(net, dataloader, device, config):
    """Precision, recall and localisation error of the detector on a single-image loader with 'keypoints' labels;
    same signature and return value as the reference (evaluation.py:10-54): (precision, recall, prob, dist)."""
    from .utils import box_nms, data_to_device
    tp, fp, prob, n_gt, dist = [], [], [], 0, []
    for data in dataloader:
        data = data_to_device(data, device)
        out = net(data)
        if config['nms'] > 0:
            pred = box_nms(out['prob'], config['nms'], config['detection_threshold'], valid_mask=data['valid_mask'])
        else:
            pred = out['prob'] * data['valid_mask'].to(out['prob'].dtype)
        for item in detector_records(pred, data['keypoints']):
            tp.append(item[0]); fp.append(item[1]); prob.append(item[2]); n_gt += item[3]; dist.append(item[4])
    tp, fp = np.concatenate(tp), np.concatenate(fp)
    prob, dist = np.concatenate(prob), np.concatenate(dist)
    # evaluation.py:38-54
    sort_idx = np.argsort(prob)[::-1]
    tp, fp, prob = tp[sort_idx], fp[sort_idx], prob[sort_idx]
    tp_cum, fp_cum = np.cumsum(tp), np.cumsum(fp)
    recall = div0(tp_cum, n_gt)
    precision = div0(tp_cum, tp_cum + fp_cum)
    recall = np.concatenate([[0], recall, [1]])
    precision = np.concatenate([[0], precision, [0]])
    precision = np.maximum.accumulate(precision[::-1])[::-1]
    return precision, recall, prob, dist
