"""Host-side mirror of multipoint/utils/matching.py: get_matches with the mutual-nearest-neighbour methods of the hot
path ('bfmatcher' with crossCheck=True, 'nnmatcher') and the remaining modes ('bfmatcher' without crossCheck, the
knn_matches ratio test, 'thresholdmatcher'); every distance matrix is evaluated on the GPU and never materialised."""
import ctypes

import numpy as np
import torch

from .. import _lib

__all__ = ['get_matches', 'NNMatcher', 'ThresholdMatcher', 'DMatch', 'match_pairs', 'knn2_pairs']


class DMatch:
    """Stand-in for cv2.DMatch (queryIdx, trainIdx, distance) -- cv2 is not a dependency here."""
    __slots__ = ('queryIdx', 'trainIdx', 'distance', 'imgIdx')

    def __init__(self, queryIdx, trainIdx, distance):
        self.queryIdx, self.trainIdx, self.distance, self.imgIdx = int(queryIdx), int(trainIdx), float(distance), 0

    def __repr__(self):
        return 'DMatch(queryIdx=%d, trainIdx=%d, distance=%.6f)' % (self.queryIdx, self.trainIdx, self.distance)


def match_pairs(descA, countA, descB, countB, threshold=-1.0):
    """Mutual NN for P independent pairs on the GPU.
    descA/descB [P,K,D] fp32 unit rows, countA/countB [P] int32.
    Returns (match_idx [P,K] int32 (-1 = none), match_dist [P,K] f32, match_count [P] int32)."""
    dev = descA.device
    P, K, D = descA.shape
    descA = descA.contiguous(); descB = descB.contiguous()
    midx = torch.empty((P, K), dtype=torch.int32, device=dev)
    mdist = torch.empty((P, K), dtype=torch.float32, device=dev)
    mcnt = torch.empty((P,), dtype=torch.int32, device=dev)
    h = _lib.get_handle(dev)
    with torch.cuda.device(dev):
        h.check(h.lib.mp_match_mutual_nn(h.ptr, _lib.ptr(descA), _lib.ptr(countA.contiguous()),
                                         _lib.ptr(descB), _lib.ptr(countB.contiguous()),
                                         K * D, 1, P, K, D, float(threshold), _lib.ptr(midx),
                                         _lib.ptr(mdist), _lib.ptr(mcnt), _lib.stream_ptr(dev)))
    return midx, mdist, mcnt


def _mutual_nn(desc_1, desc_2, threshold):
    d1 = torch.as_tensor(desc_1); d2 = torch.as_tensor(desc_2)
    if d1.shape[0] == 0 or d2.shape[0] == 0:              # matching.py:46-47
        return []
    if d1.shape[1] != d2.shape[1]:
        raise AssertionError('descriptor sizes differ')    # matching.py:45
    dev = d1.device if d1.device.type == 'cuda' else _lib.require_cuda(None)
    N, M, D = d1.shape[0], d2.shape[0], d1.shape[1]
    K = max(N, M)
    A = torch.zeros((1, K, D), dtype=torch.float32, device=dev); A[0, :N] = d1.to(dev, torch.float32)
    Bm = torch.zeros((1, K, D), dtype=torch.float32, device=dev); Bm[0, :M] = d2.to(dev, torch.float32)
    nA = torch.tensor([N], dtype=torch.int32, device=dev); nB = torch.tensor([M], dtype=torch.int32, device=dev)
    midx, mdist, _ = match_pairs(A, nA, Bm, nB, threshold)
    midx = midx[0, :N].cpu().numpy(); mdist = mdist[0, :N].cpu().numpy()
    q = np.nonzero(midx >= 0)[0]
    return [DMatch(i, midx[i], mdist[i]) for i in q]


class NNMatcher():
    """multipoint/utils/matching.py:35-72 (mutual nearest neighbour + distance threshold)."""

    def __init__(self, threshold=0.7):
        self.nn_thresh = threshold
        if threshold < 0.0:
            raise ValueError('\'threshold\' should be non-negative')

    def match(self, desc1, desc2):
        return _mutual_nn(desc1, desc2, float(self.nn_thresh))


def _pad_pair(desc_1, desc_2):
    d1 = torch.as_tensor(desc_1); d2 = torch.as_tensor(desc_2)
    if d1.dim() != 2 or d2.dim() != 2 or d1.shape[1] != d2.shape[1]:
        raise AssertionError('descriptor sizes differ')
    dev = d1.device if d1.device.type == 'cuda' else _lib.require_cuda(None)
    N, M, D = d1.shape[0], d2.shape[0], d1.shape[1]
    K = max(N, M, 1)
    A = torch.zeros((1, K, D), dtype=torch.float32, device=dev); A[0, :N] = d1.to(dev, torch.float32)
    Bm = torch.zeros((1, K, D), dtype=torch.float32, device=dev); Bm[0, :M] = d2.to(dev, torch.float32)
    nA = torch.tensor([N], dtype=torch.int32, device=dev); nB = torch.tensor([M], dtype=torch.int32, device=dev)
    return A, nA, Bm, nB, N, M, K, D, dev


def knn2_pairs(descA, countA, descB, countB):
    """The two nearest rows of descB for every row of descA (L2), P independent pairs on the GPU.
    Returns (nn_idx [P,K,2] int32 (-1 = none), nn_dist [P,K,2] f32)."""
    dev = descA.device
    P, K, D = descA.shape
    idx = torch.full((P, K, 2), -1, dtype=torch.int32, device=dev)
    dist = torch.zeros((P, K, 2), dtype=torch.float32, device=dev)
    h = _lib.get_handle(dev)
    with torch.cuda.device(dev):
        h.check(h.lib.mp_match_knn2(h.ptr, _lib.ptr(descA.contiguous()), _lib.ptr(countA.contiguous()),
                                    _lib.ptr(descB.contiguous()), _lib.ptr(countB.contiguous()), K * D, 1, P, K, D,
                                    _lib.ptr(idx), _lib.ptr(dist), _lib.stream_ptr(dev)))
    return idx, dist


class _BFMatcher():
    """cv2.BFMatcher(cv2.NORM_L2, crossCheck=...) as get_matches builds it (matching.py:7): `match` with crossCheck is
    the symmetric mutual nearest neighbour (the shipped configuration, mp_match_mutual_nn), without it the nearest
    train row of every query; `knnMatch(d1, d2, k <= 2)` the k nearest (mp_match_knn2)."""

    def __init__(self, crossCheck=False, **kwargs):
        if kwargs:
            raise TypeError('unsupported BFMatcher arguments: %s' % sorted(kwargs))
        self.cross_check = bool(crossCheck)

    def _knn2(self, desc1, desc2):
        if len(desc1) == 0 or len(desc2) == 0:
            return np.zeros((len(desc1), 2), np.int32) - 1, np.zeros((len(desc1), 2), np.float32)
        A, nA, Bm, nB, N, M, K, D, dev = _pad_pair(desc1, desc2)
        idx, dist = knn2_pairs(A, nA, Bm, nB)
        return idx[0, :N].cpu().numpy(), dist[0, :N].cpu().numpy()

    def match(self, desc1, desc2):
        if self.cross_check:
            return _mutual_nn(desc1, desc2, -1.0)
        idx, dist = self._knn2(desc1, desc2)
        return [DMatch(i, idx[i, 0], dist[i, 0]) for i in range(len(idx)) if idx[i, 0] >= 0]

    def knnMatch(self, desc1, desc2, k):
        if self.cross_check and k != 1:
            raise ValueError('BFMatcher: crossCheck=True supports knnMatch with k=1 only (as OpenCV)')
        if k not in (1, 2):
            raise NotImplementedError('knnMatch is implemented for k <= 2 (get_matches uses k = 2)')
        if self.cross_check:
            return [[m] for m in self.match(desc1, desc2)]
        idx, dist = self._knn2(desc1, desc2)
        return [[DMatch(i, idx[i, c], dist[i, c]) for c in range(k) if idx[i, c] >= 0] for i in range(len(idx))]


class ThresholdMatcher():
    """multipoint/utils/matching.py:74-99: every (i, j) closer than the threshold, in row-major order."""

    def __init__(self, threshold=0.4):
        self.threshold = threshold
        if threshold < 0.0:
            raise ValueError('\'threshold\' should be non-negative')

    def match(self, desc1, desc2):
        if len(desc1) == 0 or len(desc2) == 0:        # matching.py:86-87
            return []
        A, nA, Bm, nB, N, M, K, D, dev = _pad_pair(desc1, desc2)
        h = _lib.get_handle(dev)
        cap = max(4 * K, 1024)
        while True:
            ij = torch.empty((1, cap, 2), dtype=torch.int32, device=dev)
            dd = torch.empty((1, cap), dtype=torch.float32, device=dev)
            cnt = torch.empty((1,), dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                h.check(h.lib.mp_match_threshold(h.ptr, _lib.ptr(A), _lib.ptr(nA), _lib.ptr(Bm), _lib.ptr(nB), K * D, 1,
                                                 1, K, D, float(self.threshold), cap, _lib.ptr(ij), _lib.ptr(dd),
                                                 _lib.ptr(cnt), _lib.stream_ptr(dev)))
            n = int(cnt.item())
            if n <= cap:
                break
            cap = n                                   # the list overflowed: one retry with the exact size
        ij = ij[0, :n].cpu().numpy(); dd = dd[0, :n].cpu().numpy()
        order = np.lexsort((ij[:, 1], ij[:, 0]))      # np.argwhere order (:92)
        return [DMatch(ij[o, 0], ij[o, 1], dd[o]) for o in order]


def get_matches(desc_1, desc_2, method='bfmatcher', knn_matches=False, **kwargs):
    """multipoint/utils/matching.py:4-33.  desc_1 (N,D), desc_2 (M,D): numpy arrays or tensors.
    Returns a list of DMatch ordered by queryIdx."""
    if method == 'bfmatcher':
        matcher = _BFMatcher(**kwargs)
    elif method == 'nnmatcher':
        matcher = NNMatcher(**kwargs)
    elif method == 'thresholdmatcher':
        matcher = ThresholdMatcher(**kwargs)
    elif method == 'flann':
        raise NotImplementedError("matching method 'flann' (cv2.FlannBasedMatcher: an approximate randomised kd-tree "
                                  "index) has no exact GPU counterpart; use 'bfmatcher', which returns the exact "
                                  "neighbours flann approximates")
    else:
        raise ValueError('unknown matching method')
    if knn_matches:
        all_matches = matcher.knnMatch(desc_1, desc_2, 2)       # AttributeError for nnmatcher / thresholdmatcher, as in the reference
        ratio_thresh = 0.9                                      # Lowe's ratio test (:22-27)
        matches = []
        for m, n in all_matches:
            if m.distance < ratio_thresh * n.distance:
                matches.append(m)
        return matches
    return matcher.match(desc_1, desc_2)
