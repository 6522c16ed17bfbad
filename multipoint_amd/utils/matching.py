"""Host-side mirror of multipoint/utils/matching.py: get_matches with the mutual-nearest-neighbour
methods of the hot path ('bfmatcher' with crossCheck=True, 'nnmatcher')."""
import ctypes

import numpy as np
import torch

from .. import _lib

__all__ = ['get_matches', 'NNMatcher', 'DMatch', 'match_pairs']


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


class _CrossCheckBFMatcher():
    """cv2.BFMatcher(cv2.NORM_L2, crossCheck=True).match for L2-normalised descriptors
    (matching.py:7,31): symmetric mutual nearest neighbour, no threshold."""

    def __init__(self, crossCheck=False, **kwargs):
        if kwargs:
            raise TypeError('unsupported BFMatcher arguments: %s' % sorted(kwargs))
        if not crossCheck:
            raise NotImplementedError('bfmatcher is implemented for crossCheck=True only '
                                      '(the configuration the reference ships and evaluates with)')

    def match(self, desc1, desc2):
        return _mutual_nn(desc1, desc2, -1.0)


def get_matches(desc_1, desc_2, method='bfmatcher', knn_matches=False, **kwargs):
    """multipoint/utils/matching.py:4-33.  desc_1 (N,D), desc_2 (M,D): numpy arrays or tensors.
    Returns a list of DMatch ordered by queryIdx."""
    if method == 'bfmatcher':
        matcher = _CrossCheckBFMatcher(**kwargs)
    elif method == 'nnmatcher':
        matcher = NNMatcher(**kwargs)
    elif method in ('flann', 'thresholdmatcher'):
        raise NotImplementedError("matching method '%s' is outside the accelerated hot path "
                                  "(supported: 'bfmatcher' with crossCheck=True, 'nnmatcher')" % method)
    else:
        raise ValueError('unknown matching method')
    if knn_matches:
        raise NotImplementedError('knn_matches (Lowe ratio test) is outside the accelerated hot path')
    return matcher.match(desc_1, desc_2)
