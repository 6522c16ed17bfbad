"""TEST INFRASTRUCTURE ONLY -- independent CPU restatement of `cv2.findHomography(src, dst, cv2.RANSAC, thr)` as the
reference calls it (multipoint/utils/evaluation.py:349, predict_align_image_pair.py:216; defaults maxIters = 2000,
confidence = 0.995).

OpenCV is a third-party dependency of the reference (requirements.txt:1, opencv-python==4.2.0.34) whose source is NOT
under /root/reference and which is not installable here, so this file restates its PUBLISHED algorithm (OpenCV 4.2,
modules/calib3d/src/fundam.cpp `cv::findHomography`, `HomographyEstimatorCallback`, `HomographyRefineCallback`;
modules/calib3d/src/ptsetreg.cpp `RANSACPointSetRegistrator::run`, `RANSACUpdateNumIters`) -- NOT the product's kernel
(multipoint_amd/csrc/homography.hip), which is a different design (a fixed number of hypotheses evaluated in parallel, DLT
refit, no LM).  "PARITY UNPINNED": the reference holds no vector for this call and OpenCV's RNG stream is not reproduced,
so the comparison is statistical (reprojection error, inlier-set IoU, h_correctness), never bit equality.

  RANSAC loop          niters = maxIters; draw 4 distinct correspondences; reject degenerate subsets (checkSubset: three
                       collinear points, or a sample whose orientation signs differ between source and target);
                       4-point model by the normalised DLT (runKernel); inliers = squared reprojection error of
                       H*src against dst <= thr^2 (computeError / findInliers); keep the model with the most inliers and
                       shrink niters = RANSACUpdateNumIters(confidence, outlier ratio, 4, niters).
  final estimate       runKernel on ALL inliers of the best model (least-squares normalised DLT), then Levenberg-Marquardt
                       on the 8 free parameters minimising the reprojection error over the inliers (createLMSolver(..., 10)),
                       H /= H[2,2].
"""
import numpy as np


def _normalised_dlt(src, dst):
    """HomographyEstimatorCallback::runKernel: centroid / mean-absolute-deviation normalisation of both point sets, the
    9x9 normal matrix L^T L of the 2n x 9 DLT system, its eigenvector of the smallest eigenvalue, de-normalised and
    scaled to H[2,2] = 1.  Returns None when the configuration is degenerate."""
    src = np.asarray(src, np.float64); dst = np.asarray(dst, np.float64)
    n = len(src)
    cm = dst.mean(0); cM = src.mean(0)
    sm = np.abs(dst - cm).mean(0); sM = np.abs(src - cM).mean(0)
    if min(sm.min(), sM.min()) < np.finfo(np.float64).eps:
        return None
    sm = 1.0 / sm; sM = 1.0 / sM
    invHnorm = np.array([[1 / sm[0], 0, cm[0]], [0, 1 / sm[1], cm[1]], [0, 0, 1]])
    Hnorm2 = np.array([[sM[0], 0, -cM[0] * sM[0]], [0, sM[1], -cM[1] * sM[1]], [0, 0, 1]])
    LtL = np.zeros((9, 9))
    for i in range(n):
        x, y = (dst[i] - cm) * sm
        X, Y = (src[i] - cM) * sM
        lx = np.array([X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x])
        ly = np.array([0, 0, 0, X, Y, 1, -y * X, -y * Y, -y])
        LtL += np.outer(lx, lx) + np.outer(ly, ly)
    w, v = np.linalg.eigh(LtL)
    h0 = v[:, 0].reshape(3, 3)
    H = invHnorm @ h0 @ Hnorm2
    if abs(H[2, 2]) < np.finfo(np.float64).eps:
        return None
    return H / H[2, 2]


def _reproj_err2(H, src, dst):
    """HomographyEstimatorCallback::computeError: squared distance between H*src (dehomogenised) and dst."""
    p = np.concatenate([src, np.ones((len(src), 1))], 1) @ H.T
    w = p[:, 2]
    w = np.where(np.abs(w) > np.finfo(np.float64).eps, 1.0 / w, 0.0)
    d = p[:, :2] * w[:, None] - dst
    return (d * d).sum(1)


def _check_subset(ms1, ms2):
    """HomographyEstimatorCallback::checkSubset for a 4-point sample: no three points (nearly) collinear in either set
    (haveCollinearPoints) and the same orientation of every point triple in both sets (the "convexity" sign test)."""
    for pts in (ms1, ms2):
        for i in range(4):
            for j in range(i + 1, 4):
                for k in range(j + 1, 4):
                    d1 = pts[j] - pts[i]; d2 = pts[k] - pts[i]
                    if abs(d1[0] * d2[1] - d1[1] * d2[0]) <= np.finfo(np.float32).eps * (abs(d1).sum()) * (abs(d2).sum()):
                        return False
    neg = 0
    for (i, j, k) in ((0, 1, 2), (1, 2, 3), (0, 2, 3), (0, 1, 3)):
        A = np.array([[ms1[i][0], ms1[i][1], 1], [ms1[j][0], ms1[j][1], 1], [ms1[k][0], ms1[k][1], 1]])
        B = np.array([[ms2[i][0], ms2[i][1], 1], [ms2[j][0], ms2[j][1], 1], [ms2[k][0], ms2[k][1], 1]])
        neg += (np.linalg.det(A) * np.linalg.det(B)) < 0
    return neg == 0 or neg == 4


def ransac_update_num_iters(p, ep, model_points, max_iters):
    """RANSACUpdateNumIters (ptsetreg.cpp): iterations needed to draw an outlier-free sample with probability p at outlier
    ratio ep, capped at max_iters."""
    p = min(max(p, 0.0), 1.0); ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, np.finfo(np.float64).tiny)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < np.finfo(np.float64).tiny:
        return 0
    num = np.log(num); denom = np.log(denom)
    return max_iters if (denom >= 0 or -num >= max_iters * (-denom)) else int(round(num / denom))


def _lm_refine(H, src, dst, iters=10):
    """HomographyRefineCallback + createLMSolver(cb, 10): Levenberg-Marquardt on h = H.ravel()[:8] (H[2,2] = 1) minimising
    the reprojection residuals (x' - x, y' - y) over the inliers."""
    h = (H / H[2, 2]).ravel()[:8].copy()
    X, Y = src[:, 0], src[:, 1]

    def residual_jac(h):
        ww = h[6] * X + h[7] * Y + 1.0
        ww = np.where(np.abs(ww) > np.finfo(np.float64).eps, 1.0 / ww, 0.0)
        xi = (h[0] * X + h[1] * Y + h[2]) * ww
        yi = (h[3] * X + h[4] * Y + h[5]) * ww
        r = np.empty(2 * len(X)); r[0::2] = xi - dst[:, 0]; r[1::2] = yi - dst[:, 1]
        J = np.zeros((2 * len(X), 8))
        J[0::2, 0] = X * ww; J[0::2, 1] = Y * ww; J[0::2, 2] = ww
        J[0::2, 6] = -X * ww * xi; J[0::2, 7] = -Y * ww * xi
        J[1::2, 3] = X * ww; J[1::2, 4] = Y * ww; J[1::2, 5] = ww
        J[1::2, 6] = -X * ww * yi; J[1::2, 7] = -Y * ww * yi
        return r, J

    r, J = residual_jac(h)
    cost = r @ r
    lam = 1e-3
    for _ in range(iters):
        A = J.T @ J; g = J.T @ r
        for _try in range(8):
            try:
                step = np.linalg.solve(A + lam * np.diag(np.diag(A)), -g)
            except np.linalg.LinAlgError:
                lam *= 10.0
                continue
            r2, J2 = residual_jac(h + step)
            c2 = r2 @ r2
            if c2 < cost:
                h = h + step; r, J, cost = r2, J2, c2
                lam = max(lam * 0.1, 1e-12)
                break
            lam *= 10.0
        else:
            break
    return np.append(h, 1.0).reshape(3, 3)


def find_homography_ransac(src_xy, dst_xy, reproj_threshold=3.0, max_iters=2000, confidence=0.995, seed=0):
    """cv2.findHomography(src, dst, cv2.RANSAC, reproj_threshold, maxIters=2000, confidence=0.995).
    src_xy, dst_xy: (N, 2) point coordinates (x, y).  Returns (H 3x3 float64 or None, mask (N,) uint8)."""
    src = np.asarray(src_xy, np.float64).reshape(-1, 2); dst = np.asarray(dst_xy, np.float64).reshape(-1, 2)
    n = len(src)
    if n < 4:
        return None, np.zeros(n, np.uint8)
    rng = np.random.default_rng(seed)
    thr2 = float(reproj_threshold) ** 2
    if n == 4:
        H = _normalised_dlt(src, dst) if _check_subset(src, dst) else None
        if H is None:
            return None, np.zeros(n, np.uint8)
        return H, np.ones(n, np.uint8)
    best_mask, best_count = None, 0
    niters = max_iters
    it = 0
    while it < niters:
        it += 1
        idx = None
        for _attempt in range(1000):                       # getSubset: distinct indices, non-degenerate sample
            cand = rng.choice(n, 4, replace=False)
            if _check_subset(src[cand], dst[cand]):
                idx = cand
                break
        if idx is None:
            if it == 1:
                return None, np.zeros(n, np.uint8)
            break
        H = _normalised_dlt(src[idx], dst[idx])
        if H is None:
            continue
        mask = _reproj_err2(H, src, dst) <= thr2
        cnt = int(mask.sum())
        if cnt > max(best_count, 3):
            best_mask, best_count = mask, cnt
            niters = ransac_update_num_iters(confidence, (n - cnt) / n, 4, niters)
    if best_mask is None:
        return None, np.zeros(n, np.uint8)
    s_in, d_in = src[best_mask], dst[best_mask]
    H = _normalised_dlt(s_in, d_in)
    if H is None:
        return None, np.zeros(n, np.uint8)
    H = _lm_refine(H, s_in, d_in, 10)
    return H / H[2, 2], best_mask.astype(np.uint8)
