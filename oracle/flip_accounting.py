"""TEST INFRASTRUCTURE (oracle side): accounting of end-to-end keypoint differences between two probability maps.

The reference's contract for the detector is an index list: `torch.nonzero(box_nms(prob * mask, ...) > thr)`
(multipoint/utils/evaluation.py:234-263, predict_align_image_pair.py:128-171).  Given the SAME probability map the HIP
path reproduces that list bit for bit (tests/test_gpu_parity.py::test_box_nms_bit_exact).  End to end the GPU map differs
from the CPU map by fp32 summation order (~1e-5), and greedy NMS / top-k are discontinuous: a keypoint can flip where two
scores, or a score and a threshold, are closer than that noise -- and a flip can cascade to footprint neighbours.  This
module decides, for every keypoint that differs between the two index lists, whether it is EXPLAINED by such a flip:

  stage 1, NMS without top-k.  By induction over the priority order (score desc, flat index asc) a candidate's fate can
  differ between two maps only if (a) it crosses the detection threshold, (c) its order relative to a footprint neighbour
  differs, or (d) a footprint neighbour's fate differs.  So every connected component (footprint adjacency) of the
  symmetric difference of the two survivor sets must contain a ROOT of kind (a) or (c); the root's margin
  (|s - thr| or |s_p - s_q| on the CPU map) is by construction <= the measured |prob_gpu - prob_cpu| at the pixels
  involved.  A component without a root is UNEXPLAINED (it would be a bug of the NMS, not noise).

  stage 2, top-k.  A survivor of both maps that is inside the k best on one side only has changed rank; its rank
  difference is exactly (#differing survivors ahead of it) + (#common survivors whose order relative to it flipped),
  so at least one of the two must be non-zero.

Only tests/, __graft_entry__.smoke() and bench.py's parity leg import this file."""
import numpy as np


def footprint_offsets(size, iou):
    """Offsets (dy, dx) != (0, 0) at which two size x size boxes overlap with IoU > iou, evaluated with torchvision's fp32
    formula inter / (area_i + area_j - inter) (oracle/nms_greedy.c)."""
    size = np.float32(size)
    r = int(np.ceil(float(size)))
    offs = []
    area = size * size
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            if dy == 0 and dx == 0:
                continue
            w = max(np.float32(0), size - np.float32(abs(dx)))
            h = max(np.float32(0), size - np.float32(abs(dy)))
            inter = np.float32(w * h)
            if float(np.float32(inter / np.float32(area + area - inter))) > float(iou):      # fp32 ovr against the DOUBLE threshold
                offs.append((dy, dx))
    return offs


def _before(score, idx_q, idx_p):
    """q precedes p in greedy order: higher score first, lower flat index on ties."""
    sq, sp = score.flat[idx_q], score.flat[idx_p]
    return sq > sp or (sq == sp and idx_q < idx_p)


def _topk(score, surv_idx, k):
    """Flat indices of the k best survivors under (score desc, index asc); all of them if k <= 0."""
    if k <= 0 or len(surv_idx) <= k:
        return set(int(i) for i in surv_idx)
    s = score.flat[surv_idx]
    order = np.lexsort((surv_idx, -s.astype(np.float64)))
    return set(int(i) for i in surv_idx[order[:k]])


def account_image(prob_cpu, prob_gpu, surv_cpu, surv_gpu, size, thr, iou, topk):
    """prob_*: (H, W) fp32 maps (valid mask already applied); surv_*: dense NMS outputs WITHOUT top-k of the respective map
    (zeros except survivors).  Returns a dict of counts and margins (see module docstring)."""
    prob_cpu = np.asarray(prob_cpu, np.float32); prob_gpu = np.asarray(prob_gpu, np.float32)
    H, W = prob_cpu.shape
    thr = np.float32(thr)
    eps = np.abs(prob_gpu.astype(np.float64) - prob_cpu.astype(np.float64))
    offs = footprint_offsets(size, iou)
    s_cpu = np.flatnonzero(np.asarray(surv_cpu).ravel() > 0)
    s_gpu = np.flatnonzero(np.asarray(surv_gpu).ravel() > 0)
    set_cpu, set_gpu = set(s_cpu.tolist()), set(s_gpu.tolist())
    diff = sorted(set_cpu ^ set_gpu)
    cand_cpu, cand_gpu = prob_cpu > thr, prob_gpu > thr

    def neighbours(p):
        y, x = divmod(p, W)
        for dy, dx in offs:
            yy, xx = y + dy, x + dx
            if 0 <= yy < H and 0 <= xx < W:
                yield yy * W + xx

    # ---- stage 1: components of the survivor difference, each needs a root ----
    dset = set(diff)
    root_margin = {}                    # differing survivor -> smallest margin of a root flip it takes part in
    root_eps = {}
    for p in diff:
        best = None
        if cand_cpu.flat[p] != cand_gpu.flat[p]:                                    # (a) threshold crossing
            best = (abs(float(prob_cpu.flat[p]) - float(thr)), eps.flat[p])
        for q in neighbours(p):
            if not (cand_cpu.flat[q] or cand_gpu.flat[q]):
                continue
            if _before(prob_cpu, q, p) != _before(prob_gpu, q, p):                  # (c) order flip with a neighbour
                m = (abs(float(prob_cpu.flat[p]) - float(prob_cpu.flat[q])), eps.flat[p] + eps.flat[q])
                if best is None or m[0] < best[0]:
                    best = m
        if best is not None:
            root_margin[p], root_eps[p] = best
    comp = {}
    ncomp = 0
    for p in diff:
        if p in comp:
            continue
        stack = [p]; comp[p] = ncomp
        while stack:
            a = stack.pop()
            for q in neighbours(a):
                if q in dset and q not in comp:
                    comp[q] = ncomp; stack.append(q)
        ncomp += 1
    comp_has_root = [False] * ncomp
    for p in root_margin:
        comp_has_root[comp[p]] = True
    unexplained = [p for p in diff if not comp_has_root[comp[p]]]
    # how far from noise an unexplained keypoint is: its closest decision margin beyond the measured errors
    max_unexpl = 0.0
    for p in unexplained:
        m = abs(float(prob_cpu.flat[p]) - float(thr)) - eps.flat[p]
        for q in neighbours(p):
            if cand_cpu.flat[q] or cand_gpu.flat[q]:
                m = min(m, abs(float(prob_cpu.flat[p]) - float(prob_cpu.flat[q])) - (eps.flat[p] + eps.flat[q]))
        max_unexpl = max(max_unexpl, m, 1e-30)                                       # > 0 even if a margin is tiny

    # ---- stage 2: top-k ----
    t_cpu, t_gpu = _topk(prob_cpu, s_cpu, topk), _topk(prob_gpu, s_gpu, topk)
    final_diff = sorted(t_cpu ^ t_gpu)
    common = set_cpu & set_gpu
    boundary = 0
    boundary_margin = 0.0
    boundary_margin_eps = 0.0
    for p in final_diff:
        if p in dset:
            continue
        boundary += 1
        ahead_diff = sum(1 for q in diff if _before(prob_cpu if q in set_cpu else prob_gpu, q, p))
        flipped = [q for q in common if q != p and _before(prob_cpu, q, p) != _before(prob_gpu, q, p)]
        if ahead_diff + len(flipped) == 0:
            unexplained.append(p)
            max_unexpl = max(max_unexpl, 1.0)
        elif ahead_diff == 0:
            m = min((abs(float(prob_cpu.flat[p]) - float(prob_cpu.flat[q])), eps.flat[p] + eps.flat[q]) for q in flipped)
            if m[0] > boundary_margin:
                boundary_margin, boundary_margin_eps = m
    rm = max(root_margin.values()) if root_margin else 0.0
    roots_within_noise = all(root_margin[p] <= root_eps[p] + 1e-12 for p in root_margin)
    return {
        'keypoints_cpu': len(t_cpu), 'keypoints_gpu': len(t_gpu), 'keypoints_total': len(t_cpu | t_gpu),
        'keypoints_differing': len(final_diff),
        'survivors_differing_before_topk': len(diff), 'flip_components': ncomp, 'root_flips': len(root_margin),
        'topk_boundary_flips': boundary,
        'max_root_margin': max(rm, boundary_margin),
        'max_prob_err_at_roots': max([root_eps[p] for p in root_margin] + [boundary_margin_eps, 0.0]),
        'max_prob_err': float(eps.max()),
        'roots_within_measured_noise': bool(roots_within_noise and boundary_margin <= boundary_margin_eps + 1e-12),
        'unexplained': len(unexplained), 'max_unexplained_margin': float(max_unexpl) if unexplained else 0.0,
        'final_cpu': t_cpu, 'final_gpu': t_gpu,
    }


def account_batch(prob_cpu, prob_gpu, nms_fn, size, thr, iou=0.1, topk=0):
    """prob_*: (B,1,H,W) numpy fp32.  nms_fn(prob4d) -> dense NMS map without top-k (the oracle's box_nms, applied to both
    maps: the HIP NMS is compared with the oracle on the GPU map separately, bit for bit).  Returns (summary, per_image)."""
    # image by image: images are independent (the batched coordinate-offset trick of utils.py:99-103 never lets boxes of
    # different images overlap) and the greedy kernel is quadratic in the number of candidates of one call
    per = []
    for b in range(prob_cpu.shape[0]):
        surv_c = nms_fn(prob_cpu[b:b + 1]); surv_g = nms_fn(prob_gpu[b:b + 1])
        per.append(account_image(prob_cpu[b, 0], prob_gpu[b, 0], surv_c[0, 0], surv_g[0, 0], size, thr, iou, topk))
    keys_sum = ['keypoints_cpu', 'keypoints_gpu', 'keypoints_total', 'keypoints_differing', 'survivors_differing_before_topk',
                'flip_components', 'root_flips', 'topk_boundary_flips', 'unexplained']
    keys_max = ['max_root_margin', 'max_prob_err_at_roots', 'max_prob_err', 'max_unexplained_margin']
    summary = {k: int(sum(a[k] for a in per)) for k in keys_sum}
    summary.update({k: float(max(a[k] for a in per)) for k in keys_max})
    summary['roots_within_measured_noise'] = all(a['roots_within_measured_noise'] for a in per)
    summary['images'] = len(per)
    summary['images_identical'] = sum(1 for a in per if a['keypoints_differing'] == 0)
    return summary, per
