"""TEST INFRASTRUCTURE: error statistics in units of the fp16 spacing, shared by the CPU test that pins the fp16 oracle to the
reference's autocast fixture (tests/golden/forward_f16.npz) and the GPU tests of the fp16 MFMA path."""
import numpy as np


def ulp16(x):
    """fp16 spacing at |x| (normal range; 2^-24 for subnormals)."""
    a = np.maximum(np.abs(np.asarray(x, dtype=np.float64)), 2.0 ** -14)
    return 2.0 ** (np.floor(np.log2(a)) - 10)


def step_stats(got, ref, scale):
    """Error of `got` against `ref` in fp16 steps of `scale` (broadcastable)."""
    e = (np.asarray(got, np.float64) - np.asarray(ref, np.float64)) / ulp16(scale)
    ae = np.abs(e)
    return dict(median=float(np.median(ae)), p999=float(np.percentile(ae, 99.9)), max=float(ae.max()),
                mean_signed=float(e.mean()), mean_abs=float(ae.mean()))


def logits_stats(got, ref):
    # logits are sums of O(10) terms that cancel (the dustbin sits near 11): their fp16 noise is absolute, one step = the
    # spacing at magnitude 8..16 (2^-7), not the spacing of a logit that happens to be near zero
    return step_stats(got, ref, np.maximum(np.abs(np.asarray(ref, np.float64)), 8.0))


def desc_stats(got, ref, channel_axis):
    # one step = the fp16 spacing of the cell's largest component (the raw map is rounded to fp16 BEFORE it is normalised)
    return step_stats(got, ref, np.abs(np.asarray(ref, np.float64)).max(axis=channel_axis, keepdims=True))


def prob_rel_stats(got, ref, thr=0.015):
    p, rp = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    m = rp > thr
    rel = (p[m] - rp[m]) / rp[m]
    return dict(n=int(m.sum()), median_abs_rel=float(np.median(np.abs(rel))), p999_abs_rel=float(np.percentile(np.abs(rel), 99.9)),
                max_abs_rel=float(np.abs(rel).max()), mean_signed_rel=float(rel.mean()), mean_abs_rel=float(np.abs(rel).mean()),
                # standard error of the signed mean: a bias test on a few hundred samples must allow for it
                sem_rel=float(rel.std() / np.sqrt(max(rel.size, 1))))


def unbiased(ps, frac=0.1, sigmas=4.0):
    """|signed mean| is a small fraction of the mean |error|, up to the sampling noise of the mean itself."""
    return abs(ps['mean_signed_rel']) <= frac * ps['mean_abs_rel'] + sigmas * ps['sem_rel']


def fixture_views(z, out, which):
    """(got, ref) arrays for logits / prob / desc of fixture case 'a', 'c', 'd' (full maps) or 'b' (sampled) -- `out` holds full maps."""
    if which != 'b':
        w = which
        return {'logits': (out['logits'], z[w + '_logits']), 'prob': (out['prob'], z[w + '_prob']), 'desc': (out['desc'], z[w + '_desc'])}, 1
    d = np.asarray(out['desc'])
    return {'logits': (np.asarray(out['logits']).ravel()[z['b_logits_idx']], z['b_logits_val']),
            'prob': (np.asarray(out['prob']).ravel()[z['b_prob_idx']], z['b_prob_val']),
            'desc': (d.reshape(1, d.shape[1], -1)[0][:, z['b_desc_cells']], z['b_desc_val'])}, 0
