"""CPU: the keypoint flip accounting (oracle/flip_accounting.py) that the end-to-end GPU parity test and bench.py's parity
leg use -- checked here on maps whose differences are known by construction."""
import numpy as np

from oracle import flip_accounting as FA


def _maps(seed, B=2, H=96, W=128):
    rng = np.random.default_rng(seed)
    p = rng.random((B, 1, H, W), dtype=np.float32) ** 8 * 0.3           # sparse-ish peaks above 0.015
    return p.astype(np.float32)


def test_footprint_matches_size4_iou01():
    offs = set(FA.footprint_offsets(4, 0.1))
    want = {(dy, dx) for dy in range(-3, 4) for dx in range(-3, 4)
            if (dy or dx) and (4 - abs(dy)) * (4 - abs(dx)) >= 3}
    assert offs == want


def test_identical_maps_have_no_flips(oracle):
    p = _maps(1)
    nms = lambda m: oracle.box_nms(m, 4, 0.015, keep_top_k=0)
    s, per = FA.account_batch(p, p.copy(), nms, 4, 0.015, 0.1, topk=150)
    assert s['keypoints_differing'] == 0 and s['unexplained'] == 0 and s['images_identical'] == 2
    assert s['keypoints_cpu'] == s['keypoints_gpu'] == 300


def test_noise_flips_are_explained(oracle):
    """fp32-noise-sized perturbations: whatever flips is explained and every root margin is below the injected noise."""
    p = _maps(2)
    rng = np.random.default_rng(3)
    # quantise the scores so that near-ties are frequent, then perturb by +-1e-5
    p = (np.round(p * 2000) / 2000).astype(np.float32)
    q = (p + rng.uniform(-1e-5, 1e-5, p.shape).astype(np.float32)).astype(np.float32)
    nms = lambda m: oracle.box_nms(m, 4, 0.015, keep_top_k=0)
    s, per = FA.account_batch(p, q, nms, 4, 0.015, 0.1, topk=150)
    assert s['keypoints_differing'] > 0, 'the construction should produce flips'
    assert s['unexplained'] == 0 and s['max_unexplained_margin'] == 0.0
    assert s['roots_within_measured_noise'] and s['max_root_margin'] <= 2e-5 + 1e-9
    # the final sets the accounting reports are the oracle's own top-k lists
    for b in range(2):
        full = oracle.box_nms(p[b, 0], 4, 0.015, keep_top_k=150)
        assert per[b]['final_cpu'] == set(np.flatnonzero(full.ravel() > 0).tolist())


def test_a_wrong_nms_result_is_unexplained(oracle):
    """Same map on both sides, but one side's survivor set is corrupted (a keypoint dropped far from any tie): that is not
    noise and must be reported."""
    p = _maps(4, B=1)
    surv = oracle.box_nms(p, 4, 0.015, keep_top_k=0)
    bad = surv.copy()
    idx = np.flatnonzero(bad.ravel() > 0)
    bad.ravel()[idx[len(idx) // 2]] = 0
    a = FA.account_image(p[0, 0], p[0, 0], surv[0, 0], bad[0, 0], 4, 0.015, 0.1, 0)
    assert a['unexplained'] == 1 and a['max_unexplained_margin'] > 0
