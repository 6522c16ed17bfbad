"""CPU: the NMS oracle.  torchvision (the reference's NMS) is not installable here, so the C
restatement is cross-checked against an independent numpy implementation of the same published
algorithm and against the defining property of greedy NMS."""
import numpy as np
import pytest


def _heatmap(rng, H, W, density, levels=0):
    p = rng.random((H, W), dtype=np.float32)
    p = np.where(rng.random((H, W)) < density, p, 0).astype(np.float32)
    if levels:                                   # heavy ties
        p = (np.floor(p * levels) / levels).astype(np.float32)
    return p


@pytest.mark.parametrize('seed,levels', [(0, 0), (1, 0), (2, 4), (3, 16)])
def test_c_matches_numpy_greedy(oracle, seed, levels):
    rng = np.random.default_rng(seed)
    p = _heatmap(rng, 40, 56, 0.4, levels)
    a = oracle.box_nms(p, 4, 0.015, use_c=True)
    b = oracle.box_nms(p, 4, 0.015, use_c=False)
    assert np.array_equal(a, b)
    assert (a > 0).sum() > 10


def _suppresses(dy, dx, size, iou):
    f = np.float32
    w = max(f(0), f(size) - f(abs(dy))); h = max(f(0), f(size) - f(abs(dx)))
    inter = f(w * h)
    return inter / (f(size) * f(size) * f(2) - inter) > f(iou)


@pytest.mark.parametrize('size,iou', [(4, 0.1), (2, 0.1), (8, 0.1), (4, 0.3), (3, 0.05)])
def test_greedy_property(oracle, size, iou):
    """kept <=> no kept pixel of higher priority (score desc, index asc) within the IoU footprint."""
    rng = np.random.default_rng(7)
    p = _heatmap(rng, 36, 44, 0.5, 8)
    out = oracle.box_nms(p, size, 0.015, iou=iou)
    H, W = p.shape
    R = int(np.ceil(size))
    kept = out > 0
    assert np.array_equal(out[kept], p[kept])
    for y in range(H):
        for x in range(W):
            if not p[y, x] > 0.015:
                assert not kept[y, x]
                continue
            blocked = False
            for dy in range(-R, R + 1):
                for dx in range(-R, R + 1):
                    yy, xx = y + dy, x + dx
                    if (dy or dx) and 0 <= yy < H and 0 <= xx < W and kept[yy, xx] and _suppresses(dy, dx, size, iou):
                        hp = p[yy, xx] > p[y, x] or (p[yy, xx] == p[y, x] and (yy, xx) < (y, x))
                        if hp:
                            blocked = True
                        else:
                            assert not kept[y, x]          # two kept pixels never overlap
            assert kept[y, x] == (not blocked)


def test_footprint_size4(oracle):
    """size=4, iou=0.1: the 7x7 window minus the 12 corner cells (hard part 3)."""
    fp = {(dy, dx) for dy in range(-4, 5) for dx in range(-4, 5) if _suppresses(dy, dx, 4, 0.1)}
    assert len(fp) == 37 and (3, 1) in fp and (3, 2) not in fp and (2, 2) in fp and (0, 3) in fp and (0, 4) not in fp


def test_topk_and_batch_independence(oracle):
    rng = np.random.default_rng(11)
    p = np.stack([_heatmap(rng, 48, 64, 0.3, 0) for _ in range(3)])[:, None]
    full = oracle.box_nms(p, 4, 0.015)
    for b in range(3):
        assert np.array_equal(full[b, 0], oracle.box_nms(p[b, 0], 4, 0.015))     # images independent
    k = 20
    top = oracle.box_nms(p, 4, 0.015, keep_top_k=k)
    for b in range(3):
        kept = np.sort(full[b, 0][full[b, 0] > 0])[::-1]
        assert (top[b, 0] > 0).sum() == min(k, len(kept))
        assert np.array_equal(np.sort(top[b, 0][top[b, 0] > 0])[::-1], kept[:k])
    # ties at the k-th score: lower row-major index wins
    q = np.zeros((16, 32), dtype=np.float32)
    q[2, 3] = q[2, 20] = q[10, 5] = q[10, 25] = 0.5
    t = oracle.box_nms(q, 4, 0.015, keep_top_k=2)
    assert t[2, 3] == 0.5 and t[2, 20] == 0.5 and t[10, 5] == 0 and t[10, 25] == 0


def test_edge_cases(oracle):
    z = np.zeros((16, 24), dtype=np.float32)
    assert not oracle.box_nms(z, 4, 0.015).any()                    # no candidates
    assert not oracle.box_nms(z[None, None], 4, 0.015, keep_top_k=5).any()
    with pytest.raises(ValueError):
        oracle.box_nms(np.zeros((2, 16, 24), dtype=np.float32), 4, 0.015)
    c = np.full((16, 24), 0.3, dtype=np.float32)                    # all equal: pure index tie-break
    out = oracle.box_nms(c, 4, 0.015)
    assert out[0, 0] == np.float32(0.3) and out[0, 1] == 0 and (out > 0).sum() > 6
    assert oracle.keypoints_from_map(out, 0.015).dtype == np.int64


def test_box_nms_dispatch_does_not_change_the_result():
    """torchvision.ops.batched_nms loops over the images above 4000 box coordinates and takes ONE coordinate-offset call below:
    the oracle restates both dispatches (bench.py times the reference's), and they must agree -- images never interact."""
    import numpy as np
    from oracle import mp_oracle as O
    rng = np.random.default_rng(5)
    prob = rng.random((3, 1, 40, 56), dtype=np.float32)
    prob[prob < 0.6] = 0.0
    prob[1, 0, 10:14, 10:14] = 0.9                          # exact ties
    assert (prob > 0.015).sum() * 4 > 4000                  # the per-image branch is taken
    for topk in (0, 50):
        a = O.box_nms(prob, 4, 0.015, keep_top_k=topk)
        b = O.box_nms(prob, 4, 0.015, keep_top_k=topk, dispatch='torchvision')
        assert np.array_equal(a, b)
    small = np.zeros((2, 1, 16, 16), dtype=np.float32); small[0, 0, 3, 3] = 0.5; small[1, 0, 8, 9] = 0.7
    assert np.array_equal(O.box_nms(small, 4, 0.015), O.box_nms(small, 4, 0.015, dispatch='torchvision'))


def test_threshold_is_compared_in_double(oracle):
    """torchvision's CPU kernel is nms_kernel_impl(dets, scores, double iou_threshold): the fp32 overlap ratio is promoted and
    compared with the caller's Python float.  Boxes of size 11 at offset 9 overlap by 22 / 220, which rounds to 0.1f --
    greater than the double 0.1 (suppressed), not greater than 0.1f (what a float comparison would say)."""
    f = np.float32
    ovr = f(22) / f(220)
    assert ovr == f(0.1) and float(ovr) > 0.1 and not (ovr > f(0.1))
    p = np.zeros((8, 40), dtype=np.float32)
    p[4, 10] = 0.9; p[4, 19] = 0.5                        # offset (0, 9)
    for use_c in (True, False):
        out = oracle.box_nms(p, 11, 0.015, iou=0.1, use_c=use_c)
        assert out[4, 10] == f(0.9) and out[4, 19] == 0    # suppressed, as by torchvision
        out = oracle.box_nms(p, 11, 0.015, iou=float(f(0.1)), use_c=use_c)
        assert out[4, 19] == f(0.5)                        # a threshold of exactly 0.1f is not exceeded
    from oracle import flip_accounting as FA
    assert (0, 9) in FA.footprint_offsets(11, 0.1) and (9, 0) in FA.footprint_offsets(11, 0.1)
    assert (0, 9) not in FA.footprint_offsets(11, float(f(0.1)))
