"""Developer tool (GPU box): get_matches('nnmatcher') against the oracle's NNMatcher restatement on random list sizes (0, 1,
non-multiples of 32, > 1024), descriptor sizes 64 / 128 / 256, duplicated rows (exact distance ties: lowest index must win) and
thresholds.    python tools/fuzz_match.py [trials] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import mp_oracle as O
import multipoint_amd.utils as U
NTR = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
worst = 0.0
for trial in range(NTR):
    D = int(rng.choice([64, 64, 128, 256]))
    N = int(rng.choice([0, 1, 2, 31, 32, 33, 100, 257, 1000, 1500])); M = int(rng.choice([0, 1, 5, 32, 63, 64, 65, 300, 999, 1100]))
    d1 = rng.standard_normal((N, D)).astype(np.float32); d2 = rng.standard_normal((M, D)).astype(np.float32)
    if N and M and trial % 2 == 0:           # plant correspondences and exact duplicates
        k = min(N, M) // 2
        d2[:k] = d1[:k] + 0.05 * rng.standard_normal((k, D)).astype(np.float32)
        if M > 3: d2[M - 1] = d2[0]; d2[M - 2] = d2[1]
        if N > 3: d1[N - 1] = d1[0]
    d1 /= np.maximum(np.linalg.norm(d1, axis=1, keepdims=True), 1e-12); d2 /= np.maximum(np.linalg.norm(d2, axis=1, keepdims=True), 1e-12)
    thr = float(rng.choice([0.7, 0.3, 1.2]))
    q, t, d = O.nn_match(d1, d2, thr)
    got = U.get_matches(torch.from_numpy(d1).cuda(), torch.from_numpy(d2).cuda(), 'nnmatcher', threshold=thr)
    gq = np.array([m.queryIdx for m in got], dtype=np.int64); gt = np.array([m.trainIdx for m in got], dtype=np.int64)
    gd = np.array([m.distance for m in got], dtype=np.float32)
    same = len(gq) == len(q) and np.array_equal(gq, q) and np.array_equal(gt, t)
    note = 'ok'
    if not same:
        # numpy's BLAS gives IDENTICAL rows different dot products depending on their position in the matrix (blocking), the GPU
        # does not: an index difference is legitimate iff it is between candidates whose reference distances agree to rounding
        dm = O.distance_matrix(d1, d2)
        mine = {m.queryIdx: m.trainIdx for m in got}; theirs = dict(zip(q.tolist(), t.tolist()))
        for i in set(mine) | set(theirs):
            a, b = mine.get(i), theirs.get(i)
            if a == b:
                continue
            cols = [c for c in (a, b) if c is not None]
            best = dm[i].min()
            assert all(abs(dm[i, c] - best) < 1e-5 for c in cols), ('not a near-tie', trial, i, a, b)
            # (a match that exists on one side only: the mutual test flipped on the same kind of tie in the other direction)
            assert a is None or b is None or abs(dm[i, a] - dm[i, b]) < 1e-5
        note = 'near-tie'
        err = 0.0
    else:
        err = float(np.abs(gd * gd - d * d).max()) if len(d) else 0.0     # (on d^2: the sqrt amplifies a dot product's rounding by 1 / d)
    worst = max(worst, err)
    print(trial, N, M, D, thr, len(q), note, '%.1e' % err, flush=True)
    assert err <= 4e-6
print('all', NTR, 'trials: indices exact, error of the squared distance <=', worst)
