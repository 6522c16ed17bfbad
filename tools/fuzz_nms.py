"""Developer tool (GPU box): box_nms / detect_keypoints against the C oracle (oracle/nms_greedy.c) on random frame sizes
(ANY H x W since round 6 -- widths that are no multiple of 4 take the padded work map --, partial tiles, frames smaller than a tile),
densities, tie levels, box sizes up to 16 (incl. the sizes where the fp32 overlap ratio lands ON the threshold: 11 / 0.1), top-k and masks.
    python tools/fuzz_nms.py [trials] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import mp_oracle as O
import multipoint_amd.utils as U
NTR = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
for trial in range(NTR):
    B = int(rng.integers(1, 4)); H = int(rng.integers(1, 150)); W = int(rng.integers(1, 280)) if trial % 2 else 4 * int(rng.integers(1, 70))
    density = float(rng.choice([0.02, 0.1, 0.4, 1.0])); levels = int(rng.choice([0, 0, 2, 5]))
    size = float(rng.choice([1, 2, 3, 4, 4, 4, 5, 8, 9, 11, 12, 16, 2.5, 6.5])); iou = float(rng.choice([0.05, 0.1, 0.1, 0.3, 0.5])); topk = int(rng.choice([0, 0, 7, 100]))
    p = rng.random((B, 1, H, W), dtype=np.float32)
    p = np.where(rng.random((B, 1, H, W)) < density, p, 0).astype(np.float32)
    if levels:
        p = (np.floor(p * levels) / levels).astype(np.float32)
    mask = (rng.random((B, 1, H, W)) < 0.8) if trial % 3 == 0 else None
    ref = O.box_nms(p * mask if mask is not None else p, size, 0.015, iou=iou, keep_top_k=topk)
    vm = torch.from_numpy(mask).cuda() if mask is not None else None
    out = U.box_nms(torch.from_numpy(p).cuda(), size, 0.015, iou=iou, keep_top_k=topk, valid_mask=vm).cpu().numpy()
    ok = np.array_equal(out, ref)
    kp, sc, cnt = U.detect_keypoints(torch.from_numpy(p).cuda(), size, 0.015, iou=iou, keep_top_k=topk, capacity=H * W, valid_mask=vm)
    for b in range(B):
        okp = O.keypoints_from_map(ref[b, 0], 0.015)
        n = int(cnt[b])
        ok = ok and n == len(okp) and np.array_equal(kp[b, :n].cpu().numpy().astype(np.int64), okp)
    print(trial, B, H, W, density, levels, size, iou, topk, 'mask' if mask is not None else '-', 'ok' if ok else 'MISMATCH', flush=True)
    assert ok
print('all', NTR, 'trials bit-exact')
