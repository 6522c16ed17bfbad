"""Developer tool (GPU box): the post-processing stages timed alone (nothing else on the GPU), c3 or c5 shapes.
    python tools/post_kernels.py [c3|c5]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mp_oracle as O          # weight / image generators only
import multipoint_amd.models as M
import multipoint_amd.utils as U
from multipoint_amd.datasets import SyntheticPairs

C5 = len(sys.argv) > 1 and sys.argv[1] == 'c5'
cfg = dict(O.SHIPPED_MODEL_CONFIG); cfg['mixed_precision'] = C5
H, W, K, P = (1024, 1280, 2000, 8) if C5 else (480, 640, 1000, 32)
net = M.MultiPoint(cfg); net.load_state_dict(O.make_weights(0, cfg)); net.to('cuda'); net.eval()
imgs = np.empty((2 * P, 1, H, W), np.float32)
for p in range(P):
    imgs[2 * p], imgs[2 * p + 1] = SyntheticPairs.make_pair(0, p, H, W)
out = net({'image': torch.from_numpy(imgs).cuda()})
prob, desc = out['prob'], out['desc']


def timed(name, fn, n=20):
    for _ in range(3):
        r = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    print('%-44s %8.3f ms' % (name, (time.perf_counter() - t0) / n * 1e3))
    return r


for rounds in (1, 2, 4, 8):
    timed('detect_keypoints (NMS %d rounds + select)' % rounds,
          lambda: U.detect_keypoints(prob, 4, 0.015, keep_top_k=K, capacity=K, max_rounds=rounds))
    print('   unresolved after %d rounds: %d' % (rounds, U.nms_unresolved()))
kp, sc, cnt = U.detect_keypoints(prob, 4, 0.015, keep_top_k=K, capacity=K, max_rounds=8)
timed('extract_keypoints (threshold only)', lambda: U.extract_keypoints(prob, 0.5, K))
d = timed('interpolate_descriptors_batched', lambda: U.interpolate_descriptors_batched(kp, cnt, desc, H, W))
A, B = d[0::2].contiguous(), d[1::2].contiguous()
timed('match_pairs (mutual NN, %d pairs, K=%d)' % (P, K), lambda: U.match_pairs(A, cnt[0::2].contiguous(), B, cnt[1::2].contiguous()))
