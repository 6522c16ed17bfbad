"""Developer tool (GPU box): single-pair latency of the path as the predict_* scripts drive it (batch of one pair,
host synchronised after every call), per stage."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mp_oracle as O          # weight generator only
import multipoint_amd.models as models
import multipoint_amd.utils as U
from multipoint_amd.pipeline import PairPipeline

cfg = dict(O.SHIPPED_MODEL_CONFIG)
net = models.MultiPoint(cfg); net.load_state_dict(O.make_weights(0, cfg)); net.to('cuda'); net.eval()
pred = {'nms': 4, 'detection_threshold': 0.015, 'topk': 1000,
        'matching': {'method': 'bfmatcher', 'method_kwargs': {'crossCheck': True}, 'knn_matches': False}}
for H, W in ((240, 320), (480, 640)):
    img = torch.rand(2, 1, H, W, device='cuda')
    pipe = PairPipeline(net, pred, capacity=1000, overlap_post=False)

    def timed(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
        return np.median(t) * 1e3

    def fwd():
        return net({'image': img})
    out = fwd()

    def nms():
        return U.box_nms(out['prob'], 4, 0.015, keep_top_k=1000)

    def whole():
        r = pipe.run_interleaved(img); r.wait(); return r
    net.profile(True); fwd(); torch.cuda.synchronize(); gpu = sum(ms for _, ms, _ in net.profile_read()); net.profile(False)
    print('%dx%d: forward %.3f ms (kernels %.3f ms), box_nms %.3f ms, whole pair (pipeline) %.3f ms'
          % (H, W, timed(fwd), gpu, timed(nms), timed(whole)))
