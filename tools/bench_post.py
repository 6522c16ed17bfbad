"""Developer tool (GPU box): the post-processing chain (NMS, top-k, sampling, matching) of one batch ALONE on the GPU: total
time, and how many NMS rounds the batch needs.   python3 tools/bench_post.py [rounds]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mp_oracle as O
import multipoint_amd.models as models
from multipoint_amd.pipeline import PairPipeline
cfg = O.SHIPPED_MODEL_CONFIG
net = models.MultiPoint(cfg); net.load_state_dict(O.make_weights(0, cfg)); net.to('cuda'); net.eval()
PRED = {'nms': 4, 'detection_threshold': 0.015, 'topk': 1000, 'matcher': 'bfmatcher', 'cross_check': True}
img = O.make_images(1, 64, 480, 640).cuda()
out = net({'image': img})
for rounds in ([int(sys.argv[1])] if len(sys.argv) > 1 else [8, 6, 5, 4, 3, 2]):
    pipe = PairPipeline(net, PRED, capacity=1000, nms_rounds=rounds, overlap_post=False)
    for _ in range(3): res = pipe._post(out, None, img.device, 64, 480, 640)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): res = pipe._post(out, None, img.device, 64, 480, 640)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    pipe._last = res
    try:
        pipe.check_converged('cuda:0'); ok = 'converged'
    except RuntimeError as e:
        ok = str(e)[:80]
    print('rounds %d: post alone %.3f ms per batch of 32 pairs; %s' % (rounds, ms, ok))
