"""Developer probe (GPU box): two forwards in flight -- two models (two handles = two activation workspaces) and two PairPipelines
driven from two caller streams, steps alternating between them -- against the one-pipeline loop of bench.py.  Question: do the
workgroup tails and launch gaps of one forward get filled by the other's kernels?"""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mp_oracle as O
import multipoint_amd.models as M
from multipoint_amd.pipeline import PairPipeline
cfg = O.SHIPPED_MODEL_CONFIG
sd = O.make_weights(0, cfg)
pred = {'nms': 4, 'detection_threshold': 0.015, 'topk': 1000,
        'matching': {'method': 'bfmatcher', 'method_kwargs': {'crossCheck': True}, 'knn_matches': False}}
def mk():
    n = M.MultiPoint(cfg); n.load_state_dict(sd); n.to('cuda'); n.eval()
    return PairPipeline(n, pred, capacity=1000)
pipes = [mk(), mk()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
imgs = [torch.rand(64, 1, 480, 640, device='cuda') for _ in range(2)]
torch.cuda.synchronize()
def loop(n, dual):
    for i in range(n):
        k = i % 2 if dual else 0
        with torch.cuda.stream(streams[k]):
            pipes[k].run_interleaved(imgs[k])
for dual in (False, True, False, True):
    loop(6, dual); torch.cuda.synchronize()
    t = time.perf_counter(); loop(40, dual); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print('dual' if dual else 'single', '%.3f ms per step, %.1f pairs/s' % (dt / 40 * 1e3, 32 * 40 / dt), flush=True)
