"""Developer tool (GPU box): step time with / without the post-processing on the side stream, and forward only.
    python tools/post_prof.py [c3|c5]"""
import torch, time, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mp_oracle as O
import multipoint_amd.models as M, multipoint_amd.utils as U
from multipoint_amd.pipeline import PairPipeline
from multipoint_amd.datasets import SyntheticPairs
C5 = len(sys.argv) > 1 and sys.argv[1] == 'c5'
cfg = dict(O.SHIPPED_MODEL_CONFIG); cfg['mixed_precision'] = C5
HH, WW, TOPK = (1024, 1280, 2000) if C5 else (480, 640, 1000)
net = M.MultiPoint(dict(cfg)); net.load_state_dict(O.make_weights(0, cfg)); net.to("cuda"); net.eval()
P = 8 if C5 else 32
imgs = np.empty((2*P,1,HH,WW), np.float32)
for p in range(P): imgs[2*p], imgs[2*p+1] = SyntheticPairs.make_pair(0,p,HH,WW)
x = torch.from_numpy(imgs).cuda()
pred = {"nms":4,"detection_threshold":0.015,"topk":TOPK,"matching":{"method":"bfmatcher","method_kwargs":{"crossCheck":True},"knn_matches":False}}
for overlap in (True, False):
    pipe = PairPipeline(net, pred, capacity=TOPK, overlap_post=overlap)
    for _ in range(3): pipe.run_interleaved(x)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(10): pipe.run_interleaved(x)
    torch.cuda.synchronize(); print("overlap", overlap, "%.3f ms/step" % ((time.time()-t)*100))
for _ in range(3): net({"image": x})
torch.cuda.synchronize(); t=time.time()
for _ in range(10): net({"image": x})
torch.cuda.synchronize(); print("forward only %.3f ms" % ((time.time()-t)*100))
