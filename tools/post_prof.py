import torch, time, numpy as np, sys
sys.path.insert(0, "/root/repo")
from oracle import mp_oracle as O
import multipoint_amd.models as M, multipoint_amd.utils as U
from multipoint_amd.pipeline import PairPipeline
from multipoint_amd.datasets import SyntheticPairs
cfg = O.SHIPPED_MODEL_CONFIG
net = M.MultiPoint(dict(cfg)); net.load_state_dict(O.make_weights(0, cfg)); net.to("cuda"); net.eval()
P=32
imgs = np.empty((2*P,1,480,640), np.float32)
for p in range(P): imgs[2*p], imgs[2*p+1] = SyntheticPairs.make_pair(0,p,480,640)
x = torch.from_numpy(imgs).cuda()
pred = {"nms":4,"detection_threshold":0.015,"topk":1000,"matching":{"method":"bfmatcher","method_kwargs":{"crossCheck":True},"knn_matches":False}}
for overlap in (True, False):
    pipe = PairPipeline(net, pred, capacity=1000, overlap_post=overlap)
    for _ in range(3): pipe.run_interleaved(x)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(10): pipe.run_interleaved(x)
    torch.cuda.synchronize(); print("overlap", overlap, "%.3f ms/step" % ((time.time()-t)*100))
for _ in range(3): net({"image": x})
torch.cuda.synchronize(); t=time.time()
for _ in range(10): net({"image": x})
torch.cuda.synchronize(); print("forward only %.3f ms" % ((time.time()-t)*100))
