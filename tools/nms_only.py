"""Developer tool (GPU box): box_nms + keypoint selection of one bench batch, a few calls (for rocprofv3 --kernel-trace --stats).
    python3 tools/nms_only.py [c5]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mp_oracle as O          # weight generator only
import multipoint_amd.models as M
import multipoint_amd.utils as U
from multipoint_amd.datasets import SyntheticPairs
cfg = dict(O.SHIPPED_MODEL_CONFIG)
net = M.MultiPoint(cfg); net.load_state_dict(O.make_weights(0, cfg)); net.to('cuda'); net.eval()
C5 = len(sys.argv) > 1 and sys.argv[1] == 'c5'
P, H, W, TOPK = (8, 1024, 1280, 2000) if C5 else (32, 480, 640, 1000)
imgs = np.empty((2 * P, 1, H, W), np.float32)
for p in range(P):
    imgs[2 * p], imgs[2 * p + 1] = SyntheticPairs.make_pair(0, p, H, W)
prob = net({'image': torch.from_numpy(imgs).cuda()})['prob']
print('candidates per image: %.0f' % ((prob > 0.015).sum().item() / (2 * P)))
for _ in range(6):
    U.detect_keypoints(prob, 4, 0.015, keep_top_k=TOPK, capacity=TOPK, max_rounds=8)
torch.cuda.synchronize()
