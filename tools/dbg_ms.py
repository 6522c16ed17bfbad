import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mp_oracle as O
import multipoint_amd.models as M
cfg = dict(O.SHIPPED_MODEL_CONFIG); cfg['multispectral'] = True
sd = O.make_weights(3, cfg)
net = M.MultiPoint(cfg); net.load_state_dict(sd); net.to('cuda')
for flags in ([True, False, True], [True, True, True], [False, False, False], [False, True, False]):
    img = O.make_images(13, 3, 32, 48)
    fl = torch.tensor(flags).reshape(-1, 1)
    ref = O.forward(sd, img, cfg, is_optical=fl)
    for rep in range(2):
        out = net({'image': img.cuda(), 'is_optical': fl})
        torch.cuda.synchronize()
        pe = (out['prob'].cpu() - ref['prob']).abs().flatten(1).max(1).values
        de = (out['desc'].cpu() - ref['desc']).abs().flatten(1).max(1).values
        print(flags, rep, 'prob err per image', pe.tolist(), 'desc', de.tolist())
