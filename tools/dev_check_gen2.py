"""Developer check (GPU box): conv_wino43b.hip (MP_DEBUG=wino43_gen=2) against the CPU oracle on a few shapes, incl. frames that are
no multiple of the 4x4 tile.  Run as: MP_DEBUG=wino43_gen=2 python tools/dev_check_gen2.py"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mp_oracle as O
import multipoint_amd.models as models

cfg = dict(O.SHIPPED_MODEL_CONFIG)
sd = O.make_weights(0, cfg)
net = models.MultiPoint(cfg); net.load_state_dict(sd); net.to('cuda'); net.eval()
worst = 0.0
for (B, H, W) in [(2, 64, 64), (1, 240, 320), (3, 72, 104), (2, 480, 640), (1, 88, 120), (5, 128, 160)]:
    img = O.make_images(3, B, H, W)
    ref = O.forward(sd, img, cfg)
    out = net({'image': img.cuda()})
    torch.cuda.synchronize()
    p = out['prob'].cpu(); d = out['desc'].cpu()
    ep = (p - ref['prob']).abs().max().item(); ed = (d - ref['desc']).abs().max().item()
    worst = max(worst, ep, ed)
    print('gen', os.environ.get('MP_DEBUG', ''), 'fwd', (B, H, W), 'prob maxabs %.3g' % ep, 'desc maxabs %.3g' % ed,
          'nan', torch.isnan(p).any().item(), torch.isnan(d).any().item(), flush=True)
print('WORST', worst, 'OK' if worst < 1e-4 else 'FAIL')
