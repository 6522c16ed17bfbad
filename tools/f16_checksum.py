"""Developer tool (GPU box): bit-level checksum of the fp16-path forward outputs for fixed inputs, used to confirm that a
kernel change leaves every output bit unchanged (compare the printed digests of two builds)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mp_oracle as O          # weight / image generators only
import multipoint_amd.models as models

for mixed in (True, False):
    cfg = dict(O.SHIPPED_MODEL_CONFIG); cfg['mixed_precision'] = mixed
    sd = O.make_weights(0, cfg)
    # negative BatchNorm scales too (gamma < 0): flip the sign of every third channel's weight
    for k in list(sd):
        if k.endswith('.weight') and sd[k].dim() == 1:
            sd[k] = sd[k].clone(); sd[k][::3] *= -1
    net = models.MultiPoint(cfg); net.load_state_dict(sd); net.to('cuda'); net.eval()
    for shape in ((2, 1, 240, 320), (3, 1, 72, 88)):
        out = net({'image': O.make_images(5, *[shape[0], shape[2], shape[3]]).to('cuda')})
        h = hashlib.sha256(out['prob'].cpu().numpy().tobytes() + out['desc'].cpu().numpy().tobytes()).hexdigest()[:16]
        print('mixed_precision=%s %s %s' % (mixed, shape, h))
