#!/usr/bin/env python3
"""Generates tests/golden/forward_f16.npz by IMPORTING the reference (/root/reference) with `mixed_precision: true`.
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_f16.py

MultiPoint.forward (multipoint/models/MultiPoint.py:99-104) wraps forward_impl in `torch.cuda.amp.autocast()`, which needs a
CUDA device.  The SAME context manager exists for the CPU backend: the generator substitutes
`torch.autocast('cpu', dtype=torch.float16)` for the attribute the reference looks up -- the reference's own module code
then runs under autocast on PyTorch-CPU (Conv2d -> fp16 out, ReLU / MaxPool on fp16, BatchNorm2d(eval) -> fp16 out,
Softmax2d / F.normalize -> fp32; observed with forward hooks, see `dtypes` in the fixture).  The fixture is data: seeds,
the reference's outputs (logits / prob / desc) and the dtype every leaf module returned."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
from oracle import mp_oracle as O  # noqa: E402

models, utils = ref_shim.install()
torch.cuda.amp.autocast = lambda *a, **k: torch.autocast('cpu', dtype=torch.float16)


def ref_forward(cfg, sd, img, logits=False, hooks=None):
    net = models.MultiPoint(dict(cfg)).eval()
    net.load_state_dict(sd)
    if logits:
        net.set_force_return_logits(True)
    if hooks is not None:
        for n, m in net.named_modules():
            if len(list(m.children())) == 0:
                m.register_forward_hook(lambda mod, i, o, n=n: hooks.__setitem__(n, [type(mod).__name__, str(o.dtype)]))
    with torch.no_grad():
        return net({'image': img})


def main():
    cfg = dict(O.SHIPPED_MODEL_CONFIG); cfg['mixed_precision'] = True
    sd = O.make_weights(0, cfg)
    out = {}
    dtypes = {}
    img = O.make_images(11, 2, 64, 64)
    o = ref_forward(cfg, sd, img, hooks=dtypes)
    l = ref_forward(cfg, sd, img, logits=True)
    out.update(a_seed=11, a_B=2, a_H=64, a_W=64, a_prob=o['prob'].float().numpy(), a_desc=o['desc'].float().numpy(),
               a_logits=l['logits'].float().numpy())
    img = O.make_images(12, 1, 240, 320)
    o = ref_forward(cfg, sd, img)
    l = ref_forward(cfg, sd, img, logits=True)
    rng = np.random.default_rng(98)
    p = o['prob'].float().numpy().ravel(); d = o['desc'].float().numpy(); lg = l['logits'].float().numpy().ravel()
    pi = np.concatenate([rng.choice(p.size, 3072, replace=False), np.flatnonzero(p > 0.015)[:1024]])
    li = rng.choice(lg.size, 4096, replace=False)
    cells = rng.choice(d.shape[2] * d.shape[3], 64, replace=False)              # whole descriptors (all 64 channels) of 64 cells
    out.update(b_seed=12, b_B=1, b_H=240, b_W=320, b_prob_idx=pi, b_prob_val=p[pi], b_logits_idx=li, b_logits_val=lg[li],
               b_desc_cells=cells, b_desc_val=d.reshape(1, d.shape[1], -1)[0][:, cells],
               b_n_above_thr=int((p > 0.015).sum()))
    # the other model configs MultiPoint.forward wraps in autocast just the same (MultiPoint.py:99-104 does not look at the
    # config): c = channel_version 1 ([1,32,64,96,128], heads = descriptor_size), d = channel_version 2 with one convolution per
    # stage (double_convolution: false) -- full maps at 2 x 64 x 64, the config update as JSON
    for c, upd, seed in (('c', {'channel_version': 1}, 13), ('d', {'channel_version': 2, 'double_convolution': False}, 14)):
        cfg_v = dict(cfg); cfg_v.update(upd)
        sd_v = O.make_weights(0, cfg_v)
        img = O.make_images(seed, 2, 64, 64)
        o = ref_forward(cfg_v, sd_v, img)
        l = ref_forward(cfg_v, sd_v, img, logits=True)
        out.update({c + '_seed': seed, c + '_B': 2, c + '_H': 64, c + '_W': 64, c + '_cfg': json.dumps(upd),
                    c + '_prob': o['prob'].float().numpy(), c + '_desc': o['desc'].float().numpy(),
                    c + '_logits': l['logits'].float().numpy()})
    np.savez_compressed(os.path.join(HERE, 'forward_f16.npz'), weight_seed=0, dtypes=json.dumps(dtypes), **out)
    print(json.dumps(dtypes, indent=0)[:600])


if __name__ == '__main__':
    main()
