#!/usr/bin/env python3
"""Generates tests/golden/*.npz by IMPORTING the reference (/root/reference) on PyTorch-CPU.
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixtures are data (inputs + the reference's outputs), never reference source:
  forward_64x64.npz      full prob/desc of MultiPoint.forward on a 2x1x64x64 batch (shipped params.yaml)
  forward_240x320.npz    512 sampled (index, value) pairs of prob/desc + fp64 checksums (BASELINE configs[0] shape)
  forward_variants.npz   multispectral / zero-pad / bn_first / descriptor_size 256 variants at 32x48
  forward_magicleap.npz  SuperPointMagicLeap.forward (logits, desc, heat map) at 2x1x32x48
  sampling.npz           utils.interpolate_descriptors(kp, desc, H, W) rows
  matcher.npz            NNMatcher(0.7).match(d1, d2) (query, train, distance) lists
  depth_to_space.npz     utils.depth_to_space / space_to_depth
Weights/images come from oracle.mp_oracle.make_weights/make_images (numpy default_rng): the fixtures
store only seeds for them, so they reproduce anywhere.
"""
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
torch.manual_seed(0)


def ref_forward(cfg, sd, img, is_optical=None):
    net = models.MultiPoint(dict(cfg)).eval()
    net.load_state_dict(sd)
    data = {'image': img}
    if is_optical is not None:
        data['is_optical'] = is_optical
    with torch.no_grad():
        return net(data)


def main():
    cfg = dict(O.SHIPPED_MODEL_CONFIG)
    sd = O.make_weights(0, cfg)

    img = O.make_images(11, 2, 64, 64)
    out = ref_forward(cfg, sd, img)
    np.savez_compressed(os.path.join(HERE, 'forward_64x64.npz'), weight_seed=0, image_seed=11, B=2, H=64, W=64,
                        prob=out['prob'].numpy(), desc=out['desc'].numpy())

    img = O.make_images(12, 1, 240, 320)
    out = ref_forward(cfg, sd, img)
    rng = np.random.default_rng(99)
    p = out['prob'].numpy().ravel(); d = out['desc'].numpy().ravel()
    pi = rng.choice(p.size, 512, replace=False); di = rng.choice(d.size, 512, replace=False)
    np.savez_compressed(os.path.join(HERE, 'forward_240x320.npz'), weight_seed=0, image_seed=12, B=1, H=240, W=320,
                        prob_idx=pi, prob_val=p[pi], desc_idx=di, desc_val=d[di],
                        prob_sum=np.float64(p.astype(np.float64).sum()), desc_abs_sum=np.float64(np.abs(d.astype(np.float64)).sum()),
                        n_above_thr=int((p > 0.015).sum()))

    variants = {}
    for name, upd in [('multispectral', {'multispectral': True}), ('zero_pad', {'reflection_pad': False}),
                      ('bn_first', {'bn_first': True}), ('desc256', {'descriptor_size': 256}),
                      ('no_final_bn', {'final_batchnorm': False}), ('no_normalize', {'normalize_descriptors': False}),
                      ('single_conv', {'double_convolution': False}),
                      ('single_conv_ms_zero_pad', {'double_convolution': False, 'multispectral': True, 'reflection_pad': False,
                                                   'bn_first': True})]:
        c = dict(cfg); c.update(upd)
        s = O.make_weights(3, c)
        im = O.make_images(13, 3, 32, 48)
        flags = torch.tensor([[True], [False], [True]])
        o = ref_forward(c, s, im, flags)
        variants[name + '_prob'] = o['prob'].numpy(); variants[name + '_desc'] = o['desc'].numpy()
    # force_return_logits path
    net = models.MultiPoint(dict(cfg)).eval(); net.load_state_dict(sd); net.set_force_return_logits(True)
    with torch.no_grad():
        lo = net({'image': O.make_images(13, 3, 32, 48)})
    variants['logits'] = lo['logits'].numpy()
    np.savez_compressed(os.path.join(HERE, 'forward_variants.npz'), weight_seed=3, image_seed=13, **variants)

    # SuperPointMagicLeap (second model.type of the reference)
    ml = models.SuperPointMagicLeap().eval()
    msd = O.make_weights_magicleap(4)
    assert list(ml.state_dict().keys()) == list(msd.keys())
    ml.load_state_dict(msd)
    im = O.make_images(14, 2, 32, 48)
    with torch.no_grad():
        mo = ml({'image': im})
    np.savez_compressed(os.path.join(HERE, 'forward_magicleap.npz'), weight_seed=4, image_seed=14,
                        logits=mo['logits'].numpy(), desc=mo['desc'].numpy(), prob=mo['prob'].numpy())

    # sampling: keypoints incl. borders / corners
    rng = np.random.default_rng(5)
    H, W, D, Hc, Wc = 240, 320, 64, 30, 40
    desc = rng.standard_normal((D, Hc, Wc)).astype(np.float32)
    kp = np.stack([rng.integers(0, H, 200), rng.integers(0, W, 200)], 1).astype(np.int64)
    kp[:6] = [[0, 0], [H - 1, W - 1], [0, W - 1], [H - 1, 0], [H // 2, W - 1], [H - 1, W // 2]]
    rows = utils.interpolate_descriptors(torch.from_numpy(kp), torch.from_numpy(desc), H, W).numpy()
    np.savez_compressed(os.path.join(HERE, 'sampling.npz'), H=H, W=W, desc=desc, keypoints=kp, rows=rows)

    # matcher
    d1 = rng.standard_normal((150, 64)).astype(np.float32); d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    d2 = np.concatenate([d1[:100] + 0.05 * rng.standard_normal((100, 64)).astype(np.float32),
                         rng.standard_normal((70, 64)).astype(np.float32)])
    d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
    d2 = d2[rng.permutation(len(d2))]
    m = utils.NNMatcher(0.7).match(d1, d2)
    np.savez_compressed(os.path.join(HERE, 'matcher.npz'), d1=d1, d2=d2, threshold=0.7,
                        query=np.array([x.queryIdx for x in m]), train=np.array([x.trainIdx for x in m]),
                        distance=np.array([x.distance for x in m], dtype=np.float32))

    x = torch.from_numpy(rng.standard_normal((2, 64, 3, 5)).astype(np.float32))
    np.savez_compressed(os.path.join(HERE, 'depth_to_space.npz'), x=x.numpy(), d2s=utils.depth_to_space(x, 8).numpy(),
                        s2d=utils.space_to_depth(utils.depth_to_space(x, 8), 8).numpy())
    print('golden fixtures written to', HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print('  %-24s %7.1f KB' % (f, os.path.getsize(os.path.join(HERE, f)) / 1024))


if __name__ == '__main__':
    main()
