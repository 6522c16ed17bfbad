#!/usr/bin/env python3
"""Developer tool (GPU box): timing of the homographic-adaptation kernels and of the whole driver at 480x640.
    python tools/bench_ha.py [--num 65] [--pairs 1]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multipoint_amd.models as M  # noqa: E402
import multipoint_amd.utils as U  # noqa: E402
from multipoint_amd.utils import homographies as PH  # noqa: E402


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--num', type=int, default=65)
    ap.add_argument('--pairs', type=int, default=1)
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=640)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    H, W, B = a.height, a.width, a.pairs
    net = M.MultiPoint({'multispectral': False, 'descriptor_size': 64}).eval()
    net.init_random_weights(0)
    net.to(dev)
    torch.manual_seed(0)
    opt, thr = torch.rand(B, 1, H, W, device=dev), torch.rand(B, 1, H, W, device=dev)
    flags = torch.ones(B, 1, dtype=torch.bool, device=dev)
    data = {'optical': {'image': opt, 'is_optical': flags}, 'thermal': {'image': thr, 'is_optical': ~flags}}
    cfg = {'num': a.num, 'aggregation': 'prod', 'erosion_radius': 3, 'min_count': 5,
           'homographies': {'scaling_amplitude': 0.2, 'perspective_amplitude_x': 0.2, 'perspective_amplitude_y': 0.2,
                            'patch_ratio': 0.85, 'max_angle': 1.57}}
    np.random.seed(0)
    full = PH.dict_update(PH.copy.deepcopy(PH.homography_adaptation_default_config), cfg)
    homs = np.stack([PH.sample_homography(np.array([H, W]), **full['homographies']) for _ in range(a.num - 1)])
    G = min(64 // B, a.num - 1)
    inv = np.linalg.inv(homs[:G])
    px = G * B * H * W
    img64 = torch.rand(G * B, 1, H, W, device=dev)
    ms = timed(lambda: PH._warp(opt, np.repeat(inv, B, 0), G * B, (H, W), 'bilinear', 'reflection'))
    print('warp bilinear/reflection  %d maps: %.3f ms  (%.0f GB/s algorithmic: 4 B read + 4 B written per pixel)' %
          (G * B, ms, px * 8 / ms / 1e6))
    ms = timed(lambda: PH._valid_masks(inv, (H, W), 3, True, dev))
    print('valid mask r=3            %d maps: %.3f ms  (%.0f GB/s of mask written)' % (G, ms, G * H * W / ms / 1e6))
    mask = PH._valid_masks(inv, (H, W), 3, True, dev)
    prob = torch.zeros(B, 1, H, W, device=dev); count = torch.ones_like(prob)
    h = PH._lib.get_handle(dev); hom_d = PH._hom_tensor(homs[:G], dev)
    p = PH._lib.ptr

    def acc():
        h.check(h.lib.mp_ha_accumulate(h.ptr, p(img64), p(img64), p(mask), p(hom_d), G, B, H, W, 1, p(prob), p(count),
                                       PH._lib.stream_ptr(dev)))
    ms = timed(acc)
    print('accumulate (prod)         %d views: %.3f ms  (%.0f GB/s algorithmic: 2 x 4 B + 1 B read per view pixel)' %
          (G, ms, px * 9 / ms / 1e6))
    ms = timed(lambda: net({'image': img64}), 5)
    print('forward                   %d images: %.3f ms' % (G * B, ms))
    ms = timed(lambda: U.homographic_adaptation_multispectral(data, net, cfg, homographies=homs), 3)
    n_fwd = 2 * B * a.num
    print('homographic_adaptation_multispectral  num=%d, %d pair(s): %.1f ms = %.0f views/s (%d forwards of one image)'
          % (a.num, B, ms, n_fwd / ms * 1e3, n_fwd))


if __name__ == '__main__':
    main()
