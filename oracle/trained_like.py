"""TEST INFRASTRUCTURE ONLY (oracle side): inputs and weights with the statistics of a TRAINED network.

The reference's product is a trained net (model_weights/multipoint/latest.model, listed in /root/reference/.MISSING_LARGE_BLOBS
and absent here); every other parity input of this repository is white noise through benign synthetic weights
(multipoint_amd/datasets/synthetic_weights.py: gamma in [0.5,1.5], running_var in [0.5,1.5]).  A Winograd F(4x4,3x3)
convolution (the default 3x3 kernel) amplifies rounding with the dynamic range of its operands, so this module builds the
hard case on purpose:

  * structured_images: piecewise-constant polygons, linear / radial gradients, saturated (clipped) regions, DC offsets,
    periodic texture, sensor noise of different strength -- generated here from half-plane arithmetic, no drawing library
    (it does NOT restate multipoint/datasets/synthetic_dataset/draw_primitives.py).
  * trained_like_weights: starts from the seeded generator, then CALIBRATES every BatchNorm's running statistics on the
    structured images layer by layer in fp64 (what training does), after rescaling each convolution filter so that the
    calibrated running_var is log-uniform over [1e-3, 1e2]; |gamma| log-uniform over [0.1, 10] with 15 % negative
    entries, beta ~ N(0, 0.5); finally a few filters (and their biases) are multiplied by 10 WITHOUT recalibration, so some
    channels run 10x outside their statistics (heavy tails).

Only tests/ and tools/ import this file; the product path never does."""
import numpy as np
import torch
import torch.nn.functional as F

from . import mp_oracle as O


def structured_images(seed, B, H, W):
    """(B,1,H,W) fp32 in [0,1]."""
    rng = np.random.default_rng([int(seed), 77])
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    out = np.empty((B, 1, H, W), dtype=np.float32)
    for b in range(B):
        kind = b % 6
        img = np.full((H, W), rng.uniform(0.05, 0.95))                           # DC level
        if kind in (0, 3, 4, 5):                                                 # a linear or radial ramp underneath
            if rng.uniform() < 0.5:
                a = rng.uniform(0, 2 * np.pi)
                t = (np.cos(a) * xx + np.sin(a) * yy) / np.hypot(H, W)
            else:
                cy, cx = rng.uniform(0, H), rng.uniform(0, W)
                t = np.hypot(yy - cy, xx - cx) / np.hypot(H, W)
            img = img + rng.uniform(-1.5, 1.5) * t
        n_poly = int(rng.integers(4, 14))
        for _ in range(n_poly):                                                  # convex polygons = half-plane intersections
            cy, cx = rng.uniform(0, H), rng.uniform(0, W)
            r = rng.uniform(0.03, 0.35) * min(H, W)
            k = int(rng.integers(3, 7))
            ang = np.sort(rng.uniform(0, 2 * np.pi, size=k))
            inside = np.ones((H, W), dtype=bool)
            for a in ang:
                inside &= (np.cos(a) * (xx - cx) + np.sin(a) * (yy - cy)) < r * rng.uniform(0.5, 1.0)
            img = np.where(inside, rng.uniform(-0.3, 1.3), img)                  # levels beyond [0,1] saturate below
        if kind in (1, 4):                                                       # periodic texture (stripes / checkerboard)
            p = int(rng.integers(2, 9))
            tex = (((xx // p) + (yy // p if rng.uniform() < 0.5 else 0)) % 2) * rng.uniform(0.05, 0.4)
            y0, x0 = int(rng.integers(0, H // 2)), int(rng.integers(0, W // 2))
            img[y0:y0 + H // 2, x0:x0 + W // 2] += tex[y0:y0 + H // 2, x0:x0 + W // 2]
        if kind in (2, 5):                                                       # a dark and a bright saturated half
            img[:, : W // 3] *= 0.1
            img[:, 2 * W // 3:] = img[:, 2 * W // 3:] * 0.2 + 0.9
        noise = (0.0, 0.004, 0.02, 0.0, 0.01, 0.05)[kind]                        # kinds 0 and 3 stay exactly piecewise
        if noise:
            img = img + rng.normal(0.0, noise, size=(H, W))
        out[b, 0] = np.clip(img, 0.0, 1.0).astype(np.float32)
    return torch.from_numpy(out)


def _log_uniform(rng, lo, hi, size):
    return np.exp(rng.uniform(np.log(lo), np.log(hi), size=size))


def trained_like_weights(seed, cfg=None, calib=None, var_range=(1e-3, 1e2), gamma_range=(0.1, 10.0), n_hot=3,
                         hot_factor=10.0, det_keypoint_fraction=0.02):
    """state_dict (reference key layout) whose BatchNorm statistics are calibrated on `calib` images (default: 4 structured
    images 96x128).  Single-encoder, double_convolution configs (the shipped params.yaml family)."""
    cfg = O.full_config(cfg)
    assert not cfg['multispectral'] and not cfg.get('mixed_precision')
    rng = np.random.default_rng([int(seed), 991])
    sd = O.make_weights(seed, cfg, sharpen=False)
    sd = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in sd.items()}
    if calib is None:
        calib = structured_images(seed + 1, 6, 96, 128)
    x = calib.double()

    def calibrate(x, conv_key, bn_key, k3=True):
        """rescale the filter, set running stats from the data, draw gamma/beta; returns the block's output (fp64)."""
        w, b = sd[conv_key + '.weight'], sd[conv_key + '.bias']
        co = w.shape[0]
        xin = O._pad(x, cfg) if k3 else x
        pre = F.conv2d(xin, w, b)
        act = pre if (cfg['bn_first'] or not k3) else F.relu(pre)
        var0 = act.var(dim=(0, 2, 3), unbiased=False).clamp_min(1e-12)
        target = torch.from_numpy(_log_uniform(rng, var_range[0], var_range[1], co))
        s = (target / var0).sqrt()
        sd[conv_key + '.weight'] = w * s[:, None, None, None]
        sd[conv_key + '.bias'] = b * s
        pre = pre * s[None, :, None, None]                       # positively homogeneous: conv and ReLU commute with s > 0
        act = pre if (cfg['bn_first'] or not k3) else F.relu(pre)
        sd[bn_key + '.running_mean'] = act.mean(dim=(0, 2, 3))
        sd[bn_key + '.running_var'] = act.var(dim=(0, 2, 3), unbiased=False)
        g = _log_uniform(rng, gamma_range[0], gamma_range[1], co) * np.where(rng.uniform(size=co) < 0.15, -1.0, 1.0)
        sd[bn_key + '.weight'] = torch.from_numpy(g)
        sd[bn_key + '.bias'] = torch.from_numpy(rng.normal(0.0, 0.5, size=co))
        y = O._bn_eval(act, sd, bn_key)
        return F.relu(y) if (cfg['bn_first'] and k3) else y

    with torch.no_grad():
        for l in O.encoder_layout(cfg):
            x = calibrate(x, 'encoder.%d' % l['conv'], 'encoder.%d' % l['bn'])
            if l['pool']:
                x = F.max_pool2d(x, 2, 2)
        bn_i = 2 if cfg['bn_first'] else 3
        heads = ['detector_head_convolutions'] + (['descriptor_head_convolutions'] if cfg['descriptor_head'] else [])
        for name in heads:
            h = calibrate(x, name + '.1', '%s.%d' % (name, bn_i))
            if cfg['final_batchnorm']:
                h = calibrate(h, name + '.4', name + '.5', k3=False)
            else:
                h = F.conv2d(h, sd[name + '.4.weight'], sd[name + '.4.bias'])
            if name.startswith('detector'):
                # a trained detector: the dustbin wins almost everywhere, a few percent of the pixels exceed the threshold
                if cfg['final_batchnorm']:
                    g = torch.from_numpy(rng.uniform(1.5, 4.0, size=65))
                    sd[name + '.5.weight'] = g
                    beta = torch.from_numpy(rng.normal(0.0, 0.3, size=65))
                    sd[name + '.5.bias'] = beta
                    logits = O._bn_eval(F.conv2d(O._block(x, sd, cfg, name + '.1', '%s.%d' % (name, bn_i)),
                                                 sd[name + '.4.weight'], sd[name + '.4.bias']), sd, name + '.5')
                    lo, hi = 0.0, 40.0
                    for _ in range(30):                           # bisection on the dustbin bias
                        mid = 0.5 * (lo + hi)
                        lg = logits.clone(); lg[:, 64] += mid - beta[64]
                        frac = float((torch.softmax(lg, 1)[:, :-1] > 0.015).double().mean())
                        lo, hi = (mid, hi) if frac > det_keypoint_fraction else (lo, mid)
                    beta[64] = 0.5 * (lo + hi)
                    sd[name + '.5.bias'] = beta
    # a few hot filters, NOT recalibrated: those channels run hot_factor x outside their BatchNorm statistics
    conv_keys = ['encoder.%d' % l['conv'] for l in O.encoder_layout(cfg)] + [n + '.1' for n in heads]
    for key in conv_keys:
        hot = rng.choice(sd[key + '.weight'].shape[0], size=n_hot, replace=False)
        sd[key + '.weight'][hot] *= hot_factor
        sd[key + '.bias'][hot] *= hot_factor
    out = {}
    for k, v in sd.items():
        out[k] = v.float().contiguous() if v.dtype == torch.float64 else v
    return out


def forward64(sd, image, cfg=None):
    """The oracle's forward evaluated in fp64 on the fp32 weights / images: the ground truth every fp32 implementation (ATen
    CPU, direct MFMA, Winograd F(4x4)) is measured against."""
    sd64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in sd.items()}
    return O.forward(sd64, image.double(), cfg)


SEVERITIES = {
    # running_var range, |gamma| range, hot filters per layer
    'benign': None,                                                  # the seeded generator every other test uses
    'mild': dict(var_range=(1e-2, 1e1), gamma_range=(0.3, 3.0), n_hot=0),
    'wide': dict(var_range=(1e-3, 1e2), gamma_range=(0.1, 10.0), n_hot=0),
    'wide+hot': dict(var_range=(1e-3, 1e2), gamma_range=(0.1, 10.0), n_hot=3),
}

# kernel families of the 3x3 layers: the default (conv_wino43.hip where it applies, first block fused), the any-frame-size
# F(4x4,3x3) kernel on every layer (conv_wino43b.hip), the direct implicit-GEMM kernels
VARIANT_ENV = {'F(4x4,3x3)': {}, 'F(4x4,3x3) general': {'MP_DEBUG': 'wino43_gen=2'}, 'direct': {'MP_DEBUG': 'no_winograd'}}


def case(severity, seed, B, H, W, cfg=None):
    """(sd, images, fp32 oracle outputs, fp64 oracle outputs) for one severity; logits included."""
    cfg = dict(cfg or O.SHIPPED_MODEL_CONFIG)
    par = SEVERITIES[severity]
    sd = O.make_weights(seed, cfg) if par is None else trained_like_weights(seed, cfg, **par)
    img = O.make_images(seed + 3, B, H, W) if par is None else structured_images(seed + 3, B, H, W)
    r32 = _with_prob(O.forward(sd, img, cfg, return_logits=True))
    sd64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in sd.items()}
    r64 = _with_prob(O.forward(sd64, img.double(), cfg, return_logits=True))
    return cfg, sd, img, r32, r64


def _with_prob(out):
    """prob from the logits with the ops MultiPoint.forward applies (Softmax2d, drop the dustbin, PixelShuffle(8))."""
    out['prob'] = O.depth_to_space(torch.softmax(out['logits'], dim=1)[:, :-1], 8)
    return out


def gpu_outputs(cfg, sd, img, env):
    """prob / desc / logits of the HIP path with the given kernel-selection environment (a new handle reads it)."""
    import os
    import multipoint_amd.models as M
    old = {k: os.environ.get(k) for k in ('MP_DEBUG',)}
    try:
        for k in old:
            os.environ.pop(k, None)
        os.environ.update(env)
        net = M.MultiPoint(dict(cfg)); net.load_state_dict(sd); net.to('cuda'); net.eval()
        out = net({'image': img.cuda()})
        cfg_l = dict(cfg); cfg_l['force_return_logits'] = True
        net_l = M.MultiPoint(cfg_l); net_l.load_state_dict(sd); net_l.to('cuda'); net_l.eval()
        lg = net_l({'image': img.cuda()})['logits']
        got = {'prob': out['prob'].cpu(), 'desc': out['desc'].cpu(), 'logits': lg.cpu()}
        if img.shape[0] % 2 == 0:
            # the heat map the PRODUCT's drivers extract keypoints from: PairPipeline.run_converged re-evaluates images whose top-k
            # cut fell inside a plateau of tied scores with the tie-exact algorithm (top-k tie guard, include/multipoint_hip.h)
            from multipoint_amd.pipeline import PairPipeline
            pred = {'nms': 4, 'detection_threshold': 0.015, 'topk': 1000,
                    'matching': {'method': 'bfmatcher', 'method_kwargs': {'crossCheck': True}, 'knn_matches': False}}
            pipe = PairPipeline(net, pred, capacity=1000, keep_maps=True)
            res = pipe.run_converged(img.cuda())
            got['prob_tie_robust'] = res.prob.cpu()
            got['tie_redone'] = pipe.tie_redone
        return got
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def errors(got, r32, r64):
    """max abs error of prob / desc / logits against the fp64 truth and against the fp32 CPU oracle."""
    e = {}
    for k in ('prob', 'desc', 'logits'):
        e[k + '_vs_f64'] = float((got[k].double() - r64[k]).abs().max())
        e[k + '_vs_cpu32'] = float((got[k] - r32[k]).abs().max())
    return e
