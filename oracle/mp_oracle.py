"""
TEST INFRASTRUCTURE ONLY.

CPU restatement ("oracle") of the MultiPoint inference hot path of ethz-asl/multipoint.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product path (``multipoint_amd``) never does and fails loudly without its HIP library.

Every function cites the reference file:line it restates (paths relative to /root/reference).
Floating-point stages are restated with the same ATen CPU ops the reference dispatches to
(torch.nn.functional), integer/index stages in numpy / plain C (oracle/nms_greedy.c).

Pinning status (see tests/test_oracle_vs_reference.py and tests/golden/make_golden.py):
  * forward, depth_to_space, interpolate_descriptors, NNMatcher: PINNED against the imported
    reference (runs only in the build container, where /root/reference exists) and against the
    committed golden vectors generated from it.
  * box_nms (torchvision.ops.nms / batched_nms) and cv2.BFMatcher: third-party code absent from
    /root/reference and not installable here -> "PARITY UNPINNED" at those two boundaries; the
    published algorithms are restated (oracle/nms_greedy.c, ``bf_match_crosscheck``).
  * ``mixed_precision: true`` (forward under torch.cuda.amp.autocast, MultiPoint.py:99-103): PINNED (round 3) against
    tests/golden/forward_f16.npz, which tests/golden/make_golden_f16.py generates by running the imported reference under
    torch.autocast('cpu', dtype=torch.float16) -- the same context manager for the CPU backend; the fixture also records the
    dtype every leaf module returned.  The rounding points are restated in ``_block`` / ``_head`` (fp16 conv in/out with fp32
    accumulation, fp32 BatchNorm arithmetic with fp16 results, fp32 softmax and normalisation); two correct fp16 evaluations
    agree to rounding flips, so the pin is distributional (tests/test_oracle_golden.py, oracle/f16_stats.py).
"""
import collections
import ctypes
import os

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------------------------
# model description (multipoint/models/MultiPoint.py:9-23 default_config, :38-53 channel presets)
# --------------------------------------------------------------------------------------------
DEFAULT_MODEL_CONFIG = {
    'multispectral': True,
    'descriptor_head': True,
    'intepolation_mode': 'bilinear',
    'descriptor_size': 256,
    'normalize_descriptors': True,
    'final_batchnorm': True,
    'reflection_pad': True,
    'bn_first': False,
    'double_convolution': True,
    'channel_version': 0,
    'verbose': False,
    'mixed_precision': False,
    'force_return_logits': False,
}

# the shipped model_weights/multipoint/params.yaml:1-11
SHIPPED_MODEL_CONFIG = {
    'bn_first': False,
    'descriptor_head': True,
    'descriptor_size': 64,
    'final_batchnorm': True,
    'highres_descriptor': False,
    'intepolation_mode': 'bilinear',
    'multispectral': False,
    'normalize_descriptors': True,
    'reflection_pad': True,
    'type': 'MultiPoint',
}


def full_config(cfg=None):
    out = dict(DEFAULT_MODEL_CONFIG)
    if cfg:
        out.update(cfg)
    return out


def _channels(cfg):
    """MultiPoint.py:38-53."""
    v = cfg['channel_version']
    if v == 1:
        return [1, 32, 64, 96, 128], cfg['descriptor_size']
    if v == 2:
        return [1, 8, 16, 32, 64], cfg['descriptor_size']
    return [1, 64, 64, 128, 128], 256


def encoder_layout(cfg):
    """Sequential indices of (conv, bn) modules in generate_encoder (MultiPoint.py:168-185) and
    whether a MaxPool2d(2,2) follows the block.  Only double_convolution=True, bn_first=False or
    True are laid out (indices are the same: pad, conv, X, Y)."""
    ch, _ = _channels(cfg)
    out = []
    idx = 0
    for stage in range(4):
        cin, cout = ch[stage], ch[stage + 1]
        convs = [(cin, cout), (cout, cout)] if cfg['double_convolution'] else [(cin, cout)]
        for k, (ci, co) in enumerate(convs):
            conv_idx = idx + 1
            # getNonlinearity (MultiPoint.py:137-141): (ReLU, BN) or (BN, ReLU)
            bn_idx = idx + 2 if cfg['bn_first'] else idx + 3
            last = (k == len(convs) - 1)
            out.append(dict(conv=conv_idx, bn=bn_idx, cin=ci, cout=co, pool=(last and stage < 3)))
            idx += 4
        if stage < 3:
            idx += 1  # the MaxPool2d module
    return out


def state_dict_spec(cfg=None):
    """Ordered (key, shape, dtype) list of the reference state_dict (train.py:161-173 format)."""
    cfg = full_config(cfg)
    ch, head = _channels(cfg)
    spec = []

    def conv(prefix, co, ci, k):
        spec.append((prefix + '.weight', (co, ci, k, k), torch.float32))
        spec.append((prefix + '.bias', (co,), torch.float32))

    def bn(prefix, c):
        spec.append((prefix + '.weight', (c,), torch.float32))
        spec.append((prefix + '.bias', (c,), torch.float32))
        spec.append((prefix + '.running_mean', (c,), torch.float32))
        spec.append((prefix + '.running_var', (c,), torch.float32))
        spec.append((prefix + '.num_batches_tracked', (), torch.int64))

    enc_names = ['encoder_thermal', 'encoder_optical'] if cfg['multispectral'] else ['encoder']
    for name in enc_names:
        for l in encoder_layout(cfg):
            a, b = sorted([('conv', l['conv']), ('bn', l['bn'])], key=lambda t: t[1])
            for kind, i in (a, b):
                if kind == 'conv':
                    conv('%s.%d' % (name, i), l['cout'], l['cin'], 3)
                else:
                    bn('%s.%d' % (name, i), l['cout'])
    heads = [('detector_head_convolutions', 65)]
    if cfg['descriptor_head']:
        heads.append(('descriptor_head_convolutions', cfg['descriptor_size']))
    for name, nout in heads:
        if cfg['bn_first']:
            conv(name + '.1', head, ch[4], 3); bn(name + '.2', head)
        else:
            conv(name + '.1', head, ch[4], 3); bn(name + '.3', head)
        conv(name + '.4', nout, head, 1)
        if cfg['final_batchnorm']:
            bn(name + '.5', nout)
    return spec


# --------------------------------------------------------------------------------------------
# deterministic synthetic weights / inputs (numpy default_rng so they reproduce on the GPU box)
# --------------------------------------------------------------------------------------------
def make_weights(seed=0, cfg=None, sharpen=True):
    """Seeded synthetic state_dict in the reference key layout (the pretrained blob
    model_weights/multipoint/latest.model is listed in /root/reference/.MISSING_LARGE_BLOBS).
    The value generator is shared data-generation code of the package (multipoint_amd/datasets/synthetic_weights.py,
    which bench.py and the examples use without touching this test infrastructure); the KEY LAYOUT comes from this
    file's own restatement of the reference module tree (state_dict_spec above), so a layout error on either side shows."""
    from multipoint_amd.datasets.synthetic_weights import make_weights_from_spec
    cfg = full_config(cfg)
    return make_weights_from_spec(state_dict_spec(cfg), seed, sharpen, cfg['final_batchnorm'])


def make_images(seed, B, H, W):
    """Grayscale fp32 images uniform[0,1), shape (B,1,H,W)."""
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.random((B, 1, H, W), dtype=np.float32))


# --------------------------------------------------------------------------------------------
# forward (MultiPoint.py:99-185)
# --------------------------------------------------------------------------------------------
def _bn_eval(x, sd, p):
    # nn.BatchNorm2d in eval mode, eps 1e-5 (torch default; MultiPoint.py:139,141)
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'],
                        sd[p + '.weight'], sd[p + '.bias'], training=False, eps=1e-5)


def _pad(x, cfg):
    # MultiPoint.py:33-36: ReflectionPad2d(1) or ZeroPad2d(1)
    return F.pad(x, (1, 1, 1, 1), mode='reflect') if cfg['reflection_pad'] else F.pad(x, (1, 1, 1, 1))


def _h(t):
    """Round to fp16 and back (round-to-nearest-even): the value an fp16 tensor would hold."""
    return t.half().float()


def _block(x, sd, cfg, conv_key, bn_key):
    # MultiPoint.py:143-148 + :137-141
    if cfg.get('mixed_precision'):
        # MultiPoint.py:99-103: forward under torch.cuda.amp.autocast (pinned against the reference run under CPU
        # autocast, tests/golden/forward_f16.npz).  Restated autocast semantics: Conv2d casts input, weight and bias to fp16,
        # accumulates in fp32 and returns fp16; ReLU / pad / max-pool are exact on fp16; BatchNorm2d(eval) on an
        # fp16 tensor evaluates its affine form in fp32 and returns fp16.  x arrives holding fp16 values.
        x = _h(F.conv2d(_pad(x, cfg), _h(sd[conv_key + '.weight']), _h(sd[conv_key + '.bias'])))
        if cfg['bn_first']:
            return F.relu(_h(_bn_eval(x, sd, bn_key)))
        return _h(_bn_eval(F.relu(x), sd, bn_key))
    x = F.conv2d(_pad(x, cfg), sd[conv_key + '.weight'], sd[conv_key + '.bias'])
    if cfg['bn_first']:
        return F.relu(_bn_eval(x, sd, bn_key))
    return _bn_eval(F.relu(x), sd, bn_key)


def encoder(x, sd, cfg, name='encoder'):
    """generate_encoder, MultiPoint.py:168-185."""
    for l in encoder_layout(cfg):
        x = _block(x, sd, cfg, '%s.%d' % (name, l['conv']), '%s.%d' % (name, l['bn']))
        if l['pool']:
            x = F.max_pool2d(x, 2, 2)
    return x


def depth_to_space(x, block_size):
    """multipoint/utils/utils.py:64-69 (== nn.PixelShuffle(8), MultiPoint.py:75):
    out[n, c, h*bs+i, w*bs+j] = x[n, (i*bs+j)*C' + c, h, w] with C' = C/bs^2."""
    N, C, H, W = x.shape
    bs = block_size
    x = x.reshape(N, bs, bs, C // (bs * bs), H, W).permute(0, 3, 4, 1, 5, 2)
    return x.reshape(N, C // (bs * bs), H * bs, W * bs).contiguous()


def _head(x, sd, cfg, name):
    bn_i = 2 if cfg['bn_first'] else 3
    x = _block(x, sd, cfg, name + '.1', '%s.%d' % (name, bn_i))
    if cfg.get('mixed_precision'):
        x = _h(F.conv2d(x, _h(sd[name + '.4.weight']), _h(sd[name + '.4.bias'])))
        if cfg['final_batchnorm']:
            x = _h(_bn_eval(x, sd, name + '.5'))
        return x       # fp16 values; softmax / F.normalize below run in fp32 (autocast's fp32 op list)
    x = F.conv2d(x, sd[name + '.4.weight'], sd[name + '.4.bias'])
    if cfg['final_batchnorm']:
        x = _bn_eval(x, sd, name + '.5')
    return x


def forward(sd, image, cfg=None, is_optical=None, return_logits=False):
    """MultiPoint.forward_impl (MultiPoint.py:106-135) in eval mode.
    image: (B,1,H,W) fp32.  Returns dict(prob (B,1,H,W) | logits (B,65,H/8,W/8), desc (B,D,H/8,W/8))."""
    cfg = full_config(cfg)
    with torch.no_grad():
        if cfg.get('mixed_precision'):
            image = _h(image)             # autocast casts the first convolution's input to fp16
        if cfg['multispectral']:
            # MultiPoint.py:107-122: route each image through encoder_optical / encoder_thermal
            B, _, H, W = image.shape
            ch, _ = _channels(cfg)
            x = torch.zeros((B, ch[4], H // 8, W // 8), dtype=image.dtype)
            opt = is_optical[:, 0].bool()
            if opt.sum() > 0:
                x[opt] = encoder(image[opt], sd, cfg, 'encoder_optical')
            if (~opt).sum() > 0:
                x[~opt] = encoder(image[~opt], sd, cfg, 'encoder_thermal')
        else:
            x = encoder(image, sd, cfg, 'encoder')
        logits = _head(x, sd, cfg, 'detector_head_convolutions')          # MultiPoint.py:150-151
        out = {}
        if return_logits:
            out['prob'], out['logits'] = None, logits
        else:
            prob = torch.softmax(logits, dim=1)                           # nn.Softmax2d, :74,156
            out['prob'] = depth_to_space(prob[:, :-1], 8)                 # :157
            out['logits'] = None
        if cfg['descriptor_head']:
            d = _head(x, sd, cfg, 'descriptor_head_convolutions')         # :160-161
            if cfg['normalize_descriptors']:
                d = F.normalize(d, p=2, dim=1)                            # :163-164 (eps 1e-12)
            out['desc'] = d
        return out


# --------------------------------------------------------------------------------------------
# SuperPointMagicLeap (multipoint/models/SuperPointMagicLeap.py)
# --------------------------------------------------------------------------------------------
MAGICLEAP_LAYERS = (('conv1a', 64, 1, 3), ('conv1b', 64, 64, 3), ('conv2a', 64, 64, 3), ('conv2b', 64, 64, 3),
                    ('conv3a', 128, 64, 3), ('conv3b', 128, 128, 3), ('conv4a', 128, 128, 3), ('conv4b', 128, 128, 3),
                    ('convPa', 256, 128, 3), ('convPb', 65, 256, 1), ('convDa', 256, 128, 3), ('convDb', 256, 256, 1))


def make_weights_magicleap(seed=0):
    """Seeded synthetic SuperPointMagicLeap state_dict (SuperPointMagicLeap.py:16-29 key layout)."""
    rng = np.random.default_rng(seed)
    sd = collections.OrderedDict()
    for name, co, ci, k in MAGICLEAP_LAYERS:
        b = np.sqrt(6.0 / (ci * k * k))
        sd[name + '.weight'] = torch.from_numpy(rng.uniform(-b, b, size=(co, ci, k, k)).astype(np.float32))
        sd[name + '.bias'] = torch.from_numpy(rng.normal(0.0, 0.05, size=(co,)).astype(np.float32))
    sd['convPb.weight'] = sd['convPb.weight'] * 6.0           # sharper detector, dustbin dominant
    bias = sd['convPb.bias'].clone(); bias[64] = 6.0; sd['convPb.bias'] = bias
    return sd


def forward_magicleap(sd, image):
    """SuperPointMagicLeap.forward (:31-66) + generate_heatmap (:68-85).  image (B,1,H,W) fp32."""
    with torch.no_grad():
        def cv(x, n, relu=True):
            k = sd[n + '.weight'].shape[-1]
            y = F.conv2d(x, sd[n + '.weight'], sd[n + '.bias'], padding=k // 2)
            return F.relu(y) if relu else y
        x = cv(cv(image, 'conv1a'), 'conv1b'); x = F.max_pool2d(x, 2, 2)
        x = cv(cv(x, 'conv2a'), 'conv2b'); x = F.max_pool2d(x, 2, 2)
        x = cv(cv(x, 'conv3a'), 'conv3b'); x = F.max_pool2d(x, 2, 2)
        x = cv(cv(x, 'conv4a'), 'conv4b')
        semi = cv(cv(x, 'convPa'), 'convPb', relu=False)
        desc = cv(cv(x, 'convDa'), 'convDb', relu=False)
        dn = torch.norm(desc, p=2, dim=1)
        desc = desc.div(torch.unsqueeze(dn, 1))
        # generate_heatmap: numpy, float32, no max subtraction
        out = torch.zeros(image.shape)
        for i, sample in enumerate(semi):
            dense = np.exp(sample.numpy())
            dense = dense / (np.sum(dense, axis=0) + .00001)
            nodust = dense[:-1].transpose(1, 2, 0)
            Hc, Wc = image.shape[-2] // 8, image.shape[-1] // 8
            hm = np.reshape(nodust, [Hc, Wc, 8, 8]).transpose(0, 2, 1, 3).reshape(Hc * 8, Wc * 8)
            out[i, 0] = torch.from_numpy(hm)
        return {'logits': semi, 'desc': desc, 'prob': out}


# --------------------------------------------------------------------------------------------
# box NMS (multipoint/utils/utils.py:78-122 around torchvision nms)
# --------------------------------------------------------------------------------------------
_nms_lib = None


def _load_nms_lib():
    global _nms_lib
    if _nms_lib is None:
        path = os.path.join(_HERE, '_build', 'libnms_greedy.so')
        if not os.path.exists(path):
            raise RuntimeError('oracle C library missing: run __graft_entry__.build() '
                               '(gcc -O2 -shared -fPIC -o %s oracle/nms_greedy.c)' % path)
        lib = ctypes.CDLL(path)
        lib.oracle_nms_greedy.restype = ctypes.c_int64
        lib.oracle_nms_greedy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                          ctypes.c_double, ctypes.c_void_p]
        _nms_lib = lib
    return _nms_lib


def nms_greedy(boxes, scores, iou):
    """torchvision.ops.nms semantics (oracle/nms_greedy.c). boxes (n,4) f32, scores (n,) f32.
    Returns kept candidate indices in descending-score (stable) order, int64."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    n = scores.shape[0]
    keep = np.empty(max(n, 1), dtype=np.int64)
    nk = _load_nms_lib().oracle_nms_greedy(boxes.ctypes.data, scores.ctypes.data, n,
                                           ctypes.c_double(iou), keep.ctypes.data)
    return keep[:nk].copy()


def nms_greedy_py(boxes, scores, iou):
    """Same algorithm in pure numpy/Python (small cases; cross-checks the C build)."""
    boxes = np.asarray(boxes, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    order = np.argsort(-scores, kind='stable')
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    sup = np.zeros(len(scores), dtype=bool)
    keep = []
    for a, i in enumerate(order):
        if sup[i]:
            continue
        keep.append(i)
        rest = order[a + 1:]
        xx1 = np.maximum(boxes[i, 0], boxes[rest, 0]); yy1 = np.maximum(boxes[i, 1], boxes[rest, 1])
        xx2 = np.minimum(boxes[i, 2], boxes[rest, 2]); yy2 = np.minimum(boxes[i, 3], boxes[rest, 3])
        w = np.maximum(np.float32(0), xx2 - xx1); h = np.maximum(np.float32(0), yy2 - yy1)
        inter = (w * h).astype(np.float32)
        ovr = inter / (area[i] + area[rest] - inter)
        sup[rest[ovr.astype(np.float64) > float(iou)]] = True          # fp32 ovr against the double threshold (nms_greedy.c)
    return np.asarray(keep, dtype=np.int64)


def box_nms(prob, size, min_prob, iou=0.1, keep_top_k=0, use_c=True, dispatch='single'):
    """utils.py:78-122.  prob: numpy (H,W) or (B,1,H,W) fp32.  Returns the dense NMS'ed map.

    dispatch (4-D input only; the RESULT does not depend on it, the run time does): 'single' = one greedy pass over the
    coordinate-offset concatenation of all images, whatever its size; 'torchvision' = what torchvision.ops.batched_nms does on the
    CPU: that single call up to 4000 box coordinates (boxes.numel()), above it one nms call per image (_batched_nms_vanilla) and the
    kept indices ordered by score.  bench.py's cpu_baseline times the second: it is the reference's path.

    :97  points = (prob > min_prob).nonzero()                 (row-major)
    :101 boxes = cat(points[:,2:] - size*0.5, points[:,2:] + size*0.5)   (fp32)
    :103 batched_nms -> coordinate-offset trick: boxes + idx * (boxes.max() + 1), one nms call
    :109-116 per-image top-k = first k indices of that image in descending-score order
    :119-120 scatter kept scores into zeros_like(prob)
    """
    prob = np.asarray(prob, dtype=np.float32)
    if prob.ndim not in (2, 4):
        raise ValueError('The probability must be either 2D (H,W), or 4D (B, 1, H, W)')
    fn = nms_greedy if use_c else nms_greedy_py
    pts = np.argwhere(prob > np.float32(min_prob))
    scores = prob[tuple(pts.T)]
    out = np.zeros_like(prob)
    if pts.shape[0] == 0:
        return out
    half = np.float32(size * 0.5)
    if prob.ndim == 4:
        yx = pts[:, 2:].astype(np.float32)
        boxes = np.concatenate([yx - half, yx + half], axis=1)
        idxs = pts[:, 0]
        if dispatch == 'torchvision' and boxes.size > 4000:
            parts = []
            for b in np.unique(idxs):                               # images never interact: per-image greedy passes
                sel = np.nonzero(idxs == b)[0]
                parts.append(sel[fn(boxes[sel], scores[sel], iou)])
            keep = np.sort(np.concatenate(parts))
            keep = keep[np.argsort(-scores[keep], kind='stable')]   # descending score, ascending index on ties (the stated rule)
        else:
            max_coordinate = boxes.max()
            offsets = idxs.astype(np.float32) * (max_coordinate + np.float32(1))
            keep = fn(boxes + offsets[:, None], scores, iou)
        if keep_top_k > 0:
            sel = [keep[idxs[keep] == b][:keep_top_k] for b in range(prob.shape[0])]
            keep = np.concatenate(sel) if sel else keep
    else:
        yx = pts.astype(np.float32)
        boxes = np.concatenate([yx - half, yx + half], axis=1)
        keep = fn(boxes, scores, iou)
        if keep_top_k > 0:
            keep = keep[:keep_top_k]
    out[tuple(pts[keep].T)] = scores[keep]
    return out


def keypoints_from_map(prob_hw, thr):
    """predict_align_image_pair.py:170-171 / evaluation.py:262-263:
    torch.nonzero((prob.squeeze() > thr).float()) -> (N,2) int64 (y,x), row-major."""
    return np.argwhere(np.asarray(prob_hw) > np.float32(thr)).astype(np.int64)


# --------------------------------------------------------------------------------------------
# descriptor sampling (multipoint/utils/utils.py:159-167)
# --------------------------------------------------------------------------------------------
def interpolate_descriptors(keypoints, desc_lowres, H, W):
    """Manual restatement of F.grid_sample(bilinear, zeros padding, align_corners=True) + normalize.
    keypoints (N,2) int (y,x); desc_lowres numpy (D,Hc,Wc) fp32; all arithmetic in fp32 in the
    order of utils.py:162-163 and ATen's grid_sampler_2d CPU kernel:
        g = kp / (S*0.5) - 1 ; i = ((g + 1) / 2) * (Sc - 1)
        out = nw*w_nw + ne*w_ne + sw*w_sw + se*w_se   (corner weights = products of distances)
    then x / max(||x||_2, 1e-12)."""
    kp = np.asarray(keypoints)
    d = np.asarray(desc_lowres, dtype=np.float32)
    D, Hc, Wc = d.shape
    f = np.float32
    gy = kp[:, 0].astype(f) / f(float(H) * 0.5) - f(1.0)
    gx = kp[:, 1].astype(f) / f(float(W) * 0.5) - f(1.0)
    iy = ((gy + f(1)) / f(2)) * f(Hc - 1)
    ix = ((gx + f(1)) / f(2)) * f(Wc - 1)
    y0 = np.floor(iy); x0 = np.floor(ix)
    y1 = y0 + f(1); x1 = x0 + f(1)
    w_nw = (x1 - ix) * (y1 - iy); w_ne = (ix - x0) * (y1 - iy)
    w_sw = (x1 - ix) * (iy - y0); w_se = (ix - x0) * (iy - y0)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy <= Hc - 1) & (xx >= 0) & (xx <= Wc - 1)
        yi = np.clip(yy, 0, Hc - 1).astype(np.int64); xi = np.clip(xx, 0, Wc - 1).astype(np.int64)
        v = d[:, yi, xi].T                                   # (N,D)
        return np.where(ok[:, None], v, f(0))

    out = (tap(y0, x0) * w_nw[:, None]).astype(f)
    out = out + tap(y0, x1) * w_ne[:, None]
    out = out + tap(y1, x0) * w_sw[:, None]
    out = out + tap(y1, x1) * w_se[:, None]
    out = out.astype(f)
    nrm = np.sqrt((out.astype(f) ** 2).sum(axis=1, dtype=f))
    return (out / np.maximum(nrm, f(1e-12))[:, None]).astype(f)


def interpolate_descriptors_torch(keypoints, desc_lowres, H, W):
    """The same stage through the ATen ops the reference calls (utils.py:159-167), used by the
    CPU-baseline timing leg and to pin ``interpolate_descriptors`` above."""
    kp = torch.as_tensor(np.asarray(keypoints)).float().clone()
    kp[:, 0] = (kp[:, 0] / (float(H) * 0.5)) - 1.0
    kp[:, 1] = (kp[:, 1] / (float(W) * 0.5)) - 1.0
    grid = torch.flip(kp.view(1, 1, -1, 2), [3])
    d = torch.as_tensor(np.asarray(desc_lowres)).unsqueeze(0)
    s = F.grid_sample(d, grid, align_corners=True)[0, :, 0, :].transpose(0, 1)
    return F.normalize(s, p=2, dim=1).numpy()


# --------------------------------------------------------------------------------------------
# matching (multipoint/utils/matching.py)
# --------------------------------------------------------------------------------------------
def nn_match(desc1, desc2, threshold=None):
    """NNMatcher.match, matching.py:41-72.  desc1 (N,D), desc2 (M,D) fp32 unit rows.
    Returns (query_idx int64[K], train_idx int64[K], distance f32[K]) ordered by query index.
    threshold=None disables the ``scores < nn_thresh`` test (:56) -- used to restate
    cv2.BFMatcher(NORM_L2, crossCheck=True) (matching.py:7,31) as plain mutual NN."""
    d1 = np.asarray(desc1, dtype=np.float32); d2 = np.asarray(desc2, dtype=np.float32)
    if d1.shape[0] == 0 or d2.shape[0] == 0:                              # :46-47
        z = np.zeros(0, dtype=np.int64)
        return z, z.copy(), np.zeros(0, dtype=np.float32)
    dmat = np.dot(d1, d2.T)                                               # :50
    dmat = np.sqrt(2 - 2 * np.clip(dmat, -1, 1))                          # :51
    idx = np.argmin(dmat, axis=1)                                         # :53 lowest index wins
    scores = dmat[np.arange(dmat.shape[0]), idx]                          # :54
    keep = np.ones_like(scores, dtype=bool) if threshold is None else scores < threshold   # :56
    idx2 = np.argmin(dmat, axis=0)                                        # :58
    keep_bi = np.arange(len(idx)) == idx2[idx]                            # :59
    keep = np.logical_and(keep, keep_bi)                                  # :60
    q = np.arange(d1.shape[0])[keep]
    return q.astype(np.int64), idx[keep].astype(np.int64), scores[keep].astype(np.float32)


def distance_matrix(desc1, desc2):
    """matching.py:50-51 (the N x M matrix itself; used for near-tie analysis in tests)."""
    d1 = np.asarray(desc1, dtype=np.float32); d2 = np.asarray(desc2, dtype=np.float32)
    return np.sqrt(2 - 2 * np.clip(np.dot(d1, d2.T), -1, 1))


def threshold_match(desc1, desc2, threshold=0.4):
    """ThresholdMatcher.match (matching.py:81-99): every (i, j) with sqrt(2 - 2 clip(d1.d2^T)) < threshold, in
    np.argwhere (row-major) order.  Returns (query_idx, train_idx, distance)."""
    d1 = np.asarray(desc1, dtype=np.float32); d2 = np.asarray(desc2, dtype=np.float32)
    if d1.shape[0] == 0 or d2.shape[0] == 0:                              # :86-87
        z = np.zeros(0, dtype=np.int64)
        return z, z.copy(), np.zeros(0, dtype=np.float32)
    dmat = np.sqrt(2 - 2 * np.clip(np.dot(d1, d2.T), -1, 1))             # :90-91
    idx = np.argwhere(dmat < threshold)                                   # :94
    return idx[:, 0].astype(np.int64), idx[:, 1].astype(np.int64), dmat[idx[:, 0], idx[:, 1]].astype(np.float32)


def bf_knn(desc1, desc2, k=2):
    """cv2.BFMatcher(cv2.NORM_L2).knnMatch(d1, d2, k) (matching.py:21; opencv-python==4.2.0.34, absent -> PARITY
    UNPINNED, published semantics restated): per query row the k nearest train rows under ||a - b||_2, nearest first,
    lower train index first on exact ties (a candidate displaces an entry only when strictly closer).
    Returns (idx (N,k) int64 with -1 padding, dist (N,k) float32)."""
    d1 = np.asarray(desc1, dtype=np.float32); d2 = np.asarray(desc2, dtype=np.float32)
    N, M = d1.shape[0], d2.shape[0]
    idx = np.full((N, k), -1, dtype=np.int64); dist = np.zeros((N, k), dtype=np.float32)
    if N == 0 or M == 0:
        return idx, dist
    diff = d1[:, None, :] - d2[None, :, :]
    dmat = np.sqrt((diff * diff).sum(-1, dtype=np.float32))
    order = np.argsort(dmat, axis=1, kind='stable')[:, :k]
    kk = order.shape[1]
    idx[:, :kk] = order
    dist[:, :kk] = np.take_along_axis(dmat, order, axis=1)
    return idx, dist


def bf_ratio_match(desc1, desc2, ratio_thresh=0.9):
    """get_matches(..., 'bfmatcher', knn_matches=True) (matching.py:20-27): Lowe's ratio test on the two nearest."""
    idx, dist = bf_knn(desc1, desc2, 2)
    keep = (idx[:, 1] >= 0) & (dist[:, 0] < np.float32(ratio_thresh) * dist[:, 1].astype(np.float64))
    q = np.nonzero(keep)[0]
    return q.astype(np.int64), idx[q, 0], dist[q, 0]


def bf_match_crosscheck(desc1, desc2):
    """cv2.BFMatcher(cv2.NORM_L2, crossCheck=True).match(d1, d2) (matching.py:7,31), restated from
    OpenCV's published BFMatcher::knnMatchImpl semantics (opencv-python==4.2.0.34 pinned in
    requirements.txt:1, not installable here -> PARITY UNPINNED): plain Euclidean distance matrix,
    (i,j) kept iff j is i's nearest train descriptor and i is j's nearest query descriptor,
    lowest index on exact ties, DMatch.distance = ||d1_i - d2_j||_2."""
    d1 = np.asarray(desc1, dtype=np.float32); d2 = np.asarray(desc2, dtype=np.float32)
    if d1.shape[0] == 0 or d2.shape[0] == 0:
        z = np.zeros(0, dtype=np.int64)
        return z, z.copy(), np.zeros(0, dtype=np.float32)
    diff = d1[:, None, :] - d2[None, :, :]
    dmat = np.sqrt((diff * diff).sum(-1, dtype=np.float32))
    idx = np.argmin(dmat, axis=1); idx2 = np.argmin(dmat, axis=0)
    keep = np.arange(len(idx)) == idx2[idx]
    q = np.arange(d1.shape[0])[keep]
    return q.astype(np.int64), idx[keep].astype(np.int64), dmat[q, idx[keep]].astype(np.float32)


# --------------------------------------------------------------------------------------------
# whole path on one batch of pairs (evaluation.py:224-285 loop head), used by the CPU baseline
# --------------------------------------------------------------------------------------------
def warp_keypoints(keypoints, homography):
    """multipoint/utils/homographies.py:331-346 with return_type float: cv2.perspectiveTransform on float64 (x,y)
    points (cv2 is absent -> PARITY UNPINNED; OpenCV's perspectiveTransform_64f arithmetic restated:
    w = x*m6 + y*m7 + m8; w = 1/w if |w| > eps else 0; x' = (x*m0 + y*m1 + m2)*w; y' = (x*m3 + y*m4 + m5)*w).
    keypoints (N,2) as (y,x); returns (N,2) float64 as (y,x)."""
    kp = np.asarray(keypoints, dtype=np.float64)
    if len(kp) == 0:
        return kp.reshape(0, 2)
    m = np.asarray(homography, dtype=np.float64).reshape(9)
    y, x = kp[:, 0], kp[:, 1]
    w = x * m[6] + y * m[7] + m[8]
    w = np.where(np.abs(w) > np.finfo(np.float64).eps, 1.0 / np.where(w == 0, 1.0, w), 0.0)
    xo = (x * m[0] + y * m[1] + m[2]) * w
    yo = (x * m[3] + y * m[4] + m[5]) * w
    return np.stack([yo, xo], axis=1)


def descriptor_metrics_pair(kp_optical, kp_thermal, match_query, match_train, h_optical, h_thermal,
                            threshold_keypoints, H, W):
    """Per-sample arithmetic of utils.compute_descriptor_metrics (multipoint/utils/evaluation.py:259,287-328).
    kp_* (N,2) int (y,x); match_query/match_train: mutual matches (optical index, thermal index).
    Returns dict(n_gt_optical, n_gt_thermal, tp_optical (per match), tp_thermal (per match), N_optical, N_thermal)."""
    ho = torch.as_tensor(h_optical, dtype=torch.float32); ht = torch.as_tensor(h_thermal, dtype=torch.float32)
    gt = torch.mm(ht, ho.inverse())                                                   # :259
    kp_o = torch.as_tensor(np.asarray(kp_optical).reshape(-1, 2), dtype=torch.int64)
    kp_t = torch.as_tensor(np.asarray(kp_thermal).reshape(-1, 2), dtype=torch.int64)
    warped_o = warp_keypoints(kp_o.float().numpy(), gt.numpy())                        # :287
    warped_t = warp_keypoints(kp_t.float().numpy(), gt.inverse().numpy())              # :288
    dist = torch.from_numpy(warped_o).unsqueeze(1) - kp_t.unsqueeze(0)                 # :291 (float64)
    correct_o = torch.norm(dist.float(), dim=-1) <= threshold_keypoints                # :292
    dist = torch.from_numpy(warped_t).unsqueeze(1) - kp_o.unsqueeze(0)
    correct_t = torch.norm(dist.float(), dim=-1) <= threshold_keypoints
    q = np.asarray(match_query, dtype=np.int64); t = np.asarray(match_train, dtype=np.int64)

    def inside(pts):                                                                    # filter_points, homographies.py:358-372
        return int(((pts[:, 0] >= 0) & (pts[:, 1] >= 0) & (pts[:, 0] < H) & (pts[:, 1] < W)).sum()) if len(pts) else 0
    return dict(n_gt_optical=int(correct_o.sum(1).nonzero().shape[0]) if correct_o.numel() else 0,
                n_gt_thermal=int(correct_t.sum(1).nonzero().shape[0]) if correct_t.numel() else 0,
                tp_optical=correct_o[q, t].numpy() if len(q) else np.zeros(0, bool),
                tp_thermal=correct_t[t, q].numpy() if len(q) else np.zeros(0, bool),
                N_optical=inside(warped_o), N_thermal=inside(warped_t))


def repeatability_pair(kp_optical, kp_thermal, h_optical, h_thermal, H, W, distance_thresh):
    """Per-sample arithmetic of utils.compute_repeatability_multispectral (multipoint/utils/evaluation.py:165-199);
    warp_keypoints here has its default integer return type (truncation after each warp).
    Returns (count1, count2, N_thermal, N_optical)."""
    ho = torch.as_tensor(h_optical, dtype=torch.float32); ht = torch.as_tensor(h_thermal, dtype=torch.float32)
    kp_o = np.asarray(kp_optical, dtype=np.int64).reshape(-1, 2); kp_t = np.asarray(kp_thermal, dtype=np.int64).reshape(-1, 2)

    def warp_int(kp, h):
        return warp_keypoints(kp, h).astype(int) if len(kp) else kp

    def filt(p):
        return p[(p[:, 0] >= 0) & (p[:, 1] >= 0) & (p[:, 0] < H) & (p[:, 1] < W)] if len(p) else p
    w_o = filt(warp_int(warp_int(kp_o, ho.inverse().numpy()), ht.numpy()))           # :168-170
    w_t = filt(warp_int(warp_int(kp_t, ht.inverse().numpy()), ho.numpy()))           # :173-175
    count1 = count2 = 0
    if len(kp_o) and len(w_t):
        d = np.linalg.norm(w_t[:, None] - kp_o[None], axis=2)                         # :186,191-193
        count1 = int((d.min(axis=1) <= distance_thresh).sum())
    if len(kp_t) and len(w_o):
        d = np.linalg.norm(w_o[:, None] - kp_t[None], axis=2)
        count2 = int((d.min(axis=1) <= distance_thresh).sum())
    return count1, count2, len(w_t), len(w_o)


_M64 = (1 << 64) - 1


def compute_tp_fp_dist(prob, keypoint_map, zero_threshold=1e-4, distance_thresh=2.0):
    """compute_tp_fp_dist (multipoint/utils/evaluation.py:56-97) for one (H,W) map, restated with numpy and the
    reference's sequential loop: predictions = pixels with prob > zero_threshold ranked by prob descending (ties:
    row-major index ascending -- torch.sort leaves them open), ground truth = nonzero label pixels in row-major order,
    matches = float32 distance <= distance_thresh; each ranked prediction takes the FIRST ground-truth point it
    matches and is a true positive iff that point is still unmatched (:84-93).
    Returns (tp, fp, prob_sorted, n_gt, dist[matches])."""
    prob = np.asarray(prob, np.float32)
    kp = np.argwhere(np.asarray(keypoint_map) != 0)                                   # :65
    mask = np.argwhere(prob > np.float32(zero_threshold))                             # :68
    p = prob[mask[:, 0], mask[:, 1]]
    order = np.lexsort((np.arange(len(p)), -p.astype(np.float64)))                    # :71, stated tie-break
    p, pred = p[order], mask[order]
    diff = (pred[:, None, :] - kp[None, :, :]).astype(np.float32)                     # :80
    dist = np.sqrt((diff * diff).sum(-1, dtype=np.float32)).astype(np.float32)        # :81
    matches = dist <= np.float32(distance_thresh)                                     # :82
    tp = []
    matched = np.zeros(len(kp), bool)
    for m in matches:                                                                 # :86-93
        if m.any() and not matched.all():
            g = int(np.argmax(m))
            tp.append(not matched[g])
            matched[g] = True
        else:
            tp.append(False)
    tp = np.array(tp, bool)
    return tp, np.logical_not(tp), p, len(kp), dist[matches]


def detector_precision_recall(tp, fp, prob, n_gt):
    """Tail of compute_detector_metrics (evaluation.py:33-54) on the concatenated per-image lists."""
    sort_idx = np.argsort(prob)[::-1]
    tp, fp, prob = tp[sort_idx], fp[sort_idx], prob[sort_idx]
    tp_cum, fp_cum = np.cumsum(tp), np.cumsum(fp)
    with np.errstate(divide='ignore', invalid='ignore'):
        def div0(a, b):
            c = np.true_divide(a, b)
            idx = ~np.isfinite(c)
            c[idx] = np.where(a[idx] == 0, 1, 0)
            return c
        recall = div0(tp_cum, n_gt)
        precision = div0(tp_cum, tp_cum + fp_cum)
    recall = np.concatenate([[0], recall, [1]])
    precision = np.concatenate([[0], precision, [0]])
    precision = np.maximum.accumulate(precision[::-1])[::-1]
    return precision, recall, prob


def _mix64(z):
    z = (z + 0x9e3779b97f4a7c15) & _M64
    z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & _M64
    z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & _M64
    return z ^ (z >> 31)


def ransac_homography(pts_optical_xy, pts_thermal_xy, reproj_threshold, max_iters, seed, pair_index):
    """CPU restatement of the PRODUCT's RANSAC (multipoint_amd/csrc/homography.hip) -- the stand-in for
    cv2.findHomography(..., cv2.RANSAC, thr) (predict_align_image_pair.py:216; OpenCV absent: PARITY UNPINNED against
    it).  Same counter-based sampling, exact 4-point solve, forward reprojection test, lowest-index tie-break and
    normalised-DLT refit, so the GPU result can be checked for the same winner / inlier set and H to rounding.
    Returns (H 3x3 float64 or None, inlier mask (n,) bool)."""
    a = np.asarray(pts_optical_xy, dtype=np.float32).astype(np.float64).reshape(-1, 2)
    b = np.asarray(pts_thermal_xy, dtype=np.float32).astype(np.float64).reshape(-1, 2)
    n = len(a)
    if n < 4:
        return None, np.zeros(n, bool)
    thr2 = float(reproj_threshold) ** 2

    def inliers(h):
        w = h[6] * a[:, 0] + h[7] * a[:, 1] + h[8]
        ok = np.abs(w) >= 1e-12
        iw = 1.0 / np.where(ok, w, 1.0)
        du = (h[0] * a[:, 0] + h[1] * a[:, 1] + h[2]) * iw - b[:, 0]
        dv = (h[3] * a[:, 0] + h[4] * a[:, 1] + h[5]) * iw - b[:, 1]
        return ok & (du * du + dv * dv <= thr2)

    def hypothesis(t):
        ctr = _mix64((seed ^ (pair_index << 32) ^ t) & _M64)
        idx = []
        while len(idx) < 4:
            ctr = _mix64(ctr)
            c = ctr % n
            if c not in idx:
                idx.append(c)
        m = np.zeros((8, 8)); r = np.zeros(8)
        for k, i in enumerate(idx):
            x, y = a[i]; u, v = b[i]
            m[2 * k] = [x, y, 1, 0, 0, 0, -u * x, -u * y]; r[2 * k] = u
            m[2 * k + 1] = [0, 0, 0, x, y, 1, -v * x, -v * y]; r[2 * k + 1] = v
        try:
            if abs(np.linalg.det(m)) < 1e-300 or np.linalg.cond(m) > 1e13:
                return None
            return np.append(np.linalg.solve(m, r), 1.0)
        except np.linalg.LinAlgError:
            return None
    best_cnt, best_mask = 0, None
    for t in range(max_iters):
        h = hypothesis(t)
        if h is None:
            continue
        mk = inliers(h)
        if mk.sum() > best_cnt:
            best_cnt, best_mask = int(mk.sum()), mk
    if best_cnt < 4:
        return None, np.zeros(n, bool)
    pa, pb = a[best_mask], b[best_mask]
    ca, cb = pa.mean(0), pb.mean(0)
    sa = np.sqrt(2.0) / max(np.sqrt(((pa - ca) ** 2).sum(1)).mean(), 1e-12)
    sb = np.sqrt(2.0) / max(np.sqrt(((pb - cb) ** 2).sum(1)).mean(), 1e-12)
    x, y = ((pa - ca) * sa).T; u, v = ((pb - cb) * sb).T
    o, z = np.ones_like(x), np.zeros_like(x)
    A = np.concatenate([np.stack([x, y, o, z, z, z, -u * x, -u * y, -u], 1), np.stack([z, z, z, x, y, o, -v * x, -v * y, -v], 1)])
    w, vecs = np.linalg.eigh(A.T @ A)
    hn = vecs[:, 0].reshape(3, 3)
    t1 = np.array([[sa, 0, -sa * ca[0]], [0, sa, -sa * ca[1]], [0, 0, 1]])
    t2i = np.array([[1 / sb, 0, cb[0]], [0, 1 / sb, cb[1]], [0, 0, 1]])
    H = t2i @ hn @ t1
    return H / H[2, 2], best_mask


def process_pairs(sd, cfg, optical, thermal, nms=4, detection_threshold=0.015, topk=1000,
                  mask_optical=None, mask_thermal=None):
    """optical/thermal: (P,1,H,W) torch fp32.  Returns per-pair dicts with keypoints, descriptors
    and mutual-NN matches, following evaluation.py:226-282."""
    H, W = optical.shape[2:]
    out_o = forward(sd, optical, cfg, is_optical=torch.ones(optical.shape[0], 1, dtype=torch.bool))
    out_t = forward(sd, thermal, cfg, is_optical=torch.zeros(thermal.shape[0], 1, dtype=torch.bool))
    po = out_o['prob'].numpy(); pt = out_t['prob'].numpy()
    if mask_optical is not None:
        po = po * np.asarray(mask_optical, dtype=np.float32)
    if mask_thermal is not None:
        pt = pt * np.asarray(mask_thermal, dtype=np.float32)
    if nms > 0:
        pt = box_nms(pt, nms, detection_threshold, keep_top_k=topk)
        po = box_nms(po, nms, detection_threshold, keep_top_k=topk)
    res = []
    for i in range(optical.shape[0]):
        kpo = keypoints_from_map(po[i, 0], detection_threshold)
        kpt = keypoints_from_map(pt[i, 0], detection_threshold)
        do = interpolate_descriptors(kpo, out_o['desc'][i].numpy(), H, W)
        dt = interpolate_descriptors(kpt, out_t['desc'][i].numpy(), H, W)
        q, t, dist = nn_match(do, dt, None)
        res.append(dict(kp_optical=kpo, kp_thermal=kpt, desc_optical=do, desc_thermal=dt,
                        match_query=q, match_train=t, match_dist=dist))
    return res
