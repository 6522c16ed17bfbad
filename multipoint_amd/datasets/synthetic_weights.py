"""Seeded synthetic state_dict in the reference key layout (train.py:161-173 saves net.state_dict(); the pretrained blob
model_weights/multipoint/latest.model is not shipped with the reference checkout).  Used by bench.py, the examples and
the tests; numpy default_rng so that the same weights reproduce on every box."""
import collections

import numpy as np
import torch

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


def make_weights_from_spec(spec, seed=0, sharpen=True, final_batchnorm=True):
    """spec: ordered (key, shape, dtype) list (models.MultiPoint.state_dict_spec()).

    Conv weights ~ U(-b, b), b = sqrt(6 / fan_in) (keeps activations O(1) through ReLU); BN: gamma in [0.5,1.5] with ~10 %
    negative entries (so BN must stay *before* the max-pool), beta ~ N(0,0.1), running_mean ~ N(0.2,0.1), running_var in
    [0.5,1.5].  `sharpen`: the final detector BatchNorm2d(65) gets a large gamma and the dustbin a positive beta so that,
    like a trained net, a few thousand pixels exceed detection_threshold = 0.015."""
    rng = np.random.default_rng(seed)
    sd = collections.OrderedDict()
    bn_prefixes = {k.rsplit('.', 1)[0] for k, _, _ in spec if k.endswith('.running_mean')}
    for key, shape, dtype in spec:
        prefix, leaf = key.rsplit('.', 1)
        if dtype == torch.int64:
            sd[key] = torch.tensor(1000, dtype=torch.int64)
            continue
        if len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            b = np.sqrt(6.0 / fan_in)
            arr = rng.uniform(-b, b, size=shape)
        elif leaf == 'bias' and prefix not in bn_prefixes:
            arr = rng.normal(0.0, 0.05, size=shape)            # conv bias
        elif leaf == 'weight':
            arr = rng.uniform(0.5, 1.5, size=shape)
            flip = rng.uniform(size=shape) < 0.1
            arr = np.where(flip, -arr, arr)                    # some negative gammas
        elif leaf == 'bias':
            arr = rng.normal(0.0, 0.1, size=shape)
        elif leaf == 'running_mean':
            arr = rng.normal(0.2, 0.1, size=shape)
        elif leaf == 'running_var':
            arr = rng.uniform(0.5, 1.5, size=shape)
        else:
            raise AssertionError(key)
        sd[key] = torch.from_numpy(arr.astype(np.float32))
    if sharpen and final_batchnorm:
        g = sd['detector_head_convolutions.5.weight']
        sd['detector_head_convolutions.5.weight'] = (g.abs() * 2.5).contiguous()
        b = sd['detector_head_convolutions.5.bias'].clone()
        b[64] = 11.0                                           # dustbin dominates most cells
        sd['detector_head_convolutions.5.bias'] = b
    return sd


def make_weights(seed=0, cfg=None, sharpen=True):
    """Seeded state_dict for a MultiPoint model config (default: the shipped params.yaml)."""
    from ..models import MultiPoint
    net = MultiPoint(dict(cfg) if cfg is not None else dict(SHIPPED_MODEL_CONFIG))
    return make_weights_from_spec(net.state_dict_spec(), seed, sharpen, net.config['final_batchnorm'])


def make_images(seed, B, H, W):
    """Grayscale fp32 images uniform[0,1), shape (B,1,H,W)."""
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.random((B, 1, H, W), dtype=np.float32))
