"""Host-side mirror of multipoint/models/SuperPointMagicLeap.py (the MagicLeap SuperPoint network the
reference can select with `model.type: SuperPointMagicLeap`): same layer shapes as MultiPoint's
channel_version 0, but zero padding, no BatchNorm, D = 256, named convolutions, and a heat map computed
as exp(x) / (sum exp(x) + 1e-5) (generate_heatmap, SuperPointMagicLeap.py:68-85).  Runs on the same
gfx950 kernels through the same C ABI; forward returns 'logits', 'desc' and 'prob' like the reference."""
import torch

from .. import _lib
from .multipoint import MultiPoint


class SuperPointMagicLeap(MultiPoint):
    default_config = {}

    # the reference class takes a config argument and ignores it (SuperPointMagicLeap.py:10)
    _fixed = {'multispectral': False, 'descriptor_head': True, 'descriptor_size': 256,
              'normalize_descriptors': True, 'final_batchnorm': False, 'reflection_pad': False,
              'bn_first': False, 'double_convolution': True, 'channel_version': 0, 'verbose': False,
              'mixed_precision': False, 'force_return_logits': False}
    _abi_extra = {'batchnorm': 0, 'key_layout': 1, 'softmax_mode': 1}

    def __init__(self, config=None):
        fixed = dict(self._fixed)
        for k in ('conv_algorithm', 'batch_invariant'):                   # the multipoint_amd-only settings pass through
            if isinstance(config, dict) and k in config:
                fixed[k] = config[k]
        super().__init__(fixed)
        self.user_config = config

    def state_dict_spec(self):
        """SuperPointMagicLeap.py:16-29."""
        spec = []
        for name, co, ci, k in (('conv1a', 64, 1, 3), ('conv1b', 64, 64, 3), ('conv2a', 64, 64, 3), ('conv2b', 64, 64, 3),
                                ('conv3a', 128, 64, 3), ('conv3b', 128, 128, 3), ('conv4a', 128, 128, 3),
                                ('conv4b', 128, 128, 3), ('convPa', 256, 128, 3), ('convPb', 65, 256, 1),
                                ('convDa', 256, 128, 3), ('convDb', 256, 256, 1)):
            spec.append((name + '.weight', (co, ci, k, k), torch.float32))
            spec.append((name + '.bias', (co,), torch.float32))
        return spec

    def forward(self, data):
        """SuperPointMagicLeap.forward (:31-66): {'logits' (B,65,H/8,W/8), 'desc' (B,256,H/8,W/8), 'prob' (B,1,H,W)}."""
        if self._state is None:
            raise RuntimeError('SuperPointMagicLeap has no weights: call load_state_dict() or init_random_weights()')
        image = data['image']
        if self.device is None:
            self.to(image.device)
        if image.device != self.device:
            raise RuntimeError('input image is on %s but the model is on %s' % (image.device, self.device))
        if image.dim() != 4 or image.shape[1] != 1:
            raise ValueError('image must have shape (B,1,H,W), got %s' % (tuple(image.shape),))
        image = image.to(torch.float32).contiguous()
        B, _, H, W = image.shape
        Hc, Wc = H // 8, W // 8
        prob = torch.empty((B, 1, H, W), dtype=torch.float32, device=self.device)
        logits = torch.empty((B, 65, Hc, Wc), dtype=torch.float32, device=self.device)
        desc_cl = torch.empty((B, Hc, Wc, 256), dtype=torch.float32, device=self.device)
        h = self._handle
        with torch.cuda.device(self.device):
            h.check(h.lib.mp_forward(h.ptr, _lib.ptr(image), None, B, H, W, _lib.ptr(prob), _lib.ptr(logits),
                                     _lib.ptr(desc_cl), _lib.stream_ptr(self.device)))
        return {'logits': logits, 'desc': desc_cl.permute(0, 3, 1, 2), 'prob': prob}

    __call__ = forward
