"""Host-side mirror of the reference model class (multipoint/models/MultiPoint.py:8-185) for
inference: same constructor config, same state_dict key layout, same forward dict-in/dict-out
contract -- computed by hand-written gfx950 kernels behind libmultipoint_hip.so."""
import collections
import copy
import ctypes

import numpy as np
import torch

from .. import _lib
from ..utils.utils import dict_update


class MultiPoint:
    # multipoint/models/MultiPoint.py:9-23
    default_config = {
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
        'force_return_logits': False
    }
    # multipoint_amd only (no reference counterpart, hence not in default_config): optional key model.conv_algorithm, the algorithm of
    # the 3x3 convolutions -- 'auto' | 'winograd43' | 'winograd43_general' | 'direct' (include/multipoint_hip.h:
    # mp_model_config.conv_algorithm)
    CONV_ALGORITHMS = {'auto': 0, 'winograd43': 1, 'winograd43_general': 2, 'direct': 3}
    # ... and optional key model.batch_invariant (bool, default False): forwards of one or two images skip the split small launches
    # of the single-pair latency path, so that their bits equal those of the same images inside a larger batch
    # (mp_model_config.batch_invariant)

    def __init__(self, config=None):
        if config:
            self.config = dict_update(copy.deepcopy(self.default_config), config)
        else:
            self.config = copy.deepcopy(self.default_config)
        if self.config['channel_version'] not in (0, 1, 2):
            raise ValueError('channel_version must be 0, 1 or 2 (MultiPoint.py:38-53)')
        if self.config.get('conv_algorithm', 'auto') not in self.CONV_ALGORITHMS:
            raise ValueError('conv_algorithm must be one of %s' % ', '.join(self.CONV_ALGORITHMS))
        self.training = False
        self.device = None
        self._handle = None
        self._state = None           # OrderedDict name -> CPU tensor (reference state_dict layout)
        self._uploaded = False
        self._fwd_event = None       # last forward's completion, for callers that alternate streams
        self._fwd_stream = None
        if self.config['verbose']:
            n = sum(int(np.prod(s)) for k, s, d in self.state_dict_spec() if d == torch.float32
                    and not k.endswith(('running_mean', 'running_var')))
            print('MultiPoint number of trainable parameter: ' + str(n))

    # -- state_dict layout (train.py:161-173 saves net.state_dict()) ------------------------------
    def state_dict_spec(self):
        """Ordered (key, shape, dtype) of the reference nn.Module's state_dict for this config
        (nn.Sequential indices of generate_encoder, MultiPoint.py:168-185, and the two heads,
        :62-88)."""
        c = self.config
        spec = []

        def conv(p, co, ci, k):
            spec.append((p + '.weight', (co, ci, k, k), torch.float32))
            spec.append((p + '.bias', (co,), torch.float32))

        def bn(p, ch):
            for leaf in ('weight', 'bias', 'running_mean', 'running_var'):
                spec.append(('%s.%s' % (p, leaf), (ch,), torch.float32))
            spec.append((p + '.num_batches_tracked', (), torch.int64))

        # MultiPoint.py:38-53
        stage = {0: [1, 64, 64, 128, 128], 1: [1, 32, 64, 96, 128], 2: [1, 8, 16, 32, 64]}[c['channel_version']]
        head_channels = 256 if c['channel_version'] == 0 else c['descriptor_size']
        if c['double_convolution']:
            chan = [1, stage[1], stage[1], stage[2], stage[2], stage[3], stage[3], stage[4], stage[4]]
            conv_idx = [1, 5, 10, 14, 19, 23, 28, 32]
        else:       # getConvolutionBlock (MultiPoint.py:147-148): one (pad, conv, X, Y) group per stage, a pool after stages 1-3
            chan = list(stage)
            conv_idx = [1, 6, 11, 16]
        bn_off = 1 if c['bn_first'] else 2
        names = ['encoder_thermal', 'encoder_optical'] if c['multispectral'] else ['encoder']
        for name in names:
            for i, ci in enumerate(conv_idx):
                conv('%s.%d' % (name, ci), chan[i + 1], chan[i], 3)
                bn('%s.%d' % (name, ci + bn_off), chan[i + 1])
        heads = [('detector_head_convolutions', 65)]
        if c['descriptor_head']:
            heads.append(('descriptor_head_convolutions', c['descriptor_size']))
        for name, nout in heads:
            conv(name + '.1', head_channels, stage[4], 3)
            bn('%s.%d' % (name, 1 + bn_off), head_channels)
            conv(name + '.4', nout, head_channels, 1)
            if c['final_batchnorm']:
                bn(name + '.5', nout)
        return spec

    def load_state_dict(self, state_dict, strict=True):
        """nn.Module.load_state_dict semantics (strict by default), predict_align_image_pair.py:58-61."""
        spec = self.state_dict_spec()
        expected = collections.OrderedDict((k, (s, d)) for k, s, d in spec)
        missing = [k for k in expected if k not in state_dict]
        unexpected = [k for k in state_dict if k not in expected]
        errors = []
        if strict and missing:
            errors.append('Missing key(s) in state_dict: %s.' % ', '.join('"%s"' % k for k in missing))
        if strict and unexpected:
            errors.append('Unexpected key(s) in state_dict: %s.' % ', '.join('"%s"' % k for k in unexpected))
        new_state = collections.OrderedDict() if self._state is None else collections.OrderedDict(self._state)
        for k, (shape, dtype) in expected.items():
            if k not in state_dict:
                continue
            v = state_dict[k]
            if not torch.is_tensor(v):
                v = torch.as_tensor(v)
            if tuple(v.shape) != tuple(shape):
                errors.append('size mismatch for %s: copying a param with shape %s from checkpoint, '
                              'the shape in current model is %s.' % (k, tuple(v.shape), tuple(shape)))
                continue
            new_state[k] = v.detach().to('cpu', dtype).contiguous().clone()
        if errors:
            raise RuntimeError('Error(s) in loading state_dict for %s:\n\t' % type(self).__name__ + '\n\t'.join(errors))
        if strict or all(k in new_state for k in expected):
            self._state = collections.OrderedDict((k, new_state[k]) for k in expected)
            self._uploaded = False
            if self.device is not None:
                self._upload()
        return missing, unexpected

    def state_dict(self):
        if self._state is None:
            raise RuntimeError('MultiPoint has no weights: call load_state_dict() or init_random_weights()')
        return collections.OrderedDict((k, v.clone()) for k, v in self._state.items())

    def init_random_weights(self, seed=0):
        """Stand-in for nn.Module's default initialisation when the CLI runs with `-v none`
        (predict_align_image_pair.py:23,58): Kaiming-uniform convs, identity BatchNorm."""
        rng = np.random.default_rng(seed)
        sd = collections.OrderedDict()
        bounds = {}
        for k, shape, dtype in self.state_dict_spec():
            prefix, leaf = k.rsplit('.', 1)
            if dtype == torch.int64:
                sd[k] = torch.zeros((), dtype=torch.int64)
            elif len(shape) == 4:
                bounds[prefix] = 1.0 / np.sqrt(shape[1] * shape[2] * shape[3])
                sd[k] = torch.from_numpy(rng.uniform(-bounds[prefix], bounds[prefix], shape).astype(np.float32))
            elif leaf == 'bias' and prefix in bounds:
                sd[k] = torch.from_numpy(rng.uniform(-bounds[prefix], bounds[prefix], shape).astype(np.float32))
            elif leaf in ('running_var', 'weight'):
                sd[k] = torch.ones(shape)
            else:
                sd[k] = torch.zeros(shape)
        self.load_state_dict(sd)
        return self

    # -- nn.Module-like plumbing ------------------------------------------------------------------
    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError('multipoint_amd implements the inference path only (eval mode)')
        return self.eval()

    def to(self, device):
        device = _lib.require_cuda(device)
        if self.device != device or self._handle is None:
            self.device = device
            self._handle = _lib.Handle(device.index)
            self._uploaded = False
        if self._state is not None and not self._uploaded:
            self._upload()
        return self

    def direct_twin(self):
        """A second model over the SAME weights with `conv_algorithm: direct` (created on first use, on this model's device) --
        the tie-exact algorithm the top-k tie guard re-evaluates flagged images with (utils.tie_robust_redo).  None when this
        model has no other algorithm to offer: it is `direct` itself, or runs the fp16 path (`mixed_precision`)."""
        if self.config.get('conv_algorithm', 'auto') == 'direct' or self.config.get('mixed_precision') or self._state is None \
                or self.device is None:
            return None
        twin = getattr(self, '_direct_twin', None)
        if twin is None or twin._state_of is not self._state or twin.device != self.device:
            cfg = dict(self.config); cfg['conv_algorithm'] = 'direct'
            twin = type(self)(cfg)
            twin._state = self._state                    # shared host tensors (read-only)
            twin._state_of = self._state
            twin._uploaded = False
            twin.to(self.device)
            twin.eval()
            self._direct_twin = twin
        return twin

    def cuda(self, device=None):
        return self.to(torch.device('cuda', torch.cuda.current_device() if device is None else device))

    def set_force_return_logits(self, value):
        # MultiPoint.py:93-97
        if not isinstance(value, bool):
            raise ValueError('set_force_return_logits: The input value needs to be a bool')
        self.config['force_return_logits'] = value

    # extra mp_model_config fields of this model class (include/multipoint_hip.h)
    _abi_extra = {'batchnorm': 1, 'key_layout': 0, 'softmax_mode': 0}

    def _upload(self):
        c = self.config
        vals = [int(c[k]) if k in ('descriptor_size', 'channel_version') else int(bool(c[k]))
                for k in ('multispectral', 'descriptor_head', 'descriptor_size', 'normalize_descriptors',
                          'final_batchnorm', 'reflection_pad', 'bn_first', 'double_convolution', 'channel_version')]
        vals += [self._abi_extra['batchnorm'], self._abi_extra['key_layout'], self._abi_extra['softmax_mode'],
                 int(bool(c['mixed_precision'])),      # MultiPoint.py:99-103: forward under autocast -> fp16 MFMA path
                 self.CONV_ALGORITHMS[c.get('conv_algorithm', 'auto')], int(bool(c.get('batch_invariant', False)))]
        cfg = _lib.ModelConfig(*vals)
        keep = []
        arr = []
        for k, v in self._state.items():
            if v.dtype != torch.float32:
                continue                       # num_batches_tracked counters are not used in eval mode
            keep.append(v)
            arr.append(_lib.Tensor(k.encode(), ctypes.c_void_p(v.data_ptr()), v.numel()))
        tens = (_lib.Tensor * len(arr))(*arr)
        h = self._handle
        h.check(h.lib.mp_load_weights(h.ptr, ctypes.byref(cfg), tens, len(arr)))
        self._uploaded = True

    # -- forward (MultiPoint.py:99-135) -----------------------------------------------------------
    def forward(self, data):
        if self._state is None:
            raise RuntimeError('MultiPoint has no weights: call load_state_dict() or init_random_weights()')
        if self.training:
            raise NotImplementedError('multipoint_amd implements the inference path only (eval mode)')
        image = data['image']
        if self.device is None:
            self.to(image.device)
        if image.device != self.device:
            raise RuntimeError('input image is on %s but the model is on %s' % (image.device, self.device))
        if image.dim() != 4 or image.shape[1] != 1:
            raise ValueError('image must have shape (B,1,H,W), got %s' % (tuple(image.shape),))
        image = image.to(torch.float32).contiguous()
        B, _, H, W = image.shape
        is_opt = None
        if self.config['multispectral']:
            flags = data['is_optical'][:, 0].to('cpu', torch.uint8).contiguous()
            is_opt = flags
        Hc, Wc = H // 8, W // 8
        want_logits = bool(self.config['force_return_logits'])
        prob = None if want_logits else torch.empty((B, 1, H, W), dtype=torch.float32, device=self.device)
        logits = torch.empty((B, 65, Hc, Wc), dtype=torch.float32, device=self.device) if want_logits else None
        desc_cl = None
        if self.config['descriptor_head']:
            D = self.config['descriptor_size']
            desc_cl = torch.empty((B, Hc, Wc, D), dtype=torch.float32, device=self.device)
        h = self._handle
        # one workspace per handle: a forward on ANOTHER stream than the previous one is ordered behind it (same stream: in order anyway)
        cur = torch.cuda.current_stream(self.device)
        if self._fwd_event is not None and self._fwd_stream != cur:
            cur.wait_event(self._fwd_event)
        with torch.cuda.device(self.device):
            h.check(h.lib.mp_forward(h.ptr, _lib.ptr(image),
                                     ctypes.c_void_p(is_opt.data_ptr()) if is_opt is not None else None,
                                     B, H, W, _lib.ptr(prob), _lib.ptr(logits), _lib.ptr(desc_cl),
                                     _lib.stream_ptr(self.device)))
        if self._fwd_stream != cur:                  # (the event is only needed across a change of stream: record on change)
            self._fwd_event = torch.cuda.Event()
        if self._fwd_event is not None:
            self._fwd_event.record(cur)
        self._fwd_stream = cur
        out = {'prob': prob, 'logits': logits}
        if desc_cl is not None:
            # logically (B,D,Hc,Wc) like the reference, stored channels-last (values identical)
            out['desc'] = desc_cl.permute(0, 3, 1, 2)
        return out

    __call__ = forward

    # profiling hook used by bench.py (per-launch hipEvent timing)
    def profile(self, enable=True):
        h = self._handle
        h.check(h.lib.mp_profile_enable(h.ptr, 1 if enable else 0))

    def profile_read(self):
        h = self._handle
        cap = 4096
        names = (ctypes.c_char_p * cap)()
        ms = (ctypes.c_float * cap)()
        flop = (ctypes.c_double * cap)()
        n = ctypes.c_int(0)
        h.check(h.lib.mp_profile_read(h.ptr, names, ms, flop, cap, ctypes.byref(n)))
        return [(names[i].decode(), float(ms[i]), float(flop[i])) for i in range(n.value)]
