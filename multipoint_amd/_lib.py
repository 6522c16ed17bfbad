"""ctypes binding of libmultipoint_hip.so (C ABI: include/multipoint_hip.h).

There is NO fallback: if the shared library is missing or no gfx950 device is visible, every compute
entry point raises.  PyTorch is used only for device memory and streams.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MP_LIB selects another build of the library (developer tools: the -DMP_TIMING instrumented variant)
LIB_PATH = os.environ.get('MP_LIB') or os.path.join(_HERE, 'libmultipoint_hip.so')



def debug_switch(key, default=None):
    """Developer switch `key` of the MP_DEBUG environment variable (a comma-separated list of `key` / `key=value`; the library
    reads the kernel-selection keys in mp_create, see csrc/api.hip): the value behind '=', '1' for a bare key, `default` if absent."""
    for tok in os.environ.get('MP_DEBUG', '').split(','):
        k, eq, v = tok.strip().partition('=')
        if k == key:
            return v if eq else '1'
    return default


MP_OK = 0
c_void_p, c_int, c_float, c_ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong


class ModelConfig(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ('multispectral', 'descriptor_head', 'descriptor_size',
                                     'normalize_descriptors', 'final_batchnorm', 'reflection_pad',
                                     'bn_first', 'double_convolution', 'channel_version', 'batchnorm',
                                     'key_layout', 'softmax_mode', 'mixed_precision', 'conv_algorithm', 'batch_invariant')]


class Tensor(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char_p), ('data', c_void_p), ('numel', c_ll)]


# name -> (restype, argtypes); every symbol declared in include/multipoint_hip.h
SIGNATURES = {
    'mp_create': (c_int, [ctypes.POINTER(c_void_p), c_int]),
    'mp_destroy': (None, [c_void_p]),
    'mp_last_error': (ctypes.c_char_p, [c_void_p]),
    'mp_version': (ctypes.c_char_p, []),
    'mp_device_shape': (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    'mp_load_weights': (c_int, [c_void_p, ctypes.POINTER(ModelConfig), ctypes.POINTER(Tensor), c_int]),
    'mp_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                           c_void_p, c_void_p]),
    'mp_box_nms': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, ctypes.c_double,
                           c_int, c_void_p, c_int, c_void_p]),
    'mp_detect_keypoints': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float,
                                    ctypes.c_double, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'mp_nms_unresolved': (c_int, [c_void_p, ctypes.POINTER(c_int), c_void_p]),
    'mp_topk_tie_guard': (c_int, [c_void_p, c_float, c_int]),
    'mp_nms_tie_guard': (c_int, [c_void_p, c_int]),
    'mp_topk_ambiguous': (c_int, [c_void_p, ctypes.POINTER(c_int), c_int, ctypes.POINTER(c_int), c_void_p]),
    'mp_extract_keypoints': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p,
                                     c_void_p, c_void_p, c_void_p]),
    'mp_sample_descriptors': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'mp_match_mutual_nn': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_int, c_int,
                                   c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mp_match_knn2': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_int, c_int, c_int, c_int,
                              c_void_p, c_void_p, c_void_p]),
    'mp_match_threshold': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_int, c_int, c_int, c_int,
                                   c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mp_pair_metrics': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                c_void_p, c_void_p, c_void_p]),
    'mp_repeatability': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, ctypes.c_double,
                                 c_void_p, c_void_p]),
    'mp_find_homography': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_double, c_int,
                                   ctypes.c_ulonglong, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mp_detector_metrics': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mp_warp_perspective': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p, c_void_p]),
    'mp_warp_perspective_cv': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'mp_ha_valid_mask': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'mp_ha_begin': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'mp_ha_accumulate': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p]),
    'mp_ha_finalize': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    'mp_gaussian_filter': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'mp_profile_enable': (c_int, [c_void_p, c_int]),
    'mp_profile_read': (c_int, [c_void_p, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_float),
                                ctypes.POINTER(ctypes.c_double), c_int, ctypes.POINTER(c_int)]),
}

_lib = None
_lock = threading.Lock()
_handles = {}


def load_library():
    """dlopen the HIP library and bind every C-ABI symbol (no GPU needed for this step)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    'multipoint_amd: %s not found. Build it with `python -m multipoint_amd.build` '
                    '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            _lib = lib
    return _lib


class MultiPointHipError(RuntimeError):
    pass


def _err(lib, handle):
    msg = lib.mp_last_error(handle)
    return msg.decode() if msg else 'unknown error'


def check(rc, handle=None):
    if rc != MP_OK:
        lib = load_library()
        msg = _err(lib, handle)
        if rc == -1:
            raise ValueError(msg)
        raise MultiPointHipError('libmultipoint_hip error %d: %s' % (rc, msg))


def require_cuda(device=None):
    if not torch.cuda.is_available():
        raise RuntimeError('multipoint_amd needs an MI355X (gfx950) visible to PyTorch-ROCm; '
                           'torch.cuda.is_available() is False and there is no CPU fallback.')
    if device is None:
        return torch.device('cuda', torch.cuda.current_device())
    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError('multipoint_amd computes on the GPU only (got device %s)' % device)
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    return device


class Handle:
    """One mp_handle per (process, device)."""

    def __init__(self, device_index):
        self.lib = load_library()
        self.device_index = device_index
        self.ptr = c_void_p()
        rc = self.lib.mp_create(ctypes.byref(self.ptr), device_index)
        if rc != MP_OK:
            raise MultiPointHipError('mp_create failed: %s' % _err(self.lib, None))

    def check(self, rc):
        check(rc, self.ptr)

    def device_shape(self):
        """(compute units, XCDs, workgroups of a one-per-CU persistent launch) the handle derived from the device."""
        a, b, c = c_int(), c_int(), c_int()
        self.check(self.lib.mp_device_shape(self.ptr, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def __del__(self):
        try:
            if self.ptr:
                self.lib.mp_destroy(self.ptr)
        except Exception:
            pass


def get_handle(device=None, key='default'):
    """Shared handle for stateless ops (nms / sampling / matching); models own their own handle."""
    device = require_cuda(device)
    k = (device.index, key)
    with _lock:
        h = _handles.get(k)
    if h is None:
        h = Handle(device.index)
        with _lock:
            _handles[k] = h
    return h


def stream_ptr(device):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(None)
