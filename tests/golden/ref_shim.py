"""Import shim for the reference (runs ONLY in the build container where /root/reference exists;
SURVEY.md Appendix A).  Registers import-time stubs for the third-party modules that are not
installed (cv2, h5py, torchvision) and patches the two removed aliases the reference still uses.
Nothing from the reference is copied: it is imported in place."""
import collections
import collections.abc
import sys
import types

REFERENCE_ROOT = '/root/reference'


def install():
    for name in ['cv2', 'h5py', 'torchvision', 'torchvision.ops', 'torchvision.ops.boxes']:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['torchvision'].ops = sys.modules['torchvision.ops']
    sys.modules['torchvision.ops'].boxes = sys.modules['torchvision.ops.boxes']
    sys.modules['torchvision.ops'].nms = None                 # box_nms is NOT runnable (torchvision absent)
    sys.modules['torchvision.ops.boxes'].batched_nms = None
    collections.Mapping = collections.abc.Mapping             # utils.py:22 on py>=3.10
    import numpy as np
    for a, b in (('bool', bool), ('int', int), ('float', float)):
        if not hasattr(np, a):
            setattr(np, a, b)                                 # numpy>=1.24

    class _DMatch:                                            # only for NNMatcher output
        def __init__(self, q, t, d):
            self.queryIdx, self.trainIdx, self.distance = q, t, d
    sys.modules['cv2'].DMatch = _DMatch
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    import multipoint.models as models
    import multipoint.utils as utils
    return models, utils
