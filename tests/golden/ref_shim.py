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


def install_homographies():
    """multipoint.utils.homographies of the reference with its absent third-party calls (cv2.getPerspectiveTransform /
    warpPerspective / erode, kornia's dst_norm_to_dst_norm / homography_warp) supplied by the oracle's restatements
    (oracle/ha_oracle.py): what then runs is the REFERENCE's own sample_homography, compute_valid_mask and
    homographic_adaptation(_multispectral) driver code."""
    install()
    import cv2
    import multipoint.utils.homographies as RH
    from oracle import ha_oracle as HA
    cv2.getPerspectiveTransform = HA.cv2_get_perspective_transform
    cv2.INTER_NEAREST = 0
    cv2.warpPerspective = lambda src, M, dsize, flags=None: HA.cv2_warp_perspective_nearest(src, M, dsize)
    cv2.erode = lambda src, kernel, iterations=1: HA.cv2_erode(src, kernel, iterations)
    ns = types.SimpleNamespace
    RH.kornia = ns(geometry=ns(transform=ns(imgwarp=ns(dst_norm_to_dst_norm=HA.dst_norm_to_dst_norm))))
    RH.homography_warp = HA.kornia_homography_warp
    RH.kornia_available = True
    return RH
