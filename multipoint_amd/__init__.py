"""multipoint_amd -- MI355X-native (gfx950) implementation of the MultiPoint inference hot path.

Mirrors the reference package layout for that path only:
    multipoint_amd.models.MultiPoint        <- multipoint/models/MultiPoint.py
    multipoint_amd.utils.{box_nms, interpolate_descriptors, get_matches, ...}
                                            <- multipoint/utils/{utils,matching}.py
    multipoint_amd.datasets                 <- dict schema of multipoint/datasets/ImagePairDataset.py
All compute goes through the C ABI of libmultipoint_hip.so (include/multipoint_hip.h).
"""
from . import _lib  # noqa: F401
from . import models, utils, datasets  # noqa: F401
from .pipeline import PairPipeline  # noqa: F401

__version__ = '0.1.0'
