"""multipoint_amd -- MI355X-native (gfx950) implementation of the MultiPoint inference hot path.

Mirrors the reference package layout for that path only:
    multipoint_amd.models.MultiPoint        <- multipoint/models/MultiPoint.py
    multipoint_amd.utils.{box_nms, interpolate_descriptors, get_matches, ...}
                                            <- multipoint/utils/{utils,matching}.py
    multipoint_amd.datasets                 <- dict schema of multipoint/datasets/ImagePairDataset.py
All compute goes through the C ABI of libmultipoint_hip.so (include/multipoint_hip.h).
"""
import os as _os

# Hardware queues (read by the ROCm runtime when it initialises, i.e. at the first GPU call -- importing torch is not one):
# an RCCL communicator creates streams of its own, and with the runtime's default of 4 hardware queues the pipeline's
# post-processing stream then shares a queue with the convolution stream; the two serialise and a step gets 4 % longer
# (DESIGN.md section 6).  Set here, before anything of this package can touch the GPU, so that every user of
# multipoint_amd.dist gets it -- not only bench.py.  An explicit setting in the environment wins.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

from . import _lib  # noqa: F401,E402
from . import models, utils, datasets  # noqa: F401,E402
from .pipeline import PairPipeline  # noqa: F401,E402

__version__ = '0.1.0'
