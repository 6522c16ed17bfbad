# mirrors multipoint/utils/__init__.py for the hot path (matching + utils)
from .matching import *  # noqa: F401,F403
from .utils import *  # noqa: F401,F403
from .evaluation import *  # noqa: F401,F403  (compute_descriptor_metrics, pair_metrics, ...)
from .homographies import *  # noqa: F401,F403  (homographic adaptation: export_keypoints.py)
