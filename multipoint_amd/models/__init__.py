# mirrors multipoint/models/__init__.py for the network classes of the accelerated path
from .multipoint import MultiPoint  # noqa: F401
from .superpoint_magicleap import SuperPointMagicLeap  # noqa: F401
