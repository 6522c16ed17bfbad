from .multipoint import MultiPoint  # noqa: F401
