# mirrors multipoint/datasets/__init__.py for the prediction path
from .synthetic_pairs import SyntheticPairs  # noqa: F401
from .image_pair_dataset import ImagePairDataset  # noqa: F401
