# mirrors multipoint/datasets/__init__.py for the prediction path
from .synthetic_pairs import SyntheticPairs  # noqa: F401
from .image_pair_dataset import ImagePairDataset  # noqa: F401
from . import augmentation  # noqa: F401,E402


def loader_num_workers(dataset, requested):
    """DataLoader worker count for a dataset of this package: samples whose homographic augmentation runs on the GPU
    must be produced in the main process (forked loader workers cannot use the device), so `prediction.num_worker`
    of the reference configs is honoured only when the dataset does no GPU work."""
    cfg = getattr(dataset, 'config', {}) or {}
    gpu_work = bool(cfg.get('augmentation', {}).get('homographic', {}).get('enable', False))
    if gpu_work and requested:
        print('INFO: homographic augmentation runs on the GPU inside the dataset; using num_workers=0 '
              'instead of {}'.format(requested))
        return 0
    return int(requested)
from .synthetic_weights import make_weights, make_images, SHIPPED_MODEL_CONFIG  # noqa: F401,E402
