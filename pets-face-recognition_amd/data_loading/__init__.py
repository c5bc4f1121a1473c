from .synthetic import SyntheticRecDataset, RecSubset  # noqa: F401
from .pairs import PairGenerator  # noqa: F401
from .augment import DeviceAugmentation, train_augmentation, val_augmentation  # noqa: F401
from .prefetch import DevicePrefetcher  # noqa: F401
