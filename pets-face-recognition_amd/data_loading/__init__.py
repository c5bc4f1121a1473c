from .synthetic import SyntheticRecDataset, RecSubset  # noqa: F401
from .pairs import PairGenerator  # noqa: F401
