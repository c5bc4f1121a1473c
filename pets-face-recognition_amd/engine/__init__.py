from .controller import Controller  # noqa: F401
from .trainer import Trainer  # noqa: F401
from .ddp import BucketReducer, FlatDDP, GenericDDP  # noqa: F401
