from .ddp import BucketReducer, FlatDDP  # noqa: F401
