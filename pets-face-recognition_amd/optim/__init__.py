from .fused import FusedSGD, FusedAdamW  # noqa: F401
