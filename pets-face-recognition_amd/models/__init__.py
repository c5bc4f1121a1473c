"""Model registry — mirrors /root/reference/models/__init__.py (swin_t/s/b/l, SwinTransformer) and adds the
torchvision-compatible ResNets the reference's FE configs build (configs/dog_fe/fe_dogs_config.py:102-103)."""
from .resnet import ResNet, BasicBlock, Bottleneck, resnet18, resnet34, resnet50, resnet101  # noqa: F401
from .swin import SwinTransformer, swin_t, swin_s, swin_b, swin_l  # noqa: F401
