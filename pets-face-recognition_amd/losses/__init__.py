"""`SoftmaxBasedMetricLearning` — backbone + cosine-margin head + criterion, with the reference's constructor
signature, attribute names (`module`, `add_margin`, `focal_loss`) and forward contract
(/root/reference/losses/__init__.py:8-46): `forward(img, label=None)` returns the embedding tensor when `label is
None`, else `{'loss', 'emb', 'logits'}`; a list/tuple of images is embedded one by one and concatenated."""
import torch
import torch.nn as nn

from .large_margin import ArcMarginProduct, AddMarginProduct, _MarginHead
from .losses import FocalLoss


class SoftmaxBasedMetricLearning(nn.Module):
    def __init__(self, model, num_class, embedding_size=512, s=64.0, m=0.5, is_focal=False, loss_kwargs=None,
                 arc_margin=False, easy_margin=False):
        super().__init__()
        if arc_margin:
            self.add_margin = ArcMarginProduct(embedding_size, num_class, s=s, m=m, easy_margin=easy_margin)
        else:
            self.add_margin = AddMarginProduct(embedding_size, num_class, s=s, m=m)
        loss_kwargs = loss_kwargs or {}
        self.focal_loss = FocalLoss(num_class=num_class, **loss_kwargs) if is_focal else nn.CrossEntropyLoss(**loss_kwargs)
        self.module = model
        self.softmax = nn.Softmax(dim=1)
        self.return_logits = True  # the reference always returns logits; set False to skip writing them (bench)

    def _fusable(self, emb):
        if not emb.is_cuda or not isinstance(self.add_margin, _MarginHead):
            return None
        fl = self.focal_loss
        if isinstance(fl, FocalLoss) and not fl.adaptive_flag:
            return float(fl.gamma)
        if type(fl) is nn.CrossEntropyLoss and fl.weight is None and fl.reduction == "mean" and \
                getattr(fl, "label_smoothing", 0.0) == 0.0 and fl.ignore_index == -100:
            return 0.0
        return None

    def forward(self, img, label=None, **__):
        if isinstance(img, (list, tuple)):
            tensor = torch.cat([self.module(i) for i in img], dim=0)
        else:
            tensor = self.module(img)
        if label is None:
            return tensor
        gamma = self._fusable(tensor)
        if gamma is not None:
            from ._head_hip import MarginCEFunction, resolve_dtype
            head = self.add_margin
            dt = head.compute_dtype or getattr(self.module, "compute_dtype", None)
            loss, logits = MarginCEFunction.apply(tensor, head.weight, label, head.hip_mode(), head.s, head.m, gamma,
                                                  resolve_dtype(dt), self.return_logits)
            return {'loss': loss, 'emb': tensor, 'logits': logits if self.return_logits else None}
        logits = self.add_margin(tensor, label)
        loss = self.focal_loss(logits, label)
        return {'loss': loss, 'emb': tensor, 'logits': logits}


class DummyWrapper(nn.Module):
    def __init__(self, model, *_, **__):
        super().__init__()
        self.module = model

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
