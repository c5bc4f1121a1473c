"""CPU restatement (torch fp32) of the reference's cosine-margin heads, focal loss, pair similarity and SGD step.

  arc_margin_logits   ← /root/reference/losses/large_margin.py:69-84  (ArcMarginProduct.forward)
  add_margin_logits   ← /root/reference/losses/large_margin.py:30-40  (AddMarginProduct.forward)
  focal_loss          ← /root/reference/losses/losses.py:22-28        (FocalLoss.forward)
  similarity_f        ← /root/reference/configs/dog_fe/fe_dogs_config.py:89-93
  sgd_step            ← torch.optim.SGD semantics used by fe_dogs_config.py:123-133 (momentum, per-group lr / wd)
Checked against the imported reference by oracle/make_golden.py."""
import math

import torch
import torch.nn.functional as F


def _unit_rows(t, eps=1e-12):
    return t / t.norm(dim=1, keepdim=True).clamp_min(eps)


def cosine(x, w):
    return _unit_rows(x) @ _unit_rows(w).t()


def arc_margin_logits(x, w, label, s=64.0, m=0.5, easy_margin=False):
    c = cosine(x, w)
    sn = torch.sqrt(1.0 - c.pow(2))
    shifted = c * math.cos(m) - sn * math.sin(m)
    if easy_margin:
        shifted = torch.where(c > 0, shifted, c)
    else:
        shifted = torch.where(c > math.cos(math.pi - m), shifted, c - math.sin(math.pi - m) * m)
    hot = torch.zeros_like(c)
    hot[torch.arange(c.shape[0]), label.long()] = 1.0
    return (hot * shifted + (1.0 - hot) * c) * s


def add_margin_logits(x, w, label, s=64.0, m=0.5):
    c = cosine(x, w)
    hot = torch.zeros_like(c)
    hot[torch.arange(c.shape[0]), label.long()] = 1.0
    return (hot * (c - m) + (1.0 - hot) * c) * s


def focal_loss(logits, label, gamma=0.0):
    lse = torch.logsumexp(logits, dim=1)
    logp = lse - logits[torch.arange(logits.shape[0]), label.long()]
    p = torch.exp(-logp)
    return ((1.0 - p) ** gamma * logp).mean()


def similarity_f(a, b, eps=1e-8):
    """(cos+1)/2 of row pairs; F.cosine_similarity semantics (each norm clamped at eps)."""
    num = (a * b).sum(dim=1)
    den = a.norm(dim=1).clamp_min(eps) * b.norm(dim=1).clamp_min(eps)
    return (num / den + 1.0) / 2.0


def sgd_step(params, grads, bufs, lr, momentum=0.9, weight_decay=0.0, first=False):
    """in-place torch.optim.SGD update over lists of tensors"""
    for p, g, b in zip(params, grads, bufs):
        d = g + weight_decay * p if weight_decay else g.clone()
        if momentum:
            if first:
                b.copy_(d)
            else:
                b.mul_(momentum).add_(d)
            d = b
        p.sub_(lr * d)
